b() { echo "$1 bench: $(timeout -k 10 200 python bench.py --no-cpu-baseline --set $2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
for rep in 1 2 3; do
b s1 9=1
b s2 9=2
b s3 9=3
b s4 9=4
done

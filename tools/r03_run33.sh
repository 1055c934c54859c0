b() { echo "$1 bench: $(timeout -k 10 200 python bench.py --no-cpu-baseline --set $2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
for rep in 1 2; do
b base 6=24
b q6_16 6=16
b q6_32 6=32
b q6_40 6=40
b r5_16 5=16
b r5_32 5=32
b q7_48 7=48
b q7_24 7=24
done

V=fypraytracer_amd/csrc/variants
b() { echo "$1 bench: $(FYPRT_EXP_OVERLAP_CAP=$2 FYPRT_LIB=$V/libfyprt_cap.so timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
for rep in 1 2 3; do
b cap4 4
b cap3 3
b cap5 5
b cap6 6
done

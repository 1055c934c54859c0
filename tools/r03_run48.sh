V=fypraytracer_amd/csrc/variants
b() { echo "$1 bench: $(env $2 FYPRT_LIB=$V/libfyprt_prio.so timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"; }
for rep in 1 2 3; do
b front_lo A=1
b front_hi FYPRT_EXP_FRONT_HI=1
done

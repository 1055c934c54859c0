V=fypraytracer_amd/csrc/variants
b() { echo "$1 bench: $(FYPRT_LIB=${2:-fypraytracer_amd/csrc/libfyprt.so} timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernels']['k_di_part2_setup']['avg_ms_alone'])")"; }
for rep in 1 2 3; do
b base ""
b setup7 $V/libfyprt_setup7.so
b setup8 $V/libfyprt_setup8.so
done

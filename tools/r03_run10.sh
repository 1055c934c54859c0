L=fypraytracer_amd/csrc/variants/libfyprt_hack3.so
for rep in 1 2; do
FYPRT_LIB=$L python tools/band_probe.py 8 3 100 recompute 2>&1 | tail -n 1
FYPRT_LIB=$L FYPRT_HACK_3STREAM=1 python tools/band_probe.py 8 3 100 recompute 2>&1 | tail -n 1
FYPRT_LIB=$L python tools/band_probe.py 8 3 100 exchange 2>&1 | tail -n 1
FYPRT_LIB=$L FYPRT_HACK_3STREAM=1 python tools/band_probe.py 8 3 100 exchange 2>&1 | tail -n 1
done
FYPRT_LIB=$L FYPRT_HACK_3STREAM=1 python tools/band_probe.py 8 3 100 exchange 2=2 2>&1 | tail -n 1
FYPRT_LIB=$L FYPRT_HACK_3STREAM=1 python tools/band_probe.py 8 3 100 exchange 2=3 2>&1 | tail -n 1
FYPRT_LIB=$L FYPRT_HACK_3STREAM=1 python tools/band_probe.py 8 3 100 exchange 2=6 2>&1 | tail -n 1
FYPRT_LIB=$L python tools/band_probe.py 1 0 60 recompute 2>&1 | tail -n 1
FYPRT_LIB=$L FYPRT_HACK_3STREAM=1 python tools/band_probe.py 1 0 60 recompute 2>&1 | tail -n 1
FYPRT_LIB=$L FYPRT_HACK_3STREAM=1 python tools/band_probe.py 4 1 60 exchange 2>&1 | tail -n 1
FYPRT_LIB=$L python tools/band_probe.py 4 1 60 exchange 2>&1 | tail -n 1

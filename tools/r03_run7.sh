for g in 0 1 2 3 4 5; do python tools/band_probe.py 8 3 80 recompute 2=$g 2>&1 | tail -n 1; done
for g in 0 2 3 4; do python tools/band_probe.py 8 3 80 exchange 2=$g 2>&1 | tail -n 1; done
for g in 0 2 3 4; do python tools/band_probe.py 8 0 80 recompute 2=$g 2>&1 | tail -n 1; done
for g in 0 3 4; do python tools/band_probe.py 4 1 80 recompute 2=$g 2>&1 | tail -n 1; done
for g in 0 3 4; do python tools/band_probe.py 2 0 80 recompute 2=$g 2>&1 | tail -n 1; done
for g in 0 4 5; do python tools/band_probe.py 1 0 80 recompute 2=$g 2>&1 | tail -n 1; done

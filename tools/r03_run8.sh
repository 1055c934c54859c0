for kv in "2=4 8=22" "2=4 8=17" "2=4 8=15" "2=5 8=15" "2=6 8=15" "2=3 8=15" "2=0 8=15"; do python tools/band_probe.py 8 3 80 recompute $kv 2>&1 | tail -n 1; done
for kv in "2=4 8=22" "2=4 8=15" "2=5 8=15"; do python tools/band_probe.py 8 3 80 exchange $kv 2>&1 | tail -n 1; done
for kv in "2=0 8=22" "2=4 8=15" "2=0 8=15" "2=0 8=17"; do python tools/band_probe.py 1 0 80 recompute $kv 2>&1 | tail -n 1; done
for kv in "2=4 8=22" "2=4 8=15" "2=5 8=15"; do python tools/band_probe.py 4 1 80 recompute $kv 2>&1 | tail -n 1; done

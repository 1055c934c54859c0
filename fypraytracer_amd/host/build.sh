#!/usr/bin/env bash
# Builds the headless C++ harness against libfyprt.so (facade: Renderer.h, stand-in scene types: HostTypes.h).
set -euo pipefail
cd "$(dirname "$0")"
g++ -O2 -std=c++17 -Wall harness.cpp -o harness -L../csrc -lfyprt -lz -Wl,-rpath,'$ORIGIN/../csrc'
g++ -O2 -std=c++17 -Wall -I../../include misutils_check.cpp -o misutils_check -lz
g++ -O2 -std=c++17 -Wall -ffp-contract=off -I../../include scene_check.cpp -o scene_check -lz
g++ -O2 -std=c++17 -Wall -ffp-contract=off -fPIC -shared host_io.cpp -o libfyprt_host.so -lz
echo "built $(pwd)/harness $(pwd)/misutils_check $(pwd)/scene_check $(pwd)/libfyprt_host.so"

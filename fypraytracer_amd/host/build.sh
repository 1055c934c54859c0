#!/usr/bin/env bash
# Builds the headless C++ harness against libfyprt.so (facade: Renderer.h, stand-in scene types: HostTypes.h).
set -euo pipefail
cd "$(dirname "$0")"
g++ -O2 -std=c++17 -Wall harness.cpp -o harness -L../csrc -lfyprt -Wl,-rpath,'$ORIGIN/../csrc'
g++ -O2 -std=c++17 -Wall -I../../include misutils_check.cpp -o misutils_check
echo "built $(pwd)/harness $(pwd)/misutils_check"

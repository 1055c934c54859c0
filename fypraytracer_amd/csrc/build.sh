#!/usr/bin/env bash
# Builds libfyprt.so (gfx950).  Host-only builders with g++, kernels + C ABI with hipcc.
# -ffp-contract=off everywhere: the arithmetic contract of DESIGN.md §4.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CXXFLAGS="-O2 -std=c++17 -fPIC -ffp-contract=off -fno-math-errno -fopenmp -Wall"
g++ $CXXFLAGS -c bvh_build.cpp -o bvh_build.o
g++ $CXXFLAGS -c lighttree_build.cpp -o lighttree_build.o
$HIPCC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
    ${FYPRT_EXTRA_HIPCC_FLAGS:-} -c fyprt.hip -o ${FYPRT_OBJ:-fyprt.o}
$HIPCC -shared -fPIC --offload-arch=gfx950 ${FYPRT_OBJ:-fyprt.o} bvh_build.o lighttree_build.o -lgomp -o ${FYPRT_OUT:-libfyprt.so}
echo "built $(pwd)/${FYPRT_OUT:-libfyprt.so}"

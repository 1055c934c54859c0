#!/usr/bin/env bash
# Builds an alternative libfyprt (A/B experiments):  bash tools/build_variant.sh <name> <extra hipcc flags...>
#   -> fypraytracer_amd/csrc/variants/libfyprt_<name>.so   (use with FYPRT_LIB=... ; travels to the GPU box, ignored by git)
set -euo pipefail
NAME=$1; shift
cd "$(dirname "$0")/../fypraytracer_amd/csrc"
mkdir -p variants
FYPRT_EXTRA_HIPCC_FLAGS="$*" FYPRT_OBJ=variants/fyprt_$NAME.o FYPRT_OUT=variants/libfyprt_$NAME.so bash build.sh | tail -1

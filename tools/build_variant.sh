#!/bin/bash
# Builds a variant of the HIP library with extra -D flags for A/B timing:  tools/build_variant.sh <name> [-DRT_... ...]
# -> build/variants/<name>.so; run with GPURT_LIB=build/variants/<name>.so python bench.py ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $R/build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize -shared -fPIC "$@" -o $R/build/variants/$NAME.so $R/gpuraytracer_amd/csrc/rt_api.hip

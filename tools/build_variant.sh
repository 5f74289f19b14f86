#!/bin/bash
# Experiment helper: build the product library with extra -D flags into yart_amd/_variants/NAME.so
# (git-ignored; travels to the GPU box). Usage: tools/build_variant.sh NAME "-DFOO=1 ..."
set -e
cd "$(dirname "$0")/../yart_amd/csrc"
mkdir -p ../_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -fhip-fp32-correctly-rounded-divide-sqrt $2 -shared -o ../_variants/$1.so yart_hip.hip _gen/lut_data.cpp -lz -ldl
echo built $1

#!/bin/bash
# Experiment helper: build the product library with extra -D flags into yart_amd/_variants/NAME.so
# (git-ignored; travels to the GPU box). Usage: tools/build_variant.sh NAME "-DFOO=1 ..."
# (five translation units of yart_hip.hip, compiled in parallel: csrc/Makefile)
set -e
cd "$(dirname "$0")/../yart_amd/csrc"
mkdir -p ../_variants _gen/variant_$1
[ -f _gen/lut_data.cpp ] || python3 gen_lut_data.py _gen/lut_data.cpp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -w $2"
for u in 0 1 2 3 4; do /opt/rocm/bin/hipcc $F -DYART_TU=$u -c -o _gen/variant_$1/tu$u.o yart_hip.hip & done
/opt/rocm/bin/hipcc $F -x c++ -c -o _gen/variant_$1/lut.o _gen/lut_data.cpp &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../_variants/$1.so _gen/variant_$1/*.o -lz -ldl
rm -rf _gen/variant_$1
echo built $1

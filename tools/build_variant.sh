#!/bin/bash
# Experiment helper: build the product library with extra -D flags into yart_amd/_variants/NAME.so
# (git-ignored; travels to the GPU box). Usage: tools/build_variant.sh NAME "-DFOO=1 ..."
# (five translation units of yart_hip.hip, compiled in parallel: csrc/Makefile)
cd "$(dirname "$0")/../yart_amd/csrc" || exit 1
mkdir -p ../_variants _gen/variant_$1
[ -f _gen/lut_data.cpp ] || python3 gen_lut_data.py _gen/lut_data.cpp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -w $2"
pids=()
for u in 0 1 2 3 4; do /opt/rocm/bin/hipcc $F -DYART_TU=$u -c -o _gen/variant_$1/tu$u.o yart_hip.hip 2> _gen/variant_$1/tu$u.err & pids+=($!); done
/opt/rocm/bin/hipcc -O2 -fPIC -w -x c++ -c -o _gen/variant_$1/lut.o _gen/lut_data.cpp & pids+=($!)      # (plain data: no experiment flags)
ok=1
for p in "${pids[@]}"; do wait $p || ok=0; done
if [ $ok != 1 ]; then echo "build_variant $1: a unit failed to compile"; grep -h -m3 -i "error" _gen/variant_$1/*.err | cut -c1-300; rm -rf _gen/variant_$1; exit 1; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../_variants/$1.so _gen/variant_$1/*.o -lz -ldl || exit 1
rm -rf _gen/variant_$1
echo built $1

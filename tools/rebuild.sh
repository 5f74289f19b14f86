#!/bin/bash
# Rebuild every native target from the repo root (lib, instrumented twin, stats debug lib, hostsim).
set -e
cd "$(dirname "$0")/.."
if ! make -C yart_amd/csrc > /tmp/yart_make.log 2>&1; then grep -E "error" /tmp/yart_make.log | head -20; echo "BUILD FAILED"; exit 1; fi
if [ "$1" == "stats" ]; then
  (cd yart_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
     -DYART_TRACE_STATS=1 -shared -o ../libyart_hip_stats.so yart_hip.hip _gen/lut_data.cpp -lz -ldl 2>&1 | grep -E "error" || true)
fi
mkdir -p tests/hostsim/_build
g++ -std=c++17 -O2 -ffp-contract=off -o tests/hostsim/_build/hostsim tests/hostsim/hostsim.cpp yart_amd/csrc/_gen/lut_data.cpp -lpthread
ls -la yart_amd/*.so | awk '{print $5, $9}'

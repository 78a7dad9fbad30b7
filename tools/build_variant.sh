#!/bin/bash
# Build a variant of the device library for an A/B run: tools/build_variant.sh <name> [extra hipcc flags...]
# writes raytracing_amd/lib/librtamd_<name>.so (same flags as raytracing_amd/build.py plus the extras).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared "$@" \
  -I"$ROOT/include" -I"$ROOT/raytracing_amd/csrc" "$ROOT/raytracing_amd/csrc/rtamd.hip" "$ROOT/raytracing_amd/csrc/tlas_build.hip" \
  -o "$ROOT/raytracing_amd/lib/librtamd_$NAME.so"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$ROOT/raytracing_amd/lib/librtamd_$NAME.so" > /dev/null 2>&1 || true
echo "built librtamd_$NAME.so"

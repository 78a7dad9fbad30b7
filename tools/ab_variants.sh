#!/bin/bash
# A/B of several builds on the GPU box: tools/ab_variants.sh lib1.so lib2.so ...   (under raytracing_amd/lib/)
# For each variant after the first: the kernels-against-oracle parity tests, then the bench line (mesh, spheres, instanced, REF frames).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
: > gpurun_out/ab_variants.txt
first=1
for lib in "$@"; do
  if [ $first = 0 ]; then
    RTAMD_LIB=$PWD/raytracing_amd/lib/$lib timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_reference_scene.py -x -q --timeout 300 --timeout-method thread -m gpu 2>&1 | tail -1 | sed "s/^/$lib parity: /" | tee -a gpurun_out/ab_variants.txt
  fi
  first=0
done
tools/ab.sh -- "$@" | tee -a gpurun_out/ab_variants.txt

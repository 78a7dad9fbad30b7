#!/bin/bash
# Experiment runner on the GPU box: tools/exp_ab.sh <variant library under raytracing_amd/lib> [stats]
# parity suite with the variant first, then A/B against librtamd_base.so on the three bench scenes.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
V=$1
mkdir -p gpurun_out
RTAMD_LIB=$PWD/raytracing_amd/lib/$V timeout -k 10 900 python3 -u -m pytest tests/test_gpu_parity.py tests/test_reference_scene.py tests/test_dielectric.py -x -q --timeout 300 --timeout-method thread -m gpu 2>&1 | tail -5 | tee gpurun_out/exp_parity.txt
if [ "$2" = "stats" ]; then
( for w in mesh spheres instanced; do
    RTAMD_LIB=$PWD/raytracing_amd/lib/$V RTAMD_STATS=1 timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline --no-counters --no-secondary --steps 1 --warmup 0 2>&1 >/dev/null | grep stats | sed "s/^/$w: /"
  done ) > gpurun_out/exp_stats.txt
cat gpurun_out/exp_stats.txt
fi
tools/ab.sh -- librtamd_base.so $V | tee gpurun_out/exp_ab.txt
tools/ab.sh --workload instanced -- librtamd_base.so $V | tee -a gpurun_out/exp_ab.txt

#!/bin/bash
# Per-phase lane statistics of the path-pool kernel for several builds: tools/stats_variants.sh <workload> lib1.so lib2.so ...
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
WL=$1; shift
for lib in "$@"; do
  RTAMD_LIB=$PWD/raytracing_amd/lib/$lib RTAMD_STATS=1 timeout -k 10 200 python3 bench.py --workload $WL --no-cpu-baseline --no-counters --no-secondary --steps 1 --warmup 0 2>&1 >/dev/null | grep stats | sed "s/^/$lib $WL: /"
done | tee -a gpurun_out/stats_variants.txt

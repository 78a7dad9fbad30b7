#!/bin/bash
# One measuring session on the GPU box: kernel-trace profiles and counter passes of the bench step for the mesh
# (headline), spheres and 50 M-instanced workloads, and the per-phase lane statistics of the path-pool kernel.
# Usage: tools/measure.sh <tag>      (results under gpurun_out/: prof_<tag>_*, pmc_<tag>_*, <tag>_phase_lanes.txt;
#        afterwards, here: python tools/stamp_profiles.py <tag> copies what is kept into profiles/)
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
tools/gpu_profile.sh ${TAG}_mesh --steps 3 --warmup 1 --no-counters
tools/gpu_profile.sh ${TAG}_spheres --workload spheres --steps 3 --warmup 1 --no-counters
tools/gpu_profile.sh ${TAG}_instanced --workload instanced --steps 3 --warmup 1 --no-counters
tools/gpu_pmc.sh ${TAG}_mesh
tools/gpu_pmc.sh ${TAG}_spheres --workload spheres
tools/gpu_pmc.sh ${TAG}_instanced --workload instanced
( RTAMD_STATS=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-counters --no-secondary --steps 1 --warmup 0 2>&1 >/dev/null | grep stats | sed 's/^/mesh 1080p 256spp: /'
  RTAMD_STATS=1 timeout -k 10 200 python3 bench.py --workload spheres --no-cpu-baseline --no-counters --steps 1 --warmup 0 2>&1 >/dev/null | grep stats | sed 's/^/spheres 1080p 64spp: /'
  RTAMD_STATS=1 timeout -k 10 200 python3 bench.py --workload instanced --no-cpu-baseline --no-counters --steps 1 --warmup 0 2>&1 >/dev/null | grep stats | sed 's/^/instanced 1080p 64spp: /' ) > gpurun_out/${TAG}_phase_lanes.txt
cat gpurun_out/${TAG}_phase_lanes.txt

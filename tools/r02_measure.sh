#!/bin/bash
# One measuring session on the GPU box: kernel-trace profiles and counter passes of the bench step
# for the mesh (headline) and spheres workloads, plus a counter pass over the gather micro-benchmark
# to calibrate bytes per L1->L2 request.  Results under gpurun_out/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
tools/gpu_profile.sh r02_mesh --steps 3 --warmup 1 --no-counters
tools/gpu_profile.sh r02_spheres --workload spheres --steps 3 --warmup 1 --no-counters
tools/gpu_pmc.sh mesh
tools/gpu_pmc.sh spheres --workload spheres
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d "$ROOT/gpurun_out/pmc_stream" -o stream -- "$ROOT/tools/bin/l2_stream" > "$ROOT/gpurun_out/pmc_stream.log" 2>&1
cd "$ROOT"
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_stream/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, [f"{x:.4g}" for x in v])
PY
# streaming / gather ceilings of the cache levels, and the per-phase lane statistics of the path-pool kernel
timeout -k 10 120 tools/bin/l2_stream > gpurun_out/stream.json
( RTAMD_STATS=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-counters --no-secondary --steps 1 --warmup 0 2>&1 >/dev/null | grep stats | sed 's/^/mesh 1080p 256spp: /'
  RTAMD_STATS=1 timeout -k 10 200 python3 bench.py --workload spheres --no-cpu-baseline --no-counters --steps 1 --warmup 0 2>&1 >/dev/null | grep stats | sed 's/^/spheres 1080p 64spp: /'
  RTAMD_STATS=1 timeout -k 10 200 python3 bench.py --workload instanced --no-cpu-baseline --no-counters --steps 1 --warmup 0 2>&1 >/dev/null | grep stats | sed 's/^/instanced 1080p 64spp: /' ) > gpurun_out/phase_lanes.txt
cat gpurun_out/stream.json gpurun_out/phase_lanes.txt

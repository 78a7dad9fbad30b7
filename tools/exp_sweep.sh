#!/bin/bash
# tools/exp_sweep.sh <library> : nodeMin sweep of a variant library on the spheres and mesh scenes
cd "${GRAFT_REPO_ROOT:-.}"
V=$1
for wl in "spheres 64" "mesh 64" "instanced 32"; do
  set -- $wl
  for nm in 8 16 24 32 40 48; do
    RTAMD_LIB=$PWD/raytracing_amd/lib/$V RTAMD_NODE_MIN=$nm timeout -k 10 120 python3 bench.py --workload $1 --spp $2 --no-cpu-baseline --no-counters --no-secondary --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$V $1 nodeMin $nm: %.1f Mrays/s' % d['value'])"
  done
done

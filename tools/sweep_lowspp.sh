#!/bin/bash
# Threshold sweep of the path-pool kernel on frames of few samples per pixel: tools/sweep_lowspp.sh <workload> <spp>
cd "${GRAFT_REPO_ROOT:-.}"
WL=$1; SPP=$2
for rb in 8 16 24 32 40; do
  for nm in 4 8 16; do
    RTAMD_NODE_MIN=$nm RTAMD_REFILL_BELOW=$rb timeout -k 10 120 python3 bench.py --workload $WL --spp $SPP --no-cpu-baseline --no-counters --no-secondary --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$WL $SPP spp refillBelow $rb nodeMin $nm: %.3f ms %.1f Mrays/s' % (d['ms_per_step'], d['value']))"
  done
done

#!/bin/bash
# Threshold sweep of the path-pool kernel on the GPU box: tools/sweep_thresholds.sh <workload> <spp> "<nodeMin list>" "<refillBelow list>" "<leafGroupMin list>"
cd "${GRAFT_REPO_ROOT:-.}"
WL=$1; SPP=$2
for nm in $3; do
  for rb in $4; do
    for lg in $5; do
      RTAMD_NODE_MIN=$nm RTAMD_REFILL_BELOW=$rb RTAMD_LEAF_GROUP_MIN=$lg timeout -k 10 120 python3 bench.py --workload $WL --spp $SPP --no-cpu-baseline --no-counters --no-secondary --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$WL nodeMin $nm refillBelow $rb leafGroup $lg: %.1f Mrays/s' % d['value'])"
    done
  done
done

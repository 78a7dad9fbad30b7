#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
run() {  # label, env...
  label=$1; shift
  for wl in "mesh 64" "spheres 64" "instanced 32"; do
    w=${wl% *}; spp=${wl#* }
    env "$@" timeout -k 10 200 python3 bench.py --workload $w --spp $spp --no-cpu-baseline --no-counters --no-secondary --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$label $w: %.1f Mrays/s' % d['value'])"
  done
}
run base X=1
run pick_area RTAMD_EXP_PICK=1
run pick_count RTAMD_EXP_PICK=2
run cost_plain RTAMD_EXP_COST=1
run cost_plain_pick_area RTAMD_EXP_COST=1 RTAMD_EXP_PICK=1

#!/bin/bash
# tools/exp_env.sh VAR v1 v2 ... : bench the three scenes with an environment variable set to each value
cd "${GRAFT_REPO_ROOT:-.}"
VAR=$1; shift
for v in "$@"; do
  for wl in "mesh 64" "spheres 64" "instanced 32"; do
    w=${wl% *}; spp=${wl#* }
    env $VAR=$v timeout -k 10 200 python3 bench.py --workload $w --spp $spp --no-cpu-baseline --no-counters --no-secondary --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$v $w: %.1f Mrays/s' % d['value'])"
  done
done

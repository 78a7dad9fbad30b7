#!/bin/bash
# One environment knob against the three bench scenes: tools/sweep_env.sh <VARIABLE> "<values>"
cd "${GRAFT_REPO_ROOT:-.}"
VAR=$1
for v in $2; do
  for wl in mesh spheres instanced; do
    env $VAR=$v timeout -k 10 120 python3 bench.py --workload $wl --no-cpu-baseline --no-counters --no-secondary --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$v $wl: %.1f Mrays/s' % d['value'])"
  done
done

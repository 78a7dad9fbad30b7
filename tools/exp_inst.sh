#!/bin/bash
# tools/exp_inst.sh VAR v1 v2 ... : the 50 M instanced-triangle scene (64 spp) with an environment variable set to each value
cd "${GRAFT_REPO_ROOT:-.}"
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout -k 10 200 python3 bench.py --workload instanced --spp 64 --no-cpu-baseline --no-counters --no-secondary --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$VAR=$v instanced: %.1f Mrays/s' % d['value'])"
done

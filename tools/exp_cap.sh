cd "${GRAFT_REPO_ROOT:-.}"
for v in 134217728 268435456 536870912 1073741824; do
  RTAMD_POOL_MAX_PATHS=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-counters --no-secondary --steps 4 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('cap $v', d['value'], d['ms_per_step'])" || exit 1
done

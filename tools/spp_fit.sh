#!/bin/bash
# Frame time against samples per pixel (the slope is the work, the intercept the fixed cost of a frame): tools/spp_fit.sh <workload>
cd "${GRAFT_REPO_ROOT:-.}"
WL=$1
for spp in 1 2 4 8 16 32 64; do
  timeout -k 10 120 python3 bench.py --workload $WL --spp $spp --no-cpu-baseline --no-counters --no-secondary --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$WL $spp spp: %.3f ms %.1f Mrays/s kernel %.3f ms' % (d['ms_per_step'], d['value'], d['roofline']['kernel_ms'] if d.get('roofline') else -1))" || exit 1
done

#!/bin/bash
# Usage: tools/pool_sweep.sh "<workloads>" "<refill>" "<nodeMin>"   (kPool pipeline, env-var sweep)
for wl in $1; do for rf in $2; do for nm in $3; do
  printf "%s pool refill=%s nodeMin=%s " $wl $rf $nm
  RTAMD_PIPELINE=pool RTAMD_REFILL_BELOW=$rf RTAMD_NODE_MIN=$nm timeout -k 10 200 python -u bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-counters 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1),'Mrays/s', round(d['ms_per_step'],1),'ms')" || exit 1
done; done; done

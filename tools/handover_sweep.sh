#!/bin/bash
# Usage: tools/handover_sweep.sh "<workloads>" "<spreads>"   (time of the hand-over pass against the list spread)
for wl in $1; do for sp in $2; do
  printf "%s spread=%s " $wl $sp
  RTAMD_LIST_SPREAD=$sp timeout -k 10 200 python -u bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value'],1),'Mrays/s', round(d['ms_per_step'],2),'ms, hand-over', round(r['handover_ms'],3),'ms for', r['handover_pixels'],'pixels')" || exit 1
done; done

#!/bin/bash
# tools/envsweep.sh VAR "v1 v2 ..." [bench args]: bench Mrays/s for each value of one environment knob
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-counters --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
o=d.get('other_workloads',{}).get('spheres',{})
print('$VAR=$v', 'value %.1f ms %.2f' % (d['value'], d['ms_per_step']), 'spheres %.1f' % o.get('value',0))"
done

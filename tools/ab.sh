#!/bin/bash
# A/B of library builds on the GPU box: tools/ab.sh <bench args> -- lib1.so lib2.so ...   (paths under raytracing_amd/lib/)
ARGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
for lib in "$@"; do
  for rep in 1 2; do
    RTAMD_LIB=$PWD/raytracing_amd/lib/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-counters --steps 3 --warmup 1 "${ARGS[@]}" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
ow=d.get('other_workloads') or {}
print('$lib', 'rep$rep', 'value %.1f ms %.2f' % (d['value'], d['ms_per_step']), 'spheres %.1f' % ow.get('spheres',{}).get('value',0), 'instanced %.1f' % ow.get('instanced',{}).get('value',0),
      'REF ms mesh %.3f 4K %.3f spheres %.3f' % tuple(ow.get('ref',{}).get(k,{}).get('ms_per_render_call',0) for k in ('mesh_1080p','mesh_4k','spheres_1080p')))"
  done
done

#!/bin/bash
# The several-GPU configuration (3840x2160, 1024 spp in total) on ONE GPU, whole and as the share of one of 8 tile shards
# rehearsed through the shard arguments of bench.py: what one GPU does, as a yardstick for the scaling run.
cd "${GRAFT_REPO_ROOT:-.}"
timeout -k 10 400 python3 bench.py --width 3840 --height 2160 --spp 1024 --no-cpu-baseline --no-counters --no-secondary --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('4K 1024 spp, one GPU: %.1f Mrays/s, %.1f ms per step, %d segments' % (d['value'], d['ms_per_step'], d['config']['ray_segments_per_step']))"

#!/bin/bash
# Kernel-trace profile of bench.py on the GPU box.  Usage: tools/gpu_profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/ (scratch) — copy the *_kernel_stats.csv you want kept into profiles/.
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o "$TAG" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary "$@" > "$OUT/bench.log" 2>&1
tail -2 "$OUT/bench.log"
find "$OUT" -name "*kernel_stats.csv" | head -3 | while read f; do echo "== $f"; head -12 "$f"; done

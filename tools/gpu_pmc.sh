#!/bin/bash
# Hardware-counter passes over one bench launch.  Usage: tools/gpu_pmc.sh <tag> [bench args...]
# One rocprofv3 run per counter group (the tool does not split groups over passes).
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc $@ --output-format csv -d "$OUT/$name" -o "$name" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-counters --no-secondary --steps 1 --warmup 0 $BENCH_ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  python3 - "$OUT/$name" <<'PY'
import sys, glob, csv, collections
d = sys.argv[1]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float)
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if "rt::k" not in name: continue
        short = name.split("rt::")[1].split("(")[0]
        acc[(short, row["Counter_Name"])] += float(row["Counter_Value"])
    for k, v in sorted(acc.items()):
        print(f"{k[0]} {k[1]} {v:.6g}")
PY
}
BENCH_ARGS="$@"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU && \
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM && \
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum && \
run tcc2 FETCH_SIZE && \
run tcc3 WRITE_SIZE && \
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE

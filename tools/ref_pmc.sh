#!/bin/bash
# counter passes over REF-mode frames (20 frames + 5 warm-up = 25 launches of rt::kFast<0>)
ROOT=${GRAFT_REPO_ROOT:-.}
OUT=$ROOT/gpurun_out/pmc_ref
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for grp in "tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU" "grbm GRBM_GUI_ACTIVE"; do
  set -- $grp; name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc $@ --output-format csv -d $OUT/$name -o $name -- python3 $ROOT/tools/ref_frames.py mesh 1920 1080 20 > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; exit 1; }
  python3 - $OUT/$name <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "rt::kFast<0" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k, v in sorted(acc.items()):
    print(f"{k} per launch {v / max(1, n[k]):.6g}  ({n[k]} launches)")
PY
done

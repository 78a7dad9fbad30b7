#!/bin/bash
# Kernel timeline of REF-mode frames (rocprofv3 kernel trace of tools/ref_frame_latency.py): tools/ref_timeline.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d "$ROOT/gpurun_out/prof_ref" -o ref --output-format csv -- python3 "$ROOT/tools/ref_frame_latency.py" 20 > "$ROOT/gpurun_out/ref_tl.log" 2>&1 || exit 1
cd "$ROOT"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_ref/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "kFast" in r["Kernel_Name"]]
for first in (20, 45, 70):          # a late frame of each of the three scenes (25 calls each)
    a, b = idx[first], idx[first + 2]
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a - 4:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%-40s %9.3f %9.3f (%7.3f)" % (r["Kernel_Name"].split("(")[0][-40:], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
    print()
PY

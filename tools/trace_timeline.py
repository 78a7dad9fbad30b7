"""Start / end of every kernel of the last frames of a rocprofv3 kernel trace, relative to the frame's first kernel:
python3 tools/trace_timeline.py <kernel_trace.csv> [frames]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
starts = [i for i, r in enumerate(rows) if "kPool<" in r["Kernel_Name"]]
first = starts[-frames] if len(starts) >= frames else 0
t0 = int(rows[first]["Start_Timestamp"])
prev_end = t0
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-34s stream %-3s start %10.3f  end %10.3f  (%8.3f ms)" % (r["Kernel_Name"].split("(")[0][-34:], r.get("Stream_Id", "?"), (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))

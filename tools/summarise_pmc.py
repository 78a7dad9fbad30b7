#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes written by tools/gpu_pmc.sh.

    python tools/summarise_pmc.py <tag> [<tag> ...]  ->  JSON on stdout

Reads gpurun_out/pmc_<tag>/*/**/*counter_collection.csv, sums every counter over the dispatches
of the production kernel (rt::kPool, one launch per pass; RTAMD_PMC_KERNEL names another), and derives: HBM bytes per
launch (FETCH_SIZE doubled — gfx950 reports half of wide reads — plus WRITE_SIZE, both in KiB),
L2 hit rate, VALU lane utilisation and the share of cycles the SIMDs issue VALU work.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = os.environ.get("RTAMD_PMC_KERNEL", "rt::kPool<")


def summarise(tag):
    acc = collections.defaultdict(float)
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}", "*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"])
    d = dict(sorted(acc.items()))
    der = {}
    if "FETCH_SIZE" in d:
        der["hbm_bytes_per_launch_corrected"] = (2.0 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0.0)) * 1024.0
    if d.get("TCC_REQ_sum"):
        der["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    if d.get("SQ_ACTIVE_INST_VALU"):
        # SQ_THREAD_CYCLES_VALU counts active lanes x quad-cycles; a full wave is 64 lanes
        der["valu_lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / (64.0 * d["SQ_ACTIVE_INST_VALU"])
        der["wave_cycles_waiting_frac"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
    if d.get("GRBM_GUI_ACTIVE") and d.get("SQ_ACTIVE_INST_VALU"):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs issue one wave instruction per 4 cycles
        cycles = d["GRBM_GUI_ACTIVE"] / 8.0
        der["simd_valu_issue_frac"] = d["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles / 4.0)
        der["l1_accesses_per_cu_cycle"] = d.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / (256.0 * cycles)
    der["note"] = ("one launch of " + KERNEL.rstrip("(<") + " (bench.py --steps 1 --warmup 0); separate rocprofv3 --pmc passes; "
                   "FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as read")
    d["derived"] = der
    return d


def step_record(tag, workload, width, height, spp, source):
    """The figures bench.py reads back (profiles/pmc_step.json) for one workload: per step of the bench,
    i.e. all launches of the production kernel in one render."""
    d = summarise(tag)
    der = d["derived"]
    rec = {"workload": workload, "width": width, "height": height, "spp": spp, "source": source}
    if "hbm_bytes_per_launch_corrected" in der:
        rec["hbm_bytes_per_step"] = der["hbm_bytes_per_launch_corrected"]
    if "SQ_INSTS_VALU" in d:
        rec["valu_wave_insts_per_step"] = d["SQ_INSTS_VALU"]
        if d.get("SQ_INSTS_SALU"):
            rec["salu_per_valu"] = d["SQ_INSTS_SALU"] / d["SQ_INSTS_VALU"]
    if "TCP_TCC_READ_REQ_sum" in d:
        # L1 -> L2 read requests are whole 128-byte lines on this part (a counter pass over tools/l2_stream,
        # whose byte count is known, gives 128.0 bytes per request); write requests carry what one store
        # instruction wrote to a line (16 bytes per lane here) and are listed as a count
        rec["l2_read_bytes_per_step"] = 128.0 * d["TCP_TCC_READ_REQ_sum"]
        rec["l2_write_requests_per_step"] = d.get("TCP_TCC_WRITE_REQ_sum", 0.0)
    for k in ("simd_valu_issue_frac", "valu_lane_utilisation", "l2_hit_rate", "l1_accesses_per_cu_cycle", "wave_cycles_waiting_frac"):
        if k in der:
            rec[k] = der[k]
    return rec


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--step":
        # --step <out.json> <source note> tag:workload:width:height:spp ...
        out, source = sys.argv[2], sys.argv[3]
        recs = []
        for spec in sys.argv[4:]:
            tag, wl, w, h, spp = spec.split(":")
            recs.append(step_record(tag, wl, int(w), int(h), int(spp), source))
        sys.path.insert(0, ROOT)
        from raytracing_amd import build as B
        doc = {"kernel_source_sha": B.source_sha(),
               "what": "hardware-counter totals of one bench step per workload; kernel_source_sha is the fingerprint of the device "
                       "sources they were measured on (rtamd_version of the library prints the same when it was built from them)",
               "workloads": recs}
        json.dump(doc, open(out, "w"), indent=1)
        print(json.dumps(doc, indent=1))
    else:
        print(json.dumps({t: summarise(t) for t in sys.argv[1:]}, indent=1))

"""After tools/measure.sh <tag> has run on the GPU box: copy what is kept of that session into profiles/ —
counter summary, the per-step record bench.py reads back (stamped with the fingerprint of the device sources),
kernel statistics (names cut to 200 characters: rocPRIM's run to kilobytes), phase lanes, kernel resources.
    python tools/stamp_profiles.py <tag>"""
import csv
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
P = os.path.join(ROOT, "profiles")
G = os.path.join(ROOT, "gpurun_out")
py = sys.executable
tags = [f"{tag}_{w}" for w in ("mesh", "spheres", "instanced")]
with open(os.path.join(P, f"{tag}_pmc_summary.json"), "w") as f:
    subprocess.run([py, os.path.join(ROOT, "tools", "summarise_pmc.py")] + tags, stdout=f, check=True)
shutil.copy(os.path.join(P, f"{tag}_pmc_summary.json"), os.path.join(P, "r03_pmc_summary.json"))
subprocess.run([py, os.path.join(ROOT, "tools", "summarise_pmc.py"), "--step", os.path.join(P, "pmc_step.json"),
                f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc passes over one bench step, tools/gpu_pmc.sh)",
                f"{tag}_mesh:mesh:1920:1080:256", f"{tag}_spheres:spheres:1920:1080:64", f"{tag}_instanced:instanced:1920:1080:64"],
               stdout=subprocess.DEVNULL, check=True)
for wl in ("mesh", "spheres", "instanced"):
    rows = list(csv.reader(open(os.path.join(G, f"prof_{tag}_{wl}", f"{tag}_{wl}_kernel_stats.csv"))))
    rows = [rows[0]] + [[r[0] if len(r[0]) <= 200 else r[0][:197] + "..."] + r[1:] for r in rows[1:]]
    csv.writer(open(os.path.join(P, f"{tag}_{wl}_kernel_stats.csv"), "w"), quoting=csv.QUOTE_NONNUMERIC).writerows(rows)
shutil.copy(os.path.join(G, f"{tag}_phase_lanes.txt"), os.path.join(P, f"{tag}_phase_lanes.txt"))
subprocess.run([py, os.path.join(ROOT, "tools", "kernel_resources.py"), "--json", os.path.join(P, "r03_kernel_resources.json")],
               stdout=subprocess.DEVNULL, check=True)
sys.path.insert(0, ROOT)
from raytracing_amd import build as B  # noqa: E402
print("profiles stamped with", B.source_sha())

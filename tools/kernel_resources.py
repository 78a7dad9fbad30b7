#!/usr/bin/env python3
"""Registers, LDS, scratch and spill counts of every kernel in a built librtamd*.so, from the code
object's own notes (no GPU needed), with the waves per SIMD those figures allow on gfx950.

  python tools/kernel_resources.py [library] [--json out.json]

The library's .hip_fatbin section is a clang offload bundle; the gfx950 entry is an ELF code object
whose NT_AMDGPU_METADATA note lists, per kernel, .vgpr_count, .agpr_count, .sgpr_count,
.group_segment_fixed_size (LDS), .private_segment_fixed_size (scratch bytes per lane) and the
spill counts."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"

# gfx950 (CDNA4) per SIMD: 512 unified vector registers per lane, allocated in blocks of 8; at most 8 waves;
# 160 KiB of LDS per CU (4 SIMDs).
VGPR_TOTAL, VGPR_GRANULE, MAX_WAVES, LDS_PER_CU = 512, 8, 8, 160 * 1024


def code_objects(path):
    data = open(path, "rb").read()
    at = data.find(MAGIC)
    while at >= 0:
        n = struct.unpack_from("<Q", data, at + 24)[0]
        p = at + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx" in triple and size:
                yield triple, data[at + off:at + off + size]
        at = data.find(MAGIC, at + 1)


def kernels_of(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes)
        f.flush()
        text = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
    out = []
    cur = None
    for line in text.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key == "agpr_count" or (key == "args" and cur is None):
            pass
        if key in ("agpr_count", "args") and (cur is None or "name" in cur):
            cur = {}
            out.append(cur)
        if cur is None:
            continue
        if key in ("agpr_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
                   "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size", "wavefront_size", "kernarg_segment_size"):
            cur[key] = int(val)
        elif key == "name" and "name" not in cur and val.startswith("_Z"):
            cur["name"] = val
    return [k for k in out if "name" in k and "vgpr_count" in k]


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return [re.sub(r"\(.*", "", x.replace("(anonymous namespace)::", "")) for x in r]
    except Exception:
        return names


def main():
    argv = sys.argv[1:]
    out_json = None
    if "--json" in argv:
        i = argv.index("--json")
        out_json = argv[i + 1]
        del argv[i:i + 2]
    lib = argv[0] if argv else os.path.join(ROOT, "raytracing_amd", "lib", "librtamd.so")
    sha = None
    m = re.search(rb"rtamd [0-9.]+ gfx950 src:([0-9a-f]+)", open(lib, "rb").read())
    if m:
        sha = m.group(1).decode()
    result = {"library": os.path.relpath(lib, ROOT), "kernel_source_sha": sha,
              "limits": {"vgpr_per_simd_lane": VGPR_TOTAL, "max_waves_per_simd": MAX_WAVES, "lds_bytes_per_cu": LDS_PER_CU},
              "targets": []}
    for triple, elf in code_objects(lib):
        ks = kernels_of(elf)
        short = demangle([k["name"] for k in ks])
        rows = []
        for k, s in zip(ks, short):
            if "rt::" not in s:
                continue        # (rocPRIM's sort kernels of the top-level tree build)
            regs = k["vgpr_count"] + k.get("agpr_count", 0)
            alloc = -(-max(regs, 1) // VGPR_GRANULE) * VGPR_GRANULE
            by_regs = min(MAX_WAVES, VGPR_TOTAL // alloc)
            wg = k.get("max_flat_workgroup_size", 256)
            waves_per_wg = -(-wg // 64)
            lds = k.get("group_segment_fixed_size", 0)
            wgs_by_lds = LDS_PER_CU // lds if lds else 99
            # workgroups per CU: limited by registers (waves per SIMD x 4 SIMDs) and by LDS
            wgs = min(by_regs * 4 // waves_per_wg if waves_per_wg else 0, wgs_by_lds)
            rows.append({"kernel": s, "vgpr": k["vgpr_count"], "agpr": k.get("agpr_count", 0), "sgpr": k["sgpr_count"],
                         "lds_bytes": lds, "scratch_bytes_per_lane": k.get("private_segment_fixed_size", 0),
                         "vgpr_spills": k.get("vgpr_spill_count", 0), "sgpr_spills": k.get("sgpr_spill_count", 0),
                         "workgroup": wg, "waves_per_simd_by_registers": by_regs,
                         "workgroups_per_cu": wgs, "waves_per_simd_resident": wgs * waves_per_wg / 4.0})
        result["targets"].append({"target": triple, "kernels": rows})
        print(f"{triple}: {len(rows)} kernels")
        for r in rows:
            print(f"  {r['kernel'][:64]:64s} vgpr {r['vgpr']:3d} sgpr {r['sgpr']:3d} lds {r['lds_bytes']:6d} scratch {r['scratch_bytes_per_lane']:4d} "
                  f"spills v{r['vgpr_spills']} s{r['sgpr_spills']}  wg {r['workgroup']:4d}  waves/SIMD {r['waves_per_simd_resident']:.0f}")
    if out_json:
        with open(out_json, "w") as f:
            json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()

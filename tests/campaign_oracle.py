#!/usr/bin/env python3
"""Randomised full-size parity campaign against the CPU oracle (a script, not collected by pytest: it takes minutes
on the GPU box's 16 host cores).  The production kernels, through the C ABI, against oracle/ on the same scenes and
random cameras: primary ids, t, radiance and ray counts of every pixel, tolerance 0.

    python tests/campaign_oracle.py [--mode pt|ref] [--cameras N] [--spp S] [--workloads spheres,mesh,instanced,glass,soup]

Prints one line per frame and a total; exit status 1 if anything differs."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from raytracing_amd import backend, scenes, types as T  # noqa: E402


REF_MODE = False      # --mode ref: what the reference shader renders (primary + mirror ray) instead of the path tracer


def compare(rt, wl, ci, buf, frame, sky, w, h, spp, flags, threads):
    kw = dict(spp=spp, max_bounces=8, seed=3000 + ci) if not REF_MODE else {}
    t0 = time.time()
    g = rt.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, mode=T.MODE_REF if REF_MODE else T.MODE_PT, flags=flags, **kw)
    t1 = time.time()
    o = O.render(buf, frame, w, h, sky=sky, mode=0 if REF_MODE else 1, nthreads=threads, flags=flags, **kw)
    t2 = time.time()
    ids = int((g["ids"] != o["ids"]).any(-1).sum())
    tt = int(((g["t"] != o["t"]) & ~(np.isnan(g["t"]) & np.isnan(o["t"]))).sum())
    rgba = int(((g["rgba"] != o["rgba"]) & ~(np.isnan(g["rgba"]) & np.isnan(o["rgba"]))).any(-1).sum())
    rays = g["rays"] != o["counters"]["rays"]
    cover = float((o["ids"][..., 0] >= 0).mean())
    print(f"{wl} camera {ci}: coverage {cover:.2f}, {o['counters']['rays']} segments, {g['handover_pixels']} handed over, GPU {t1 - t0:.2f} s, oracle {t2 - t1:.1f} s; "
          f"mismatching ids {ids}, t {tt}, radiance pixels {rgba}, ray count {'DIFFERS' if rays else 'equal'}", flush=True)
    return ids + tt + rgba + int(rays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cameras", type=int, default=6)
    ap.add_argument("--spp", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--workloads", default="spheres,mesh")
    ap.add_argument("--mesh-tris", type=int, default=1_000_000)
    ap.add_argument("--seed", type=int, default=99)
    ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 1, 16))
    ap.add_argument("--mode", default="pt", choices=["pt", "ref"])
    args = ap.parse_args()
    global REF_MODE
    REF_MODE = args.mode == "ref"
    w, h = args.width, args.height
    O.lib()
    rt = backend.RayTracingPass(0)
    rs = np.random.RandomState(args.seed)
    bad_total = 0
    for wl in args.workloads.split(","):
        flags = T.FLAG_DIELECTRICS if wl == "glass" else 0
        if wl == "soup":
            for ci in range(args.cameras):
                buf, frame, sky = scenes.random_soup(w, h, seed=1000 + args.seed + ci)
                rt.SetScene(buf)
                rt.SetSkyBox(sky)
                bad_total += compare(rt, wl, ci, buf, frame, sky, w, h, args.spp, flags, args.threads)
            continue
        if wl == "spheres":
            buf, frame0, sky = scenes.random_spheres(w, h)
            centre, radius = np.array([0.0, 0.5, 0.0]), 18.0
        elif wl == "mesh":
            buf, frame0, sky = scenes.mesh_scene(w, h, target_tris=args.mesh_tris)
            centre, radius = np.array([0.0, 1.0, 0.0]), 6.0
        else:
            buf, frame0, sky = scenes.instanced_scene(w, h, instances=1250, tris_per_blas=2000, glass_every=3 if wl == "glass" else 0)
            centre, radius = np.array([0.0, 1.0, 0.0]), 40.0
        rt.SetScene(buf)
        rt.SetSkyBox(sky)
        for ci in range(args.cameras):
            ang = rs.uniform(0, 2 * np.pi)
            eye = centre + radius * rs.uniform(0.3, 1.2) * np.array([np.cos(ang), 0.0, np.sin(ang)]) + np.array([0, rs.uniform(0.3, 8.0), 0])
            tgt = centre + rs.uniform(-2, 2, size=3) * np.array([1, 0.2, 1])
            frame = frame0 if ci == 0 else scenes.look_at_camera(tuple(eye), tuple(tgt), float(rs.uniform(25, 70)), w, h)
            bad_total += compare(rt, wl, ci, buf, frame, sky, w, h, args.spp, flags, args.threads)
    print("total mismatches", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())

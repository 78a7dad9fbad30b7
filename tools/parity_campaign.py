#!/usr/bin/env python3
"""Randomised full-size parity campaign on the GPU: the production kernel against the
reference-order kernel (same library, flag RTAMD_FLAG_REFERENCE_ORDER) over many cameras.

    python tools/parity_campaign.py [--cameras N] [--spp S] [--workloads spheres,mesh,instanced]

Every pixel of every frame must agree bit for bit (ids, t, radiance, ray count).  Prints one line
per frame and a total; exit status 1 if anything differs.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import backend, scenes, types as T  # noqa: E402


def compare(rt, wl, ci, frame, w, h, spp, flags=0):
    kw = dict(mode=T.MODE_PT, spp=spp, max_bounces=8, seed=1000 + ci)
    a = rt.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, flags=flags, **kw)
    b = rt.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, flags=flags | T.FLAG_REFERENCE_ORDER, **kw)
    ids = int((a["ids"] != b["ids"]).any(-1).sum())
    tt = int(((a["t"] != b["t"]) & ~(np.isnan(a["t"]) & np.isnan(b["t"]))).sum())
    rgba = int(((a["rgba"] != b["rgba"]) & ~(np.isnan(a["rgba"]) & np.isnan(b["rgba"]))).any(-1).sum())
    rays = a["rays"] != b["rays"]
    cover = float((b["ids"][..., 0] >= 0).mean())
    print(f"{wl} camera {ci}: coverage {cover:.2f}, {b['rays']} segments, {a['handover_pixels']} pixels handed over; "
          f"mismatching ids {ids}, t {tt}, radiance pixels {rgba}, ray count {'DIFFERS' if rays else 'equal'}", flush=True)
    return ids + tt + rgba + int(rays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cameras", type=int, default=6)
    ap.add_argument("--spp", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--workloads", default="spheres,mesh")
    ap.add_argument("--mesh-tris", type=int, default=200_000, help="triangle count of the mesh workload (1 000 000: the large-scene variant of the pool kernel)")
    ap.add_argument("--seed", type=int, default=2025, help="seed of the random cameras (and offset of the soups' seeds)")
    args = ap.parse_args()
    w, h = args.width, args.height
    rt = backend.RayTracingPass(0)
    rs = np.random.RandomState(args.seed)
    bad_total = 0
    for wl in args.workloads.split(","):
        if wl == "soup":
            # a new random scene per "camera"
            for ci in range(args.cameras):
                buf, frame, sky = scenes.random_soup(w, h, seed=ci + abs(args.seed - 2025))
                rt.SetScene(buf)
                rt.SetSkyBox(sky)
                bad_total += compare(rt, wl, ci, frame, w, h, args.spp)
            continue
        if wl == "spheres":
            buf, frame0, sky = scenes.random_spheres(w, h)
            centre, radius = np.array([0.0, 0.5, 0.0]), 18.0
        elif wl == "mesh":
            buf, frame0, sky = scenes.mesh_scene(w, h, target_tris=args.mesh_tris)
            centre, radius = np.array([0.0, 1.0, 0.0]), 6.0
        elif wl == "glass":
            # the "diffuse + dielectric" mix: every third instance is tinted glass (the dielectric variant of the kernels)
            buf, frame0, sky = scenes.instanced_scene(w, h, instances=1250, tris_per_blas=2000, glass_every=3)
            centre, radius = np.array([0.0, 1.0, 0.0]), 40.0
        else:
            buf, frame0, sky = scenes.instanced_scene(w, h, instances=1250, tris_per_blas=2000)
            centre, radius = np.array([0.0, 1.0, 0.0]), 40.0
        rt.SetScene(buf)
        rt.SetSkyBox(sky)
        for ci in range(args.cameras):
            if ci == 0:
                frame = frame0
            else:
                ang = rs.uniform(0, 2 * np.pi)
                eye = centre + radius * rs.uniform(0.3, 1.2) * np.array([np.cos(ang), 0.0, np.sin(ang)]) + np.array([0, rs.uniform(0.3, 8.0), 0])
                tgt = centre + rs.uniform(-2, 2, size=3) * np.array([1, 0.2, 1])
                frame = scenes.look_at_camera(tuple(eye), tuple(tgt), float(rs.uniform(25, 70)), w, h)
            bad_total += compare(rt, wl, ci, frame, w, h, args.spp, T.FLAG_DIELECTRICS if wl == "glass" else 0)
    print("total mismatches", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())

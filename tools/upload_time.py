"""Scene upload with the meshes' trees built by the host (binned SAH) and by the GPU (Morton order): time of
rtamd_scene_upload, its tree-building part, and what the choice costs when tracing (PT, the bench's mesh scene).
    python tools/upload_time.py [--tris 1000000] [--spp 32]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401  (before the library is loaded: torch brings its own HIP runtime, and of two runtimes in one process its own sees no GPU)

from raytracing_amd import backend, scenes, types as T  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=1_000_000)
ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--scene", default="mesh", choices=["mesh", "instanced", "spheres"])
ap.add_argument("--procedural", action="store_true", help="mesh scene from displaced spheres of any triangle count instead of the reference's models")
a = ap.parse_args()
w, h = 1920, 1080
if a.scene == "mesh":
    buf, frame, sky = scenes.mesh_scene(w, h, target_tris=a.tris, procedural=a.procedural)
elif a.scene == "instanced":
    buf, frame, sky = scenes.instanced_scene(w, h)
else:
    buf, frame, sky = scenes.random_spheres(w, h)
print(f"scene {a.scene}: {len(buf.triangles)} triangles in {len(buf.blas_offsets)} meshes, {len(buf.objects)} instances", flush=True)
p = backend.RayTracingPass(0)
p.SetSkyBox(sky)
rgba = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
ids = {}
for who in ("host", "device", "host", "device"):
    p.SetBlasBuild(who)
    t0 = time.perf_counter()
    p.SetScene(buf)
    wall = (time.perf_counter() - t0) * 1e3
    st = p.UploadStats()
    i2 = torch.zeros((h, w, 2), dtype=torch.int32, device="cuda:0")
    p.Execute(frame, w, h, out_rgba=rgba, out_prim_id=i2, mode=T.MODE_PT, spp=a.spp, max_bounces=8, seed=1)
    best = None
    for _ in range(3):
        r = p.Execute(frame, w, h, out_rgba=rgba, mode=T.MODE_PT, spp=a.spp, max_bounces=8, seed=1)
        best = r if best is None or r["kernel_ms"] < best["kernel_ms"] else best
    ref = min(p.Execute(frame, w, h, out_rgba=rgba, mode=T.MODE_REF)["kernel_ms"] for _ in range(5))
    ids[who] = (i2.cpu().numpy(), rgba.cpu().numpy().copy())
    print(f"{who:6s} upload {st['upload_ms']:8.1f} ms (python wall {wall:8.1f})  trees {st['tree_build_ms']:8.1f} ms  on device {st['meshes_built_on_device']}"
          f"  PT {best['rays'] / best['kernel_ms'] / 1e3:8.1f} Mrays/s ({best['kernel_ms']:.2f} ms)  REF {ref:.3f} ms", flush=True)
print("ids equal:", np.array_equal(ids["host"][0], ids["device"][0]))

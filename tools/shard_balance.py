"""Load balance of tile sharding, measured on one GPU: renders each of N interleaved shards of the
frame on its own and compares the slowest with the mean (what N GPUs would see, the all-reduce
aside).  Usage: python tools/shard_balance.py [--4k] [N ...]   (--4k: the several-GPU configuration, the mesh scene at
3840x2160 / 1024 spp, wall-clock time of the whole call)"""
import sys
import time

import torch

sys.path.insert(0, ".")
from raytracing_amd import backend, scenes, types as T

big = "--4k" in sys.argv
counts = [int(a) for a in sys.argv[1:] if a != "--4k"] or [2, 4, 8]
w, h = (3840, 2160) if big else (1920, 1080)
spp = 1024 if big else 64


def timed(rt, *a, **k):
    t0 = time.perf_counter()
    r = rt.Execute(*a, **k)
    return (time.perf_counter() - t0) * 1e3 if big else r["kernel_ms"]


for name, mk in ((("mesh", scenes.mesh_scene),) if big else (("spheres", scenes.random_spheres), ("mesh", scenes.mesh_scene))):
    buf, frame, sky = mk(w, h)
    torch.cuda.init()
    rt = backend.RayTracingPass(0)
    rt.SetScene(buf)
    rt.SetSkyBox(sky)
    out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    kw = dict(mode=T.MODE_PT, spp=spp, max_bounces=8, seed=3, out_rgba=out)
    rt.Execute(frame, w, h, shard_rank=0, shard_count=8, **kw)
    whole = timed(rt, frame, w, h, **kw)
    for n in counts:
        ms = [timed(rt, frame, w, h, shard_rank=r, shard_count=n, **kw) for r in range(n)]
        print("%-8s %d shards: whole frame %.1f ms, shards %.1f .. %.1f ms, speed-up of the slowest %.2fx (%.0f %% of %d)" % (
            name, n, whole, min(ms), max(ms), whole / max(ms), 100.0 * whole / max(ms) / n, n), flush=True)
    rt.Dispose()

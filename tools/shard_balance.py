"""Load balance of tile sharding, measured on one GPU: renders each of N interleaved shards of the
frame on its own and compares the slowest with the mean (what N GPUs would see, the all-reduce
aside).  Usage: python tools/shard_balance.py [N ...]"""
import sys

sys.path.insert(0, ".")
from raytracing_amd import backend, scenes, types as T

counts = [int(a) for a in sys.argv[1:]] or [2, 4, 8]
w, h = 1920, 1080
for name, mk in (("spheres", scenes.random_spheres), ("mesh", scenes.mesh_scene)):
    buf, frame, sky = mk(w, h)
    rt = backend.RayTracingPass(0)
    rt.SetScene(buf)
    rt.SetSkyBox(sky)
    kw = dict(mode=T.MODE_PT, spp=64, max_bounces=8, seed=3)
    rt.Execute(frame, w, h, **kw)
    whole = rt.Execute(frame, w, h, **kw)["kernel_ms"]
    for n in counts:
        ms = [rt.Execute(frame, w, h, shard_rank=r, shard_count=n, **kw)["kernel_ms"] for r in range(n)]
        print("%-8s %d shards: whole frame %.1f ms, shards %.1f .. %.1f ms, speed-up of the slowest %.2fx (%.0f %% of %d)" % (
            name, n, whole, min(ms), max(ms), whole / max(ms), 100.0 * whole / max(ms) / n, n), flush=True)
    rt.Dispose()

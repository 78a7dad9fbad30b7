"""Time of rtamd_scene_refresh_objects (validation + walk of the caller's TLAS + rebuild of the
library's private tree + upload) with the tree built on the host and on the GPU, and the cost of a
1 spp frame on either tree.  Usage: python tools/refresh_time.py [instances ...]"""
import sys
import time

sys.path.insert(0, ".")
from raytracing_amd import backend, scenes, types as T

scattered = "--scattered" in sys.argv
counts = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [484, 1250, 5000, 20000, 100000]
w, h = 1920, 1080
for n in counts:
    if scattered:
        buf, frame, sky = scenes.scattered_scene(w, h, instances=n)
    elif n == 484:
        buf, frame, sky = scenes.random_spheres(w, h)
    else:
        buf, frame, sky = scenes.instanced_scene(w, h, instances=n, tris_per_blas=2000, tlas_tree=1)
    rt = backend.RayTracingPass(0)
    rt.SetSkyBox(sky)
    line = "%7d objects:" % len(buf.objects)
    for mode in ("host", "device"):
        rt.SetTlasBuild(mode)
        rt.SetScene(buf)
        reps = 20 if n <= 5000 else 5
        t1 = time.perf_counter()
        for _ in range(reps):
            rt.RefreshObjects(buf)
        t2 = time.perf_counter()
        info = rt.TlasInfo()
        rt.Execute(frame, w, h, mode=T.MODE_PT, spp=4, max_bounces=8, seed=1)
        r = rt.Execute(frame, w, h, mode=T.MODE_PT, spp=4, max_bounces=8, seed=1)
        line += "  %s refresh %.3f ms, %d nodes / %d levels, 4 spp frame %.2f ms (%.0f Mrays/s);" % (
            mode, (t2 - t1) / reps * 1e3, info["nodes"], info["levels"], r["kernel_ms"], r["rays"] / r["kernel_ms"] / 1e3)
    print(line, flush=True)
    rt.Dispose()

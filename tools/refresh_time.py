import time, sys
sys.path.insert(0, ".")
from raytracing_amd import backend, scenes
for name, mk in (("spheres", lambda: scenes.random_spheres(64, 36)), ("instanced", lambda: scenes.instanced_scene(64, 36, tris_per_blas=2000))):
    buf, frame, sky = mk()
    rt = backend.RayTracingPass(0)
    t0 = time.perf_counter(); rt.SetScene(buf); t1 = time.perf_counter()
    n = 20
    for _ in range(n): rt.RefreshObjects(buf)
    t2 = time.perf_counter()
    print(name, len(buf.objects), "objects: SetScene %.1f ms, RefreshObjects %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) / n * 1e3))

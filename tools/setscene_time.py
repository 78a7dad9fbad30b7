import time, sys
sys.path.insert(0, ".")
from raytracing_amd import backend, scenes
t0=time.perf_counter()
buf, frame, sky = scenes.mesh_scene(1920, 1080)
t1=time.perf_counter()
print("scene gen %.2f s, tris %d" % (t1-t0, len(buf.triangles)))
rt = backend.RayTracingPass(0)
for i in range(3):
    t0=time.perf_counter(); rt.SetScene(buf); t1=time.perf_counter()
    print("SetScene %.1f ms" % ((t1-t0)*1e3))

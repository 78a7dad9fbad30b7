"""Time of a REF-mode frame (what the reference shader renders: primary ray + one mirror ray) on the
benchmark scenes, production kernel and reference-order kernel.  Usage: python tools/ref_mode_time.py"""
import sys

sys.path.insert(0, ".")
from raytracing_amd import backend, scenes, types as T

for name, mk, (w, h) in (("spheres", scenes.random_spheres, (1920, 1080)), ("mesh", scenes.mesh_scene, (1920, 1080)),
                         ("mesh 4K", scenes.mesh_scene, (3840, 2160))):
    buf, frame, sky = mk(w, h)
    rt = backend.RayTracingPass(0)
    rt.SetScene(buf)
    rt.SetSkyBox(sky)
    line = "%-8s %dx%d:" % (name, w, h)
    for label, flags in (("production", 0), ("reference order", T.FLAG_REFERENCE_ORDER)):
        rt.Execute(frame, w, h, mode=T.MODE_REF, flags=flags)
        best = min((rt.Execute(frame, w, h, mode=T.MODE_REF, flags=flags) for _ in range(5)), key=lambda r: r["kernel_ms"])
        line += "  %s %.3f ms (%.0f Mrays/s, %d segments, %d pixels handed over);" % (
            label, best["kernel_ms"], best["rays"] / best["kernel_ms"] / 1e3, best["rays"], best["handover_pixels"])
    print(line, flush=True)
    rt.Dispose()

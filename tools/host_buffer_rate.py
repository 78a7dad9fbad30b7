"""Whole-call time of rtamd_render with the framebuffer in host memory against a device framebuffer
(bench scene, 1920x1080, 64 spp).  Usage: python tools/host_buffer_rate.py"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
from raytracing_amd import backend, scenes, types as T

w, h = 1920, 1080
buf, frame, sky = scenes.random_spheres(w, h)
rt = backend.RayTracingPass(0)
rt.SetScene(buf)
rt.SetSkyBox(sky)
kw = dict(mode=T.MODE_PT, spp=64, max_bounces=8, seed=1)
dev = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
host = np.zeros((h, w, 4), dtype=np.float32)
for name, out in (("device framebuffer", dev), ("host framebuffer", host)):
    rt.Execute(frame, w, h, out_rgba=out, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        r = rt.Execute(frame, w, h, out_rgba=out, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%s: %.2f ms per call, %.0f Mrays/s" % (name, dt * 1e3, r["rays"] / dt / 1e6), flush=True)

"""Wall-clock time of one REF-mode frame through the C ABI (what the reference's per-frame dispatch costs a host
that calls rtamd_render once per camera per frame): device-memory output, 200 frames back to back.
Usage: python tools/ref_frame_latency.py [frames]"""
import sys
import time

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from raytracing_amd import backend, scenes, types as T

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, mk, (w, h) in (("spheres", scenes.random_spheres, (1920, 1080)), ("mesh", scenes.mesh_scene, (1920, 1080)),
                         ("mesh 4K", scenes.mesh_scene, (3840, 2160))):
    buf, frame, sky = mk(w, h)
    rt = backend.RayTracingPass(0)
    rt.SetStream(torch.cuda.current_stream().cuda_stream)
    rt.SetScene(buf)
    rt.SetSkyBox(sky)
    out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    for _ in range(5):
        r = rt.Execute(frame, w, h, out_rgba=out, mode=T.MODE_REF)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    km = hm = 0.0
    for _ in range(frames):
        r = rt.Execute(frame, w, h, out_rgba=out, mode=T.MODE_REF)
        km += r["kernel_ms"]
        hm += r["handover_ms"]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames * 1e3
    print("%-8s %dx%d REF: %.3f ms per call (kernels %.3f ms of which hand-over %.3f ms, %d pixels handed over, %d segments)"
          % (name, w, h, dt, km / frames, hm / frames, r["handover_pixels"], r["rays"]), flush=True)
    rt.Dispose()

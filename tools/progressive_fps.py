"""Progressive rendering at low sample counts: 1080p frames per second of `spp` new samples per frame added to the
running mean (RTAMD_FLAG_ACCUMULATE), frame in device memory -- the reference's per-frame pass
(Assets/Script/RenderPass/RayTracingPass.cs:62-109) used the way a viewer uses it.  Blocking calls
(rtamd_render) against calls that return before the frame is complete (rtamd_render_begin / rtamd_render_wait,
two frames in flight) when the library has them.
Usage: python tools/progressive_fps.py [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from raytracing_amd import backend, scenes, types as T

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
w, h = 1920, 1080
for name, mk in (("spheres", scenes.random_spheres), ("mesh", scenes.mesh_scene)):
    buf, frame, sky = mk(w, h)
    rt = backend.RayTracingPass(0)
    rt.SetStream(torch.cuda.current_stream().cuda_stream)
    rt.SetScene(buf)
    rt.SetSkyBox(sky)
    out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    out2 = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    for spp in (1, 4, 8):
        kw = dict(mode=T.MODE_PT, spp=spp, max_bounces=8, seed=7, flags=T.FLAG_ACCUMULATE)
        rt.ResetAccumulation()
        for i in range(5):
            rt.Execute(frame, w, h, out_rgba=out, sample_offset=i * spp, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rays = 0
        for i in range(frames):
            rays += rt.Execute(frame, w, h, out_rgba=out, sample_offset=(5 + i) * spp, **kw)["rays"]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        line = "%-8s %2d spp per frame: blocking %.1f frames/s (%.3f ms, %.0f Mrays/s)" % (name, spp, frames / dt, dt / frames * 1e3, rays / dt / 1e6)
        if hasattr(rt, "ExecuteBegin"):
            rt.ResetAccumulation()
            bufs = (out, out2)
            for i in range(5):
                rt.Execute(frame, w, h, out_rgba=out, sample_offset=i * spp, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rays = 0
            rt.ExecuteBegin(frame, w, h, out_rgba=bufs[0], sample_offset=5 * spp, **kw)
            for i in range(1, frames):
                rt.ExecuteBegin(frame, w, h, out_rgba=bufs[i & 1], sample_offset=(5 + i) * spp, **kw)
                rays += rt.ExecuteWait()["rays"]
            rays += rt.ExecuteWait()["rays"]
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            line += "; begin / wait %.1f frames/s (%.3f ms, %.0f Mrays/s)" % (frames / dt2, dt2 / frames * 1e3, rays / dt2 / 1e6)
        print(line, flush=True)
    rt.Dispose()

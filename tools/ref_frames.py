"""REF-mode frames only (for a kernel trace of the reference shader's own image): python tools/ref_frames.py [mesh|spheres] [width height] [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from raytracing_amd import backend, scenes, types as T  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mesh"
w = int(sys.argv[2]) if len(sys.argv) > 3 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 50
buf, frame, sky = (scenes.mesh_scene if name == "mesh" else scenes.random_spheres)(w, h)
p = backend.RayTracingPass(0)
p.SetScene(buf)
p.SetSkyBox(sky)
out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
for _ in range(5):
    p.Execute(frame, w, h, out_rgba=out, mode=T.MODE_REF)
torch.cuda.synchronize()
t0 = time.perf_counter()
km = 0.0
for _ in range(frames):
    km += p.Execute(frame, w, h, out_rgba=out, mode=T.MODE_REF)["kernel_ms"]
torch.cuda.synchronize()
print(f"{name} {w}x{h}: {(time.perf_counter() - t0) / frames * 1e3:.3f} ms per call, {km / frames:.3f} ms between events")

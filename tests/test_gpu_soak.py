"""A context used the way a host uses it for a long time — uploads with either tree builder, per-frame object
refreshes (BuildAccelerateStructureSystem.cs:148-158 re-uploads every frame), REF and PT frames, frames in flight,
the running sum — must neither grow its device memory nor change its answers."""
import numpy as np
import pytest

from raytracing_amd import types as T

pytestmark = pytest.mark.gpu


def test_hundreds_of_frames_and_uploads_hold_memory_and_answers():
    import torch
    from raytracing_amd import backend, scenes
    w, h = 320, 180
    soup = scenes.random_soup(w, h, seed=21, meshes=4, tris_per_mesh=3000, instances=25)
    spheres = scenes.random_spheres(w, h, grid=6)
    p = backend.RayTracingPass(0)
    try:
        rgba = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
        rgba2 = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
        ids = torch.zeros((h, w, 2), dtype=torch.int32, device="cuda:0")

        def cycle(k):
            buf, frame, sky = soup if k % 2 == 0 else spheres
            p.SetBlasBuild("device" if k % 4 >= 2 else "host")
            p.SetScene(buf)
            p.SetSkyBox(sky)
            out = []
            p.Execute(frame, w, h, out_rgba=rgba, out_prim_id=ids, mode=T.MODE_REF)
            out.append((rgba.cpu().numpy().copy(), ids.cpu().numpy().copy()))
            for _ in range(3):
                p.RefreshObjects(buf)
                p.Execute(frame, w, h, out_rgba=rgba, mode=T.MODE_PT, spp=2, max_bounces=4, seed=5)
            out.append(rgba.cpu().numpy().copy())
            p.ResetAccumulation()
            p.ExecuteBegin(frame, w, h, out_rgba=rgba, mode=T.MODE_PT, spp=1, max_bounces=4, seed=5, flags=T.FLAG_ACCUMULATE)
            p.ExecuteBegin(frame, w, h, out_rgba=rgba2, mode=T.MODE_PT, spp=1, max_bounces=4, seed=5, sample_offset=1, flags=T.FLAG_ACCUMULATE)
            p.ExecuteWait()
            p.ExecuteWait()
            out.append(rgba2.cpu().numpy().copy())
            return out

        first = [cycle(k) for k in range(4)]          # every combination once: buffers reach their sizes
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info(0)
        for k in range(4, 64):
            got = cycle(k)
            want = first[k % 4]
            assert np.array_equal(got[0][0], want[0][0], equal_nan=True) and np.array_equal(got[0][1], want[0][1])
            assert np.array_equal(got[1], want[1], equal_nan=True)
            assert np.array_equal(got[2], want[2], equal_nan=True)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info(0)
        assert free0 - free1 < 64 << 20, (free0, free1)      # no growth beyond allocator noise over 60 uploads and 480 frames
    finally:
        p.Dispose()

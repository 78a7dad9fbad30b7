"""rtamd_render_begin / rtamd_render_wait: frames that return before they are complete, two in flight, must give the
same bits as the blocking rtamd_render (RayTracingPass.cs:62-109 is a per-frame pass: this is the call pattern of a
host that renders frame after frame)."""
import numpy as np
import pytest

from raytracing_amd import types as T

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("mode", ["ref", "pt", "pt_accumulate"])
def test_frames_in_flight_equal_blocking_frames(gpu_pass, mode):
    import torch
    from raytracing_amd import backend, scenes
    w, h = 480, 270
    buf, frame, sky = scenes.random_spheres(w, h, grid=10)
    cams = [scenes.look_at_camera((0.3 * k, 4.0 + 0.2 * k, -12.0), (0.0, 0.3, -1.0), 38.0, w, h) for k in range(6)]
    p = backend.RayTracingPass(0)
    try:
        p.SetScene(buf)
        p.SetSkyBox(sky)
        gpu_pass.SetScene(buf)
        gpu_pass.SetSkyBox(sky)
        flags = T.FLAG_ACCUMULATE if mode == "pt_accumulate" else 0
        kws = [dict(mode=T.MODE_REF) if mode == "ref" else dict(mode=T.MODE_PT, spp=3, max_bounces=5, seed=11, sample_offset=3 * k, flags=flags) for k in range(6)]
        if mode == "pt_accumulate":
            cams = [cams[0]] * 6
        # blocking frames
        want = []
        gpu_pass.ResetAccumulation()
        for k in range(6):
            rgba = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
            ids = torch.full((h, w, 2), -9, dtype=torch.int32, device="cuda:0")
            r = gpu_pass.Execute(cams[k], w, h, out_rgba=rgba, out_prim_id=ids, **kws[k])
            want.append((rgba.cpu().numpy(), ids.cpu().numpy(), r["rays"], r["handover_pixels"]))
        # the same frames, two in flight, outputs double-buffered
        outs = [(torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0"), torch.full((h, w, 2), -9, dtype=torch.int32, device="cuda:0")) for _ in range(2)]
        got = []
        p.ExecuteBegin(cams[0], w, h, out_rgba=outs[0][0], out_prim_id=outs[0][1], **kws[0])
        for k in range(1, 6):
            p.ExecuteBegin(cams[k], w, h, out_rgba=outs[k & 1][0], out_prim_id=outs[k & 1][1], **kws[k])
            r = p.ExecuteWait()                       # frame k - 1
            o = outs[(k - 1) & 1]
            got.append((o[0].cpu().numpy(), o[1].cpu().numpy(), r["rays"], r["handover_pixels"]))
        r = p.ExecuteWait()
        got.append((outs[1][0].cpu().numpy(), outs[1][1].cpu().numpy(), r["rays"], r["handover_pixels"]))
        for k in range(6):
            assert _same(want[k][0], got[k][0]), (mode, k)
            assert _same(want[k][1], got[k][1]), (mode, k)
            assert want[k][2] == got[k][2] and want[k][2] > 0, (mode, k)
            assert want[k][3] == got[k][3], (mode, k)
        assert (want[0][1][..., 0] >= 0).mean() > 0.3
    finally:
        p.Dispose()


def test_begin_wait_limits(gpu_pass):
    import torch
    from raytracing_amd import backend, scenes
    w, h = 64, 36
    buf, frame, sky = scenes.few_spheres(w, h)
    p = backend.RayTracingPass(0)
    try:
        p.SetScene(buf)
        out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
        with pytest.raises(backend.BackendError, match="no frame in flight"):
            p.ExecuteWait()
        with pytest.raises(backend.BackendError, match="device memory"):
            p.ExecuteBegin(frame, w, h, out_rgba=np.zeros((h, w, 4), np.float32), mode=T.MODE_REF)
        p.ExecuteBegin(frame, w, h, out_rgba=out, mode=T.MODE_REF)
        p.ExecuteBegin(frame, w, h, out_rgba=out, mode=T.MODE_REF)
        with pytest.raises(backend.BackendError, match="two frames are in flight"):
            p.ExecuteBegin(frame, w, h, out_rgba=out, mode=T.MODE_REF)
        with pytest.raises(backend.BackendError, match="rtamd_render_wait first"):
            p.Execute(frame, w, h, out_rgba=out, mode=T.MODE_REF)
        with pytest.raises(backend.BackendError, match="rtamd_render_wait first"):
            p.SetScene(buf)
        p.ExecuteWait()
        p.ExecuteWait()
        r = p.Execute(frame, w, h, out_rgba=out, mode=T.MODE_REF)       # and the blocking call works again
        assert r["rays"] > 0
    finally:
        p.Dispose()

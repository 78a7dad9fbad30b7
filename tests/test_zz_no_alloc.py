"""The allocation-free entry point, rtamd_render_noalloc, against rtamd_render through the C ABI:
same image, ids, t and ray count bit for bit, on two calls in a row (the second on a smaller
frame, so that the fixed working array still holds the larger frame's leftovers)."""
import numpy as np
import pytest

from raytracing_amd import types as T

pytestmark = pytest.mark.gpu


def _render(p, entry, frame, w, h, **kw):
    import torch
    dev = torch.device("cuda", 0)
    rgba = torch.full((h, w, 4), 7.0, dtype=torch.float32, device=dev)
    ids = torch.full((h, w, 2), -9, dtype=torch.int32, device=dev)
    t = torch.full((h, w), -1.0, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]          # hipMemGetInfo: moves with every hipMalloc / hipFree of the process
    r = (p.ExecuteNoAlloc if entry == "noalloc" else p.Execute)(frame, w, h, out_rgba=rgba, out_prim_id=ids, out_t=t, **kw)
    torch.cuda.synchronize()
    r["free_delta"] = torch.cuda.mem_get_info()[0] - free_before
    return rgba.cpu().numpy(), ids.cpu().numpy(), t.cpu().numpy(), r


@pytest.mark.parametrize("mode", ["ref", "pt"])
def test_noalloc_entry_equals_render(mode):
    import torch
    from raytracing_amd import backend, scenes
    p = backend.RayTracingPass(0)
    try:
        sizes = [(640, 360), (200, 120)]
        kw = dict(mode=T.MODE_REF) if mode == "ref" else dict(mode=T.MODE_PT, spp=3, max_bounces=4, seed=17)
        for i, (w, h) in enumerate(sizes):
            buf, frame, sky = scenes.random_spheres(w, h, grid=8) if i == 0 else scenes.few_spheres(w, h)
            if i == 0:
                # an axis-aligned camera whose x is a bound of the floor's box: the centre column's rays (direction.x == 0)
                # cannot be settled by the production kernel and are handed to the reference-order kernel
                from raytracing_amd import host as H
                floor = [float(v) for n in buf.tlas[1:3] for v in list(n["Lx"]) + list(n["Hx"]) if np.isfinite(v) and abs(v) > 20]
                frame = H.camera((floor[0], 1.5, -9.0), (0, 0, 0, 1), 40.0, w / h)
            p.SetScene(buf)
            p.SetSkyBox(sky)
            a = _render(p, "render", frame, w, h, **kw)
            b = _render(p, "noalloc", frame, w, h, **kw)
            assert b[3]["free_delta"] == 0                # free device memory did not move
            assert np.array_equal(a[0], b[0], equal_nan=True), (mode, w, h)
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2], equal_nan=True)
            assert a[3]["rays"] == b[3]["rays"] and a[3]["rays"] > 0
            assert a[3]["handover_pixels"] == b[3]["handover_pixels"]
            if mode == "ref" and i == 0:
                assert b[3]["handover_pixels"] > 0        # the camera stands on a box bound (below): the fixed list and mask were used
            assert (a[1][..., 0] >= 0).mean() > 0.1
            # a tile rectangle and a shard: pixels outside keep the caller's values
            c = _render(p, "render", frame, w, h, tile_x=5, tile_y=3, tile_w=w // 2, tile_h=h // 2, shard_rank=1, shard_count=2, **kw)
            d = _render(p, "noalloc", frame, w, h, tile_x=5, tile_y=3, tile_w=w // 2, tile_h=h // 2, shard_rank=1, shard_count=2, **kw)
            assert np.array_equal(c[0], d[0], equal_nan=True) and np.array_equal(c[1], d[1])
            assert (d[0] == 7.0).any() and (d[0] != 7.0).any()
    finally:
        p.Dispose()


@pytest.mark.parametrize("scene", ["mesh", "instanced"])
def test_noalloc_entry_on_scenes_with_deep_traversal_stacks(scene):
    """Stacks on these scenes outgrow the 16 entries kept on chip (22 and 27 entries at full size): the fixed
    overflow area carries them, so the call succeeds, hands over no more pixels than rtamd_render does,
    and gives the same image."""
    from raytracing_amd import backend, scenes
    w, h = 480, 270
    if scene == "mesh":
        buf, frame, sky = scenes.mesh_scene(w, h, target_tris=165_256)
    else:
        buf, frame, sky = scenes.instanced_scene(w, h, instances=400, tris_per_blas=20_000)
    p = backend.RayTracingPass(0)
    try:
        p.SetScene(buf)
        p.SetSkyBox(sky)
        for kw in (dict(mode=T.MODE_REF), dict(mode=T.MODE_PT, spp=2, max_bounces=6, seed=5)):
            a = _render(p, "render", frame, w, h, **kw)
            b = _render(p, "noalloc", frame, w, h, **kw)
            assert b[3]["free_delta"] == 0
            assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2], equal_nan=True)
            assert a[3]["rays"] == b[3]["rays"] and a[3]["rays"] > 0
            # (PT through rtamd_render hands over single paths, here whole pixels: at most one pixel per path)
            assert b[3]["handover_pixels"] <= max(a[3]["handover_pixels"], 8) + 600, (a[3]["handover_pixels"], b[3]["handover_pixels"])
    finally:
        p.Dispose()


def test_noalloc_entry_rejects_what_it_cannot_serve():
    import torch
    from raytracing_amd import backend, scenes
    p = backend.RayTracingPass(0)
    try:
        w, h = 64, 36
        buf, frame, sky = scenes.few_spheres(w, h)
        p.SetScene(buf)
        host = np.zeros((h, w, 4), dtype=np.float32)
        with pytest.raises(backend.BackendError, match="device memory"):
            p.ExecuteNoAlloc(frame, w, h, out_rgba=host, mode=T.MODE_REF)
        big = torch.zeros((8,), dtype=torch.float32, device="cuda:0")      # never written: the size check comes first
        with pytest.raises(backend.BackendError, match="larger than 3840x2160"):
            p.ExecuteNoAlloc(frame, 4096, 2304, out_rgba=big, mode=T.MODE_REF)
        with pytest.raises(backend.BackendError, match="running sum"):
            p.ExecuteNoAlloc(frame, w, h, out_rgba=torch.zeros((h, w, 4), device="cuda:0"), mode=T.MODE_PT, spp=1, max_bounces=1,
                             flags=T.FLAG_ACCUMULATE)
    finally:
        p.Dispose()

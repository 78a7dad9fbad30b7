"""GPU tests of the C-ABI behaviour around the kernels: golden vectors, tiles / shards,
accumulation, object refresh, device-pointer outputs, the presentation step, error codes."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from raytracing_amd import backend, host as H, scenes, types as T

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SUMMARY = json.load(open(os.path.join(GOLD, "summary.json")))
CASES = {"few_spheres_64x36": "scene_few", "random_spheres_g6_96x54": "scene_random_small", "mesh_20k_96x54": "scene_mesh_small"}


@pytest.mark.parametrize("case", sorted(CASES))
def test_golden_vectors(request, gpu_pass, case):
    buf, frame, sky = request.getfixturevalue(CASES[case])
    s = SUMMARY[case]
    w, h = s["width"], s["height"]
    g = np.load(os.path.join(GOLD, case + ".npz"))
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    r = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, mode=T.MODE_REF)
    assert np.array_equal(r["ids"], g["ids"]) and np.array_equal(r["t"], g["t"]) and np.array_equal(r["rgba"], g["ref_rgba"])
    p = gpu_pass.Execute(frame, w, h, out_rgba=True, mode=T.MODE_PT, **s["pt"])
    assert np.array_equal(p["rgba"], g["pt_rgba"])
    assert p["rays"] == s["pt_counters"]["rays"]
    c = gpu_pass.Execute(frame, w, h, out_rgba=True, counters=True, mode=T.MODE_PT, **s["pt"])["counters"]
    for k in ("rays", "nodes", "tris", "instances", "hits", "max_blas_stack", "max_tlas_stack"):
        assert c[k] == s["pt_counters"][k]


def test_default_dispatch_is_ref_mode(gpu_pass, oracle, scene_few):
    buf, frame, sky = scene_few
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    rgba = np.zeros((36, 64, 4), dtype=np.float32)
    ids = np.zeros((36, 64, 2), dtype=np.int32)
    gpu_pass.ExecuteDefault(frame, 64, 36, rgba, ids)         # ext == NULL
    o = oracle.render(buf, frame, 64, 36, sky=sky, mode=0)
    assert np.array_equal(rgba, o["rgba"]) and np.array_equal(ids, o["ids"])
    # primary miss: (0,0,0,0) so that the additive blit shows the raster scene (compute:427)
    miss = o["ids"][..., 0] < 0
    assert miss.any() and (rgba[miss] == 0).all() and (rgba[~miss][:, 3] == 1).all()


def test_black_sky_when_none_uploaded(gpu_pass, oracle, scene_few):
    buf, frame, _ = scene_few
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(None)
    g = gpu_pass.Execute(frame, 64, 36, out_rgba=True, mode=T.MODE_REF)
    o = oracle.render(buf, frame, 64, 36, sky=None, mode=0)
    assert np.array_equal(g["rgba"], o["rgba"])


@pytest.mark.parametrize("ref_order", [True, False])
def test_tiles_and_shards(gpu_pass, oracle, scene_random_small, ref_order):
    buf, frame, sky = scene_random_small
    w, h = 96, 54
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    fl = T.FLAG_REFERENCE_ORDER if ref_order else 0
    kw = dict(mode=T.MODE_PT, spp=2, max_bounces=3, seed=9, flags=fl)
    full = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=2, max_bounces=3, seed=9)
    acc = np.zeros((h, w, 4), dtype=np.float32)
    ids = np.full((h, w, 2), -7, dtype=np.int32)
    rays = 0
    for r in range(3):
        rays += gpu_pass.Execute(frame, w, h, out_rgba=acc, out_prim_id=ids, shard_rank=r, shard_count=3, **kw)["rays"]
    assert np.array_equal(acc, full["rgba"]) and np.array_equal(ids, full["ids"])
    assert rays == full["counters"]["rays"]
    tile = np.full((h, w, 4), 5.0, dtype=np.float32)
    gpu_pass.Execute(frame, w, h, out_rgba=tile, tile_x=10, tile_y=7, tile_w=33, tile_h=20, **kw)
    assert np.array_equal(tile[7:27, 10:43], full["rgba"][7:27, 10:43])
    outside = np.ones((h, w), dtype=bool)
    outside[7:27, 10:43] = False
    assert (tile[outside] == 5.0).all()                      # pixels outside the rectangle are untouched
    # a rectangle hanging over the image edge is clipped
    tile2 = np.zeros((h, w, 4), dtype=np.float32)
    gpu_pass.Execute(frame, w, h, out_rgba=tile2, tile_x=80, tile_y=40, tile_w=100, tile_h=100, **kw)
    assert np.array_equal(tile2[40:, 80:], full["rgba"][40:, 80:])


def test_progressive_accumulation(gpu_pass, oracle, scene_few):
    buf, frame, sky = scene_few
    w, h = 64, 36
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    gpu_pass.ResetAccumulation()
    kw = dict(mode=T.MODE_PT, max_bounces=3, seed=21)
    sums = []
    for i in range(3):
        out = gpu_pass.Execute(frame, w, h, out_rgba=True, spp=2, sample_offset=2 * i, flags=T.FLAG_ACCUMULATE, **kw)["rgba"]
        sums.append(oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=2, sample_offset=2 * i, max_bounces=3, seed=21,
                                  flags=8, want_ids=False, want_t=False)["rgba"])
        acc = np.zeros_like(sums[0])
        for s in sums:
            acc = acc + s                                       # same fp32 additions, same order
        expect = acc.copy()
        expect[..., :3] = acc[..., :3] / acc[..., 3:4]
        expect[..., 3] = 1
        assert np.array_equal(out, expect), i
    gpu_pass.ResetAccumulation()
    again = gpu_pass.Execute(frame, w, h, out_rgba=True, spp=2, sample_offset=0, flags=T.FLAG_ACCUMULATE, **kw)["rgba"]
    first = sums[0].copy()
    first[..., :3] /= first[..., 3:4]
    first[..., 3] = 1
    assert np.array_equal(again, first)
    gpu_pass.ResetAccumulation()


def test_running_sum_saved_and_restored_continues_with_the_same_bits(gpu_pass, scene_few):
    """A progressive render stopped after two frames, its running sum taken out (rtamd_accum_download), put into a
    fresh context (rtamd_accum_upload) and continued there ends with the bits of the render that never stopped."""
    from raytracing_amd import backend
    buf, frame, sky = scene_few
    w, h = 64, 36
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    kw = dict(mode=T.MODE_PT, max_bounces=3, seed=5, spp=2, flags=T.FLAG_ACCUMULATE)
    gpu_pass.ResetAccumulation()
    for i in range(4):
        straight = gpu_pass.Execute(frame, w, h, out_rgba=True, sample_offset=2 * i, **kw)["rgba"]
    gpu_pass.ResetAccumulation()
    for i in range(2):
        gpu_pass.Execute(frame, w, h, out_rgba=True, sample_offset=2 * i, **kw)
    saved = gpu_pass.DownloadAccumulation(w, h)
    assert np.all(saved[..., 3] == 4.0)
    gpu_pass.ResetAccumulation()
    other = backend.RayTracingPass(0)
    try:
        other.SetScene(buf)
        other.SetSkyBox(sky)
        other.UploadAccumulation(saved)
        for i in range(2, 4):
            resumed = other.Execute(frame, w, h, out_rgba=True, sample_offset=2 * i, **kw)["rgba"]
        assert np.array_equal(resumed, straight)
        with pytest.raises(backend.BackendError, match="no running sum of this size"):
            other.DownloadAccumulation(w + 1, h)
    finally:
        other.Dispose()


def test_refresh_objects_moves_an_instance(gpu_pass, oracle):
    sys = H.BuildAccelerateStructureSystem()
    sph = sys.BakeBLAS(H.Mesh.uv_sphere(16, 8, 0.5))
    quad = sys.BakeBLAS(H.Mesh.quad())
    sys.AddEntity(quad, H.trs((0, 0, 0), H.quat_axis_angle((1, 0, 0), 90), (20, 20, 1)), albedo=(1, 1, 1), specular=(0.1, 0.1, 0.1))
    e = sys.AddEntity(sph, H.trs((-1, 0.5, 0)), albedo=(0, 0, 0), specular=(0.8, 0.8, 0.8))
    sys.AddEntity(sph, H.trs((1, 0.5, 0)), emission=(1, 0.5, 0.2))
    frame = scenes.look_at_camera((0.2, 1.5, -4.0), (0, 0.4, 0), 45.0, 64, 36)
    sky = scenes.procedural_sky(64, 32)
    b0 = sys.OnUpdate()
    gpu_pass.SetScene(b0)
    gpu_pass.SetSkyBox(sky)
    for step in range(3):
        sys.SetLocalToWorld(e, H.trs((-1 + 0.4 * step, 0.5 + 0.2 * step, 0.3 * step), H.quat_axis_angle((0, 1, 0), 30 * step)))
        b = sys.OnUpdate()
        gpu_pass.RefreshObjects(b)
        g = gpu_pass.Execute(frame, 64, 36, out_rgba=True, out_prim_id=True, mode=T.MODE_REF)
        o = oracle.render(b, frame, 64, 36, sky=sky, mode=0)
        assert np.array_equal(g["ids"], o["ids"]) and np.array_equal(g["rgba"], o["rgba"])
    sys.close()
    with pytest.raises(backend.BackendError) as ei:
        gpu_pass.RefreshObjects(b0.__class__(b0.tlas, b0.blas_offsets, b0.blas, b0.vertices_offsets, b0.vertices,
                                             b0.triangles_offsets, b0.triangles, b0.objects[:2]))
    assert ei.value.code == -1


def test_device_pointer_outputs(gpu_pass, oracle, scene_few):
    torch = pytest.importorskip("torch")
    buf, frame, sky = scene_few
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    rgba = torch.zeros((36, 64, 4), dtype=torch.float32, device="cuda:0")
    ids = torch.zeros((36, 64, 2), dtype=torch.int32, device="cuda:0")
    t = torch.zeros((36, 64), dtype=torch.float32, device="cuda:0")
    gpu_pass.Execute(frame, 64, 36, out_rgba=rgba, out_prim_id=ids, out_t=t, mode=T.MODE_REF)
    o = oracle.render(buf, frame, 64, 36, sky=sky, mode=0)
    assert np.array_equal(rgba.cpu().numpy(), o["rgba"]) and np.array_equal(ids.cpu().numpy(), o["ids"])
    assert np.array_equal(t.cpu().numpy(), o["t"])


def test_composite_add_is_the_reference_blit(gpu_pass):
    rs = np.random.RandomState(0)
    src = rs.uniform(0, 1.5, size=(9, 16, 4)).astype(np.float32)
    src[..., 3] = rs.randint(0, 2, size=(9, 16))
    dst = rs.randint(0, 256, size=(9, 16, 4)).astype(np.uint8)
    expect = np.clip(dst.astype(np.float32) / np.float32(255) + src * src[..., 3:4], 0, 1)
    expect = np.floor(expect * np.float32(255) + np.float32(0.5)).astype(np.uint8)
    out = dst.copy()
    gpu_pass.CompositeAdd(src, out)
    assert np.array_equal(out, expect)
    assert np.array_equal(out[src[..., 3] == 0], dst[src[..., 3] == 0])   # alpha 0: the raster image shows through


def test_present_tone_maps(gpu_pass):
    rs = np.random.RandomState(1)
    src = rs.uniform(0, 6.0, size=(7, 20, 4)).astype(np.float32)
    src[..., 3] = rs.uniform(0, 1.2, size=(7, 20))
    one, h, q = np.float32(1), np.float32(0.5), np.float32(255)

    def quant(x):
        return np.floor(np.clip(x, 0, 1) * q + h).astype(np.uint8)

    out = np.zeros((7, 20, 4), dtype=np.uint8)
    gpu_pass.Present(src, out)                                   # the reference's behaviour: clamp at 1
    assert np.array_equal(out, quant(src))
    gpu_pass.Present(src, out, tonemap=1, exposure=0.5)          # Reinhard, correctly rounded divide: exact
    x = src[..., :3] * h
    assert np.array_equal(out[..., :3], quant(x / (one + x)))
    assert np.array_equal(out[..., 3], quant(src[..., 3]))
    gpu_pass.Present(src, out, tonemap=1, gamma=2.2)             # powf: within one 8-bit step
    x = src[..., :3]
    want = quant(np.power(np.clip(x / (one + x), 0, 1), one / np.float32(2.2)))
    assert np.abs(out[..., :3].astype(np.int32) - want.astype(np.int32)).max() <= 1
    with pytest.raises(backend.BackendError):
        gpu_pass.Present(src, out, tonemap=7)


def test_error_codes(scene_few):
    buf, frame, _ = scene_few
    p = backend.RayTracingPass(0)
    with pytest.raises(backend.BackendError) as ei:
        p.Execute(frame, 64, 36, out_rgba=True)
    assert ei.value.code == -6                                   # no scene: the reference skips the dispatch
    with pytest.raises(backend.BackendError) as ei:
        p.TlasInfo()
    assert ei.value.code == -6
    assert backend.lib().rtamd_set_tlas_build(p._ctx, 3) == -1 and backend.lib().rtamd_set_tlas_build(p._ctx, -1) == -1
    assert backend.lib().rtamd_set_tlas_build(None, 0) == -1 and backend.lib().rtamd_last_trace_kernel_ms(p._ctx, None) == -1
    bad = H.SceneBuffers(buf.tlas, buf.blas_offsets, buf.blas, buf.vertices_offsets, buf.vertices, buf.triangles_offsets,
                         buf.triangles, buf.objects)
    bad.triangles = buf.triangles.copy()
    bad.triangles["x"][5] = 10 ** 6
    with pytest.raises(backend.BackendError) as ei:
        p.SetScene(bad)
    assert ei.value.code == -2
    bad2 = H.SceneBuffers(buf.tlas.copy(), buf.blas_offsets, buf.blas, buf.vertices_offsets, buf.vertices,
                          buf.triangles_offsets, buf.triangles, buf.objects)
    bad2.tlas["Data"][1][0] = 10 ** 6
    with pytest.raises(backend.BackendError) as ei:
        p.SetScene(bad2)
    assert ei.value.code == -2
    p.SetScene(buf)
    for kw in (dict(mode=7), dict(mode=T.MODE_PT, spp=0), dict(shard_rank=3, shard_count=2)):
        with pytest.raises(backend.BackendError) as ei:
            p.Execute(frame, 64, 36, out_rgba=True, **kw)
        assert ei.value.code == -1
    with pytest.raises(backend.BackendError) as ei:
        p.Execute(frame, 0, 36, out_rgba=True)
    assert ei.value.code == -1
    e = p.ext()
    e.struct_size = 8
    assert backend.lib().rtamd_render(p._ctx, frame.ctypes.data_as(C.c_void_p), C.byref(e), 64, 36, None, None) == -1
    assert b"struct_size" in backend.lib().rtamd_last_error_string(p._ctx)
    p.Dispose()


def test_deep_chain_shaped_tree(gpu_pass, oracle):
    """A degenerate chain-shaped BLAS makes the reference's own stack grow past 100 entries;
    the library (which rebuilds the tree above the triangles) must return the reference's image."""
    n = 200
    mesh = H.Mesh.quad()
    # quads stacked along +z, facing +z (winding and normals flipped), seen from the far end
    pos = np.concatenate([mesh.positions * 0.9 + np.array([0, 0, 0.05 * i], dtype=np.float32) for i in range(n)])
    nrm = np.concatenate([-mesh.normals] * n)
    idx = np.concatenate([mesh.indices[:, ::-1] + 4 * i for i in range(n)])
    big = H.Mesh(pos, nrm, idx)
    aabbs, _ = big.triangle_aabbs()
    # hand-built chain: internal node k = [leaf node of 3 triangles, internal node k+1].  The
    # reference pushes both and pops the chain link first, so its stack grows by one per
    # level; a nearest-first walk from the far end does the same.
    nt = len(idx)
    leaves = (nt + 2) // 3
    nodes = np.zeros(2 + 2 * leaves, dtype=T.NODE)
    for f in ("Lx", "Ly", "Lz"):
        nodes[f][:] = np.float32(3.4028234663852886e38)
    for f in ("Hx", "Hy", "Hz"):
        nodes[f][:] = np.float32(-3.4028234663852886e38)

    def setbox(node, slot, lo, hi):
        for a, f in enumerate("xyz"):
            nodes["L" + f][node][slot] = lo[a]
            nodes["H" + f][node][slot] = hi[a]
    # layout: internal nodes 1..leaves, leaf nodes leaves+1..2*leaves
    for k in range(leaves, 0, -1):
        leaf = leaves + k
        tris = list(range(3 * (k - 1), min(3 * k, nt)))
        nodes["Flags"][leaf] = 1
        nodes["Data"][leaf] = [-1, -1, -1, -1]
        for s_, t in enumerate(tris):
            nodes["Data"][leaf][s_] = t
            setbox(leaf, s_, aabbs["Min"][t], aabbs["Max"][t])
        nodes["Data"][k] = [leaf, k + 1 if k < leaves else 0, 0, 0]
        setbox(k, 0, aabbs["Min"][tris].min(axis=0), aabbs["Max"][tris].max(axis=0))
        if k < leaves:
            setbox(k, 1, [min(nodes["L" + f][k + 1][:2]) for f in "xyz"], [max(nodes["H" + f][k + 1][:2]) for f in "xyz"])
    assert oracle.check_integrity(nodes, nt) == 0
    sys = H.BuildAccelerateStructureSystem()
    b = sys.BakeBLAS(big)
    sys.AddEntity(b, H.trs((0, 0, 0)), emission=(1, 1, 1))
    buf = sys.OnUpdate()
    sys.close()
    buf.blas = nodes
    frame = scenes.look_at_camera((0.05, 0.02, 14.0), (0, 0, 0), 10.0, 32, 32)
    o = oracle.render(buf, frame, 32, 32, mode=0)
    assert o["counters"]["max_blas_stack"] > 100 and not o["overflow"]
    assert (o["ids"][..., 0] == 0).mean() > 0.5
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(None)
    g = gpu_pass.Execute(frame, 32, 32, out_rgba=True, out_prim_id=True, mode=T.MODE_REF)
    assert np.array_equal(g["ids"], o["ids"]) and np.array_equal(g["rgba"], o["rgba"])


def test_traversal_stack_overflow_area(gpu_pass, oracle, scene_mesh_small, monkeypatch):
    """With the LDS part of the traversal stack cut to 12 entries, rays of the mesh scene move
    chunks of their stack to the overflow area in HBM and back; the image must not change."""
    buf, frame, sky = scene_mesh_small
    w, h = 96, 54
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    kw = dict(mode=T.MODE_PT, spp=4, max_bounces=4, seed=1234)
    o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=4, max_bounces=4, seed=1234)
    monkeypatch.setenv("RTAMD_STACK_LIMIT", "12")
    g = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, **kw)
    monkeypatch.delenv("RTAMD_STACK_LIMIT")
    assert np.array_equal(g["ids"], o["ids"]) and np.array_equal(g["t"], o["t"], equal_nan=True)
    assert np.array_equal(g["rgba"], o["rgba"])
    assert g["rays"] == o["counters"]["rays"]


def test_sample_shards_sum_to_the_whole(gpu_pass, oracle, scene_random_small):
    """Sample sharding as the multi-GPU bench does it: rank r renders samples [r*spp, (r+1)*spp) as raw
    sums (RTAMD_FLAG_OUTPUT_SUM); each shard equals the oracle's bit for bit, and the shards added
    and divided agree with the single render of all samples to fp32 summation-order tolerance."""
    buf, frame, sky = scene_random_small
    w, h = 96, 54
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    kw = dict(mode=T.MODE_PT, max_bounces=4, seed=9)
    total = np.zeros((h, w, 4), dtype=np.float32)
    for r in range(4):
        part = gpu_pass.Execute(frame, w, h, out_rgba=True, spp=3, sample_offset=3 * r, flags=T.FLAG_OUTPUT_SUM, **kw)["rgba"]
        o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=3, sample_offset=3 * r, max_bounces=4, seed=9, flags=T.FLAG_OUTPUT_SUM,
                          want_ids=False, want_t=False)["rgba"]
        assert np.array_equal(part, o), r
        total += part
    assert (total[..., 3] == 12).all()
    mean = total[..., :3] / total[..., 3:4]
    whole = gpu_pass.Execute(frame, w, h, out_rgba=True, spp=12, **kw)["rgba"]
    np.testing.assert_allclose(mean, whole[..., :3], rtol=2e-6, atol=1e-6)      # same 12 samples, added in a different order


def test_rejected_refresh_keeps_the_last_good_scene(oracle, scene_few):
    """A refresh whose TLAS or object array fails validation changes nothing on the device: the next
    frame renders the last good buffers, as the reference would keep drawing them."""
    buf, frame, sky = scene_few
    p = backend.RayTracingPass(0)
    try:
        p.SetScene(buf)
        p.SetSkyBox(sky)
        good = p.Execute(frame, 64, 36, out_rgba=True, out_prim_id=True, mode=T.MODE_REF)
        bad = H.SceneBuffers(buf.tlas.copy(), buf.blas_offsets, buf.blas, buf.vertices_offsets, buf.vertices,
                             buf.triangles_offsets, buf.triangles, buf.objects.copy())
        bad.tlas["Data"][1][0] = 10 ** 6
        with pytest.raises(backend.BackendError) as ei:
            p.RefreshObjects(bad)
        assert ei.value.code == -2
        bad.tlas = buf.tlas
        bad.objects["BLASIndex"][1] = 99
        with pytest.raises(backend.BackendError) as ei:
            p.RefreshObjects(bad)
        assert ei.value.code == -2
        again = p.Execute(frame, 64, 36, out_rgba=True, out_prim_id=True, mode=T.MODE_REF)
        assert np.array_equal(good["rgba"], again["rgba"]) and np.array_equal(good["ids"], again["ids"])
        # the counters flag without a place to put them is an argument error
        e = p.ext(mode=T.MODE_REF)
        e.flags = T.FLAG_COUNTERS
        assert backend.lib().rtamd_render(p._ctx, frame.ctypes.data_as(C.c_void_p), C.byref(e), 64, 36, None, None) == -1
    finally:
        p.Dispose()


def test_many_batches_are_all_timed(gpu_pass, monkeypatch):
    """A frame rendered in more than 32 launches of the path kernel (small sample buffer): the reported
    time of that kernel covers every launch, so it stays below the whole call's device time and close to it."""
    buf, frame, sky = scenes.random_spheres(320, 180)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    kw = dict(mode=T.MODE_PT, spp=40, max_bounces=6, seed=3)
    whole = gpu_pass.Execute(frame, 320, 180, out_rgba=True, **kw)
    monkeypatch.setenv("RTAMD_POOL_MAX_PATHS", "61440")       # 60 tiles x 1024 work items: one sample per batch, 40 batches
    many = gpu_pass.Execute(frame, 320, 180, out_rgba=True, **kw)
    monkeypatch.delenv("RTAMD_POOL_MAX_PATHS")
    assert np.array_equal(whole["rgba"], many["rgba"]) and whole["rays"] == many["rays"]
    assert many["trace_ms"] <= many["kernel_ms"]
    assert many["trace_ms"] > 0.6 * (many["kernel_ms"] - many["handover_ms"])

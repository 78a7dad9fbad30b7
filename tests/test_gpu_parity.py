"""GPU parity: the HIP kernels, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Bar: primary-hit (object, triangle) ids bit-exact; t and radiance
equal as IEEE values (tolerance 0; -0 == +0) because both sides evaluate the same
declared binary32 operations."""
import numpy as np
import pytest

from raytracing_amd import types as T

pytestmark = pytest.mark.gpu


def _same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def _report(name, g, o):
    ids_bad = int((g["ids"] != o["ids"]).any(-1).sum()) if g.get("ids") is not None else 0
    rgba_bad = int((g["rgba"] != o["rgba"]).any(-1).sum()) if g.get("rgba") is not None else 0
    maxd = float(np.abs(g["rgba"] - o["rgba"]).max()) if g.get("rgba") is not None else 0.0
    return f"{name}: id mismatches {ids_bad}, rgba mismatching pixels {rgba_bad}, max |d| {maxd}"


@pytest.mark.parametrize("scene_name", ["scene_few", "scene_random_small", "scene_mesh_small"])
@pytest.mark.parametrize("ref_order", [True, False])
def test_ref_mode_matches_oracle(request, gpu_pass, oracle, scene_name, ref_order):
    buf, frame, sky = request.getfixturevalue(scene_name)
    h, w = (36, 64) if scene_name == "scene_few" else (54, 96)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    flags = T.FLAG_REFERENCE_ORDER if ref_order else 0
    g = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, mode=T.MODE_REF, flags=flags)
    o = oracle.render(buf, frame, w, h, sky=sky, mode=0)
    assert _same(g["ids"], o["ids"]), _report(scene_name, g, o)
    assert _same(g["t"], o["t"])
    assert _same(g["rgba"], o["rgba"]), _report(scene_name, g, o)
    assert g["rays"] == o["counters"]["rays"]
    assert (o["ids"][..., 0] >= 0).mean() > 0.2      # the scene is actually in view


@pytest.mark.parametrize("scene_name", ["scene_few", "scene_random_small", "scene_mesh_small"])
def test_reference_order_counters(request, gpu_pass, oracle, scene_name):
    buf, frame, sky = request.getfixturevalue(scene_name)
    h, w = (36, 64) if scene_name == "scene_few" else (54, 96)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    for mode, kw in ((T.MODE_REF, {}), (T.MODE_PT, dict(spp=2, max_bounces=3, seed=11))):
        g = gpu_pass.Execute(frame, w, h, out_rgba=True, counters=True, mode=mode, **kw)
        o = oracle.render(buf, frame, w, h, sky=sky, mode=mode, spp=kw.get("spp", 1), max_bounces=kw.get("max_bounces", 0),
                          seed=kw.get("seed", 0), want_ids=False, want_t=False)
        for k in ("rays", "nodes", "tris", "instances", "hits", "max_blas_stack", "max_tlas_stack"):
            assert g["counters"][k] == o["counters"][k], (mode, k, g["counters"], o["counters"])


@pytest.mark.parametrize("scene_name", ["scene_few", "scene_random_small", "scene_mesh_small"])
@pytest.mark.parametrize("ref_order", [True, False])
def test_path_tracing_matches_oracle(request, gpu_pass, oracle, scene_name, ref_order):
    buf, frame, sky = request.getfixturevalue(scene_name)
    h, w = (36, 64) if scene_name == "scene_few" else (54, 96)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    flags = T.FLAG_REFERENCE_ORDER if ref_order else 0
    g = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, mode=T.MODE_PT, spp=4, max_bounces=4,
                         seed=1234, flags=flags)
    o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=4, max_bounces=4, seed=1234)
    assert _same(g["ids"], o["ids"]), _report(scene_name, g, o)
    assert _same(g["t"], o["t"])
    assert g["rays"] == o["counters"]["rays"]
    assert _same(g["rgba"], o["rgba"]), _report(scene_name, g, o)
    assert np.isfinite(g["rgba"]).all()


def test_many_instances_tlas_larger_than_the_lds_copy(gpu_pass, oracle):
    """3 000 instances (TLAS from the SAH builder): the TLAS has more inner nodes than the kernel
    keeps in LDS, and the three BLAS trees share what is left of it."""
    from raytracing_amd import scenes
    w, h = 96, 54
    buf, frame, sky = scenes.instanced_scene(w, h, instances=3000, tris_per_blas=600, tlas_tree=1)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    g = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, mode=T.MODE_PT, spp=3, max_bounces=3, seed=99)
    o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=3, max_bounces=3, seed=99)
    assert _same(g["ids"], o["ids"]), _report("instanced", g, o)
    assert _same(g["t"], o["t"])
    assert _same(g["rgba"], o["rgba"]), _report("instanced", g, o)
    assert g["rays"] == o["counters"]["rays"]
    assert (o["ids"][..., 0] > 0).mean() > 0.2


@pytest.mark.parametrize("workload", ["spheres", "mesh"])
def test_full_size_production_kernel_equals_reference_order_kernel(gpu_pass, oracle, workload):
    """Benchmark scenes at their full size — 485 instances; peter / bunny / cow subdivided to
    1 051 936 triangles — at 1920x1080 (2 spp, so that the CPU oracle finishes in seconds): the
    production kernel (rebuilt trees, nearest-first, LDS copies), the reference-order kernel (the
    caller's trees, the shader's visiting order) and the oracle must agree on every pixel, bit
    for bit; and a render split in shards must equal the whole."""
    from raytracing_amd import scenes
    w, h = 1920, 1080
    buf, frame, sky = scenes.random_spheres(w, h) if workload == "spheres" else scenes.mesh_scene(w, h)
    if workload == "mesh":
        assert len(buf.triangles) == 1_051_936 + 2 and buf.n_blas == 4
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    kw = dict(mode=T.MODE_PT, spp=2, max_bounces=8, seed=77)
    fast = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, **kw)
    ref = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, flags=T.FLAG_REFERENCE_ORDER, **kw)
    assert _same(fast["ids"], ref["ids"]), _report(workload, fast, ref)
    assert _same(fast["t"], ref["t"])
    assert _same(fast["rgba"], ref["rgba"]), _report(workload, fast, ref)
    assert fast["rays"] == ref["rays"]
    o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=2, max_bounces=8, seed=77, nthreads=16)
    assert _same(fast["ids"], o["ids"]), _report(workload, fast, o)
    assert _same(fast["t"], o["t"])
    assert _same(fast["rgba"], o["rgba"]), _report(workload, fast, o)
    assert fast["rays"] == o["counters"]["rays"]
    # two interleaved shards compose to the same image
    parts = np.zeros_like(fast["rgba"])
    for r in range(2):
        part = np.zeros_like(parts)
        gpu_pass.Execute(frame, w, h, out_rgba=part, shard_rank=r, shard_count=2, **kw)
        parts += part
    assert _same(parts, fast["rgba"])


def _tie_scene(w, h):
    """Geometry made to produce exact ties and near-ties: coincident instances (same mesh, same
    transform), a mesh with every triangle stored twice plus zero-area triangles, coplanar
    overlapping quads, and a second sphere mesh shifted by a hair."""
    from raytracing_amd import host as H, scenes
    sys_ = H.BuildAccelerateStructureSystem()
    sph = H.Mesh.uv_sphere(24, 12, 0.5)
    doubled = H.Mesh(sph.positions, sph.normals,
                     np.concatenate([sph.indices, sph.indices, np.repeat(sph.indices[:40, :1], 3, axis=1)]))   # twice + degenerate
    b_sph = sys_.BakeBLAS(sph)
    b_dbl = sys_.BakeBLAS(doubled)
    b_quad = sys_.BakeBLAS(H.Mesh.quad())
    rot = H.quat_axis_angle((1, 0, 0), 90.0)
    sys_.AddEntity(b_quad, H.trs((0, 0, 0), rot, (30, 30, 1)), albedo=(0.7, 0.7, 0.7), specular=(0.05, 0.05, 0.05))
    sys_.AddEntity(b_quad, H.trs((0.3, 0, 0.2), rot, (10, 10, 1)), albedo=(0.2, 0.7, 0.2), specular=(0.05, 0.05, 0.05))   # coplanar, overlapping
    for k, x in enumerate((-1.5, 0.0, 1.5)):
        m = H.trs((x, 0.5, 0.0), s=(1, 1, 1))
        sys_.AddEntity(b_sph, m, albedo=(0.8, 0.3, 0.2), specular=(0.1, 0.1, 0.1))
        sys_.AddEntity(b_sph, m, albedo=(0.2, 0.3, 0.8), specular=(0.6, 0.6, 0.6))          # coincident instance
        sys_.AddEntity(b_dbl, H.trs((x, 0.5, 1.4)), albedo=(0.9, 0.9, 0.2), specular=(0.3, 0.3, 0.3))
        sys_.AddEntity(b_sph, H.trs((x + 1e-6, 0.5, 1.4)), albedo=(0.3, 0.9, 0.9), specular=(0.0, 0.0, 0.0))   # a hair away
    sys_.AddEntity(b_quad, H.trs((0, 4.0, 0.5), H.quat_axis_angle((1, 0, 0), -90.0), (3, 3, 1)), albedo=(0, 0, 0), specular=(0, 0, 0), emission=(2, 2, 2))
    buf = sys_.OnUpdate()
    sys_.close()
    frame = scenes.look_at_camera((0.2, 2.2, -6.0), (0.0, 0.4, 0.6), 40.0, w, h)
    return buf, frame, scenes.procedural_sky()


def test_exact_ties_and_near_ties(gpu_pass, oracle):
    """Coincident instances, duplicated and zero-area triangles, coplanar quads: the winner must be
    the one the reference's visiting order gives, in the production kernel as in the oracle."""
    w, h = 160, 90
    buf, frame, sky = _tie_scene(w, h)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    for mode, kw in ((T.MODE_REF, {}), (T.MODE_PT, dict(spp=4, max_bounces=5, seed=5))):
        g = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, mode=mode, **kw)
        o = oracle.render(buf, frame, w, h, sky=sky, mode=mode, spp=kw.get("spp", 1), max_bounces=kw.get("max_bounces", 0),
                          seed=kw.get("seed", 0), nthreads=8)
        assert _same(g["ids"], o["ids"]), _report("ties", g, o)
        assert _same(g["t"], o["t"])
        assert _same(g["rgba"], o["rgba"]), _report("ties", g, o)
        assert g["rays"] == o["counters"]["rays"]
    assert len(np.unique(o["ids"][..., 0])) > 8


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_random_triangle_soups(gpu_pass, oracle, seed):
    """Random soups (slivers, zero-area triangles, shared vertices) under rotated, non-uniformly
    scaled and mirrored instances at coordinates from 1e-2 to 1e3, TLAS and BLAS from either builder."""
    from raytracing_amd import scenes
    w, h = 128, 72
    buf, frame, sky = scenes.random_soup(w, h, seed=seed)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    g = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, mode=T.MODE_PT, spp=3, max_bounces=6, seed=seed)
    o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=3, max_bounces=6, seed=seed, nthreads=8)
    assert _same(g["ids"], o["ids"]), _report("soup", g, o)
    assert _same(g["t"], o["t"])
    assert _same(g["rgba"], o["rgba"]), _report("soup", g, o)
    assert g["rays"] == o["counters"]["rays"]


@pytest.mark.parametrize("tris_per_mesh", [1, 2, 3, 5, 7])
@pytest.mark.parametrize("tlas", ["auto", "device"])
def test_tiny_meshes(gpu_pass, oracle, tris_per_mesh, tlas):
    """Meshes of 1 to 7 triangles: a mesh that fits one leaf of the rebuilt tree is entered at that leaf
    (no root node), two-triangle leaves start on a line of their own with a blank record behind a single
    triangle, and roots have fewer than four children."""
    from raytracing_amd import scenes
    w, h = 128, 72
    buf, frame, sky = scenes.random_soup(w, h, seed=40 + tris_per_mesh, meshes=5, tris_per_mesh=tris_per_mesh, instances=60)
    gpu_pass.SetTlasBuild(tlas)
    try:
        gpu_pass.SetScene(buf)
        gpu_pass.SetSkyBox(sky)
        o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=3, max_bounces=6, seed=9, nthreads=8)
        g = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, mode=T.MODE_PT, spp=3, max_bounces=6, seed=9)
        r = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, mode=T.MODE_REF)
        ro = oracle.render(buf, frame, w, h, sky=sky, mode=0)
    finally:
        gpu_pass.SetTlasBuild("auto")
    assert (o["ids"][..., 0] >= 0).mean() > 0.02
    assert _same(g["ids"], o["ids"]), _report("tiny meshes", g, o)
    assert _same(g["t"], o["t"])
    assert _same(g["rgba"], o["rgba"]), _report("tiny meshes", g, o)
    assert g["rays"] == o["counters"]["rays"]
    assert _same(r["ids"], ro["ids"]) and _same(r["t"], ro["t"]) and _same(r["rgba"], ro["rgba"])


@pytest.fixture
def device_tlas(gpu_pass):
    gpu_pass.SetTlasBuild("device")
    yield gpu_pass
    gpu_pass.SetTlasBuild("auto")


@pytest.mark.parametrize("instances", [1, 2, 3, 5, 64, 3000, 20000])
def test_tlas_built_on_the_device(device_tlas, oracle, instances):
    """The library's private tree over the instances built by the GPU kernels (Morton order, binary
    radix tree, collapse to 4-wide nodes): same image as the oracle bit for bit, whatever the tree."""
    from raytracing_amd import scenes
    w, h = 96, 54
    buf, frame, sky = scenes.instanced_scene(w, h, instances=instances, tris_per_blas=300, tlas_tree=1 if instances > 1200 else instances % 2)
    device_tlas.SetScene(buf)
    device_tlas.SetSkyBox(sky)
    info = device_tlas.TlasInfo()
    n_obj = len(buf.objects)
    assert info["built_on_device"]
    assert 1 <= info["nodes"] <= max(n_obj - 1, 1) and info["levels"] >= 1
    assert info["nodes"] >= (n_obj - 1 + 2) // 3        # a 4-wide node holds at most three more children than it replaces
    g = device_tlas.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, mode=T.MODE_PT, spp=3, max_bounces=3, seed=21)
    o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=3, max_bounces=3, seed=21, nthreads=8)
    assert _same(g["ids"], o["ids"]), _report("device tlas", g, o)
    assert _same(g["t"], o["t"])
    assert _same(g["rgba"], o["rgba"]), _report("device tlas", g, o)
    assert g["rays"] == o["counters"]["rays"]
    if instances >= 64:
        assert (o["ids"][..., 0] >= 0).mean() > 0.2


def test_tlas_built_on_the_device_ties_and_refresh(device_tlas, oracle):
    """Coincident instances give equal Morton codes and exact ties in t; a refresh with moved
    objects rebuilds the tree on the GPU; going back to the host builder gives the same image."""
    from raytracing_amd import scenes
    w, h = 160, 90
    buf, frame, sky = _tie_scene(w, h)
    device_tlas.SetScene(buf)
    device_tlas.SetSkyBox(sky)
    assert device_tlas.TlasInfo()["built_on_device"]
    kw = dict(mode=T.MODE_PT, spp=4, max_bounces=5, seed=5)
    g = device_tlas.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, **kw)
    o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=4, max_bounces=5, seed=5, nthreads=8)
    assert _same(g["ids"], o["ids"]), _report("ties", g, o)
    assert _same(g["t"], o["t"])
    assert _same(g["rgba"], o["rgba"]), _report("ties", g, o)

    buf_a, frame_a, sky_a = scenes.instanced_scene(96, 54, instances=500, tris_per_blas=300, seed=3)
    buf_b, _, _ = scenes.instanced_scene(96, 54, instances=500, tris_per_blas=300, seed=4)
    device_tlas.SetScene(buf_a)
    device_tlas.SetSkyBox(sky_a)
    device_tlas.RefreshObjects(buf_b)
    assert device_tlas.TlasInfo()["built_on_device"]
    gd = device_tlas.Execute(frame_a, 96, 54, out_rgba=True, out_prim_id=True, mode=T.MODE_PT, spp=2, max_bounces=3, seed=8)
    device_tlas.SetTlasBuild("host")
    device_tlas.RefreshObjects(buf_b)
    assert not device_tlas.TlasInfo()["built_on_device"]
    gh = device_tlas.Execute(frame_a, 96, 54, out_rgba=True, out_prim_id=True, mode=T.MODE_PT, spp=2, max_bounces=3, seed=8)
    assert _same(gd["ids"], gh["ids"]) and _same(gd["rgba"], gh["rgba"])
    ob = oracle.render(buf_b, frame_a, 96, 54, sky=sky_a, mode=1, spp=2, max_bounces=3, seed=8, nthreads=8)
    assert _same(gd["ids"], ob["ids"]) and _same(gd["rgba"], ob["rgba"])


def test_tlas_built_on_the_device_single_object(device_tlas, oracle):
    from raytracing_amd import host as H, scenes
    sys_ = H.BuildAccelerateStructureSystem()
    sys_.AddEntity(sys_.BakeBLAS(H.Mesh.uv_sphere(16, 8, 1.0)), H.trs((0, 0, 0)), albedo=(0.8, 0.4, 0.3), specular=(0.2, 0.2, 0.2))
    buf = sys_.OnUpdate()
    sys_.close()
    w, h = 64, 36
    frame = scenes.look_at_camera((0.5, 1.0, -4.0), (0, 0, 0), 40.0, w, h)
    sky = scenes.procedural_sky()
    device_tlas.SetScene(buf)
    device_tlas.SetSkyBox(sky)
    assert device_tlas.TlasInfo() == {"nodes": 1, "levels": 1, "built_on_device": True}
    g = device_tlas.Execute(frame, w, h, out_rgba=True, out_prim_id=True, mode=T.MODE_PT, spp=2, max_bounces=3, seed=2)
    o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=2, max_bounces=3, seed=2)
    assert _same(g["ids"], o["ids"]) and _same(g["rgba"], o["rgba"])
    assert (o["ids"][..., 0] >= 0).any()


def test_4k_frame_production_kernel_equals_reference_order_kernel(gpu_pass, oracle):
    """3840x2160 on the 1 051 936-triangle scene (the multi-GPU configuration's frame): production
    and reference-order kernels agree on every pixel, a 960x540 rectangle of the frame equals the
    CPU oracle's, and four interleaved tile shards compose to the whole frame."""
    from raytracing_amd import scenes
    w, h = 3840, 2160
    buf, frame, sky = scenes.mesh_scene(w, h)
    assert len(buf.triangles) == 1_051_936 + 2
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    kw = dict(mode=T.MODE_PT, spp=1, max_bounces=8, seed=4)
    fast = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, **kw)
    ref = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, flags=T.FLAG_REFERENCE_ORDER, **kw)
    tile = (1500, 700, 960, 540)
    o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=1, max_bounces=8, seed=4, tile=tile, nthreads=16)
    ys, xs = slice(tile[1], tile[1] + tile[3]), slice(tile[0], tile[0] + tile[2])
    assert _same(fast["ids"][ys, xs], o["ids"][ys, xs]) and _same(fast["t"][ys, xs], o["t"][ys, xs])
    assert _same(fast["rgba"][ys, xs], o["rgba"][ys, xs])
    assert len(np.unique(o["ids"][ys, xs, 0])) >= 4
    assert _same(fast["ids"], ref["ids"]), _report("4k", fast, ref)
    assert _same(fast["t"], ref["t"])
    assert _same(fast["rgba"], ref["rgba"]), _report("4k", fast, ref)
    assert fast["rays"] == ref["rays"]
    assert (fast["ids"][..., 0] >= 0).mean() > 0.2
    parts = np.zeros_like(fast["rgba"])
    rays = 0
    for r in range(4):
        part = np.zeros_like(parts)
        rays += gpu_pass.Execute(frame, w, h, out_rgba=part, shard_rank=r, shard_count=4, **kw)["rays"]
        parts += part
    assert _same(parts, fast["rgba"])
    assert rays == fast["rays"]


def test_fifty_million_instanced_triangles(gpu_pass, oracle):
    """1 250 instances of 40 000-triangle meshes (the largest configuration's scene): production
    kernel, on a host-built and on a device-built top-level tree, against the reference-order kernel
    on every pixel and against the CPU oracle on the whole 640x360 frame (path tracing and REF mode)."""
    from raytracing_amd import scenes
    w, h = 640, 360
    buf, frame, sky = scenes.instanced_scene(w, h, instances=1250, tris_per_blas=40_000)
    tri_count = np.array([buf.blas_slice(b)[1] for b in range(buf.n_blas)])
    assert int(tri_count[buf.objects["BLASIndex"]].sum()) > 49_000_000
    gpu_pass.SetSkyBox(sky)
    kw = dict(mode=T.MODE_PT, spp=2, max_bounces=6, seed=12)
    results = []
    for build in ("host", "device"):
        gpu_pass.SetTlasBuild(build)
        gpu_pass.SetScene(buf)
        results.append(gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, **kw))
    gpu_pass.SetTlasBuild("auto")
    ref = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, flags=T.FLAG_REFERENCE_ORDER, **kw)
    for g in results:
        assert _same(g["ids"], ref["ids"]), _report("50M", g, ref)
        assert _same(g["t"], ref["t"])
        assert _same(g["rgba"], ref["rgba"]), _report("50M", g, ref)
        assert g["rays"] == ref["rays"]
    assert (ref["ids"][..., 0] > 0).mean() > 0.2
    o = oracle.render(buf, frame, w, h, sky=sky, nthreads=16, **kw)
    g = results[0]
    assert _same(g["ids"], o["ids"]) and _same(g["t"], o["t"]), _report("50M vs oracle", g, o)
    assert _same(g["rgba"], o["rgba"]), _report("50M vs oracle", g, o)
    assert g["rays"] == o["counters"]["rays"]
    gr = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, mode=T.MODE_REF)
    orf = oracle.render(buf, frame, w, h, sky=sky, mode=0, nthreads=16)
    assert _same(gr["ids"], orf["ids"]) and _same(gr["t"], orf["t"]) and _same(gr["rgba"], orf["rgba"])


def test_both_path_tracing_kernels_give_the_same_image(gpu_pass, oracle, monkeypatch):
    """PT mode runs through the path-pool kernel (more paths than lanes per wave, full-wave shading
    passes, samples summed in order afterwards); the one-path-per-lane kernel it replaced stays
    selectable.  Same image bit for bit, in several batches as in one, and under tile shards."""
    from raytracing_amd import scenes
    w, h = 320, 180
    buf, frame, sky = scenes.random_spheres(w, h)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    kw = dict(mode=T.MODE_PT, spp=6, max_bounces=8, seed=31)
    pool = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, **kw)
    monkeypatch.setenv("RTAMD_PIPELINE", "persistent")
    lane = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, **kw)
    monkeypatch.delenv("RTAMD_PIPELINE")
    assert _same(pool["ids"], lane["ids"]) and _same(pool["t"], lane["t"])
    assert _same(pool["rgba"], lane["rgba"]), _report("pool vs lane", pool, lane)
    assert pool["rays"] == lane["rays"]
    o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=6, max_bounces=8, seed=31, nthreads=8)
    assert _same(pool["rgba"], o["rgba"]), _report("pool vs oracle", pool, o)
    assert pool["rays"] == o["counters"]["rays"]
    # sample buffer limited to 20 000 paths: 3 chunks of tiles x 6 batches of one sample
    monkeypatch.setenv("RTAMD_POOL_MAX_PATHS", "20000")
    small = gpu_pass.Execute(frame, w, h, out_rgba=True, **kw)
    # (batch i is added up on a second stream while batch i + 1 is traced; RTAMD_POOL_NO_OVERLAP keeps one stream)
    monkeypatch.setenv("RTAMD_POOL_NO_OVERLAP", "1")
    serial = gpu_pass.Execute(frame, w, h, out_rgba=True, **kw)
    monkeypatch.delenv("RTAMD_POOL_NO_OVERLAP")
    monkeypatch.delenv("RTAMD_POOL_MAX_PATHS")
    assert _same(small["rgba"], pool["rgba"]) and small["rays"] == pool["rays"]
    assert _same(serial["rgba"], pool["rgba"]) and serial["rays"] == pool["rays"]
    parts = np.zeros_like(pool["rgba"])
    for r in range(3):
        part = np.zeros_like(parts)
        gpu_pass.Execute(frame, w, h, out_rgba=part, shard_rank=r, shard_count=3, **kw)
        parts += part
    assert _same(parts, pool["rgba"])
    # the large-scene variant of the pool kernel (throughput words in LDS, few nodes in LDS) is chosen by node
    # count; forced onto this small scene it must give the same image, also with the stack overflowing
    monkeypatch.setenv("RTAMD_POOL_BIG", "1")
    big = gpu_pass.Execute(frame, w, h, out_rgba=True, **kw)
    monkeypatch.setenv("RTAMD_STACK_LIMIT", "12")
    big12 = gpu_pass.Execute(frame, w, h, out_rgba=True, **kw)
    monkeypatch.delenv("RTAMD_STACK_LIMIT")
    monkeypatch.delenv("RTAMD_POOL_BIG")
    assert _same(big["rgba"], pool["rgba"]) and big["rays"] == pool["rays"]
    assert _same(big12["rgba"], pool["rgba"]) and big12["rays"] == pool["rays"]


def test_paths_and_pixels_handed_to_the_reference_order_kernel(gpu_pass, oracle, monkeypatch):
    """Without jitter the centre row and column of the spheres scene's camera have rays with a zero direction
    component: every sample of those pixels is a path the pool kernel passes on.  With room in the list the
    paths go one by one; with a list of 5 (or none) the rest fall back to whole pixels.  Same image, same ray
    count as the oracle either way."""
    from raytracing_amd import scenes
    w, h = 192, 108
    buf, frame, sky = scenes.random_spheres(w, h)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    kw = dict(mode=T.MODE_PT, spp=5, max_bounces=6, seed=77, flags=T.FLAG_NO_JITTER)
    o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=5, max_bounces=6, seed=77, flags=T.FLAG_NO_JITTER, nthreads=8)
    handed = []
    for cap in (None, "5", "0"):
        if cap is not None:
            monkeypatch.setenv("RTAMD_POOL_PATH_CAP", cap)
        g = gpu_pass.Execute(frame, w, h, out_rgba=True, **kw)
        if cap is not None:
            monkeypatch.delenv("RTAMD_POOL_PATH_CAP")
        assert _same(g["rgba"], o["rgba"]), _report("path cap %s" % cap, g, o)
        assert g["rays"] == o["counters"]["rays"]
        handed.append(g["handover_pixels"])
    # paths with room in the list, whole pixels without: at least one of the five samples of every such pixel
    assert handed[2] >= 100 and handed[2] <= handed[0] <= 5 * handed[2], handed
    assert handed[2] <= handed[1] <= handed[2] + 5, handed


def test_random_frame_sizes_and_parameters(gpu_pass, oracle):
    """Seeded random combinations of odd frame sizes, tile rectangles, shard splits, sample counts and
    offsets, bounce limits and flags, in both modes, against the oracle with the same parameters
    (pixels outside the rectangle / shard keep the sentinel the buffer was filled with)."""
    from raytracing_amd import scenes
    rs = np.random.RandomState(77)
    for case in range(24):
        w, h = int(rs.randint(1, 130)), int(rs.randint(1, 80))
        buf, frame, sky = scenes.few_spheres(w, h) if case % 2 == 0 else scenes.random_spheres(w, h)
        gpu_pass.SetScene(buf)
        gpu_pass.SetSkyBox(sky if case % 5 else None)
        mode = T.MODE_PT if case % 3 else T.MODE_REF
        kw = dict(mode=mode)
        if mode == T.MODE_PT:
            kw.update(spp=int(rs.randint(1, 6)), max_bounces=int(rs.randint(0, 5)), seed=int(rs.randint(0, 2 ** 31)),
                      sample_offset=int(rs.randint(0, 100)))
            flags = 0
            if rs.rand() < 0.3:
                flags |= T.FLAG_NO_JITTER
            if rs.rand() < 0.3:
                flags |= T.FLAG_OUTPUT_SUM
            kw["flags"] = flags
        tile = shard = None
        if rs.rand() < 0.5:
            tx, ty = int(rs.randint(0, w)), int(rs.randint(0, h))
            tile = (tx, ty, int(rs.randint(1, w - tx + 8)), int(rs.randint(1, h - ty + 8)))
            kw.update(tile_x=tile[0], tile_y=tile[1], tile_w=tile[2], tile_h=tile[3])
        if rs.rand() < 0.4:
            n = int(rs.randint(2, 5))
            shard = (int(rs.randint(0, n)), n)
            kw.update(shard_rank=shard[0], shard_count=shard[1])
        g_rgba = np.full((h, w, 4), 7.0, dtype=np.float32)
        g_ids = np.full((h, w, 2), -9, dtype=np.int32)
        g = gpu_pass.Execute(frame, w, h, out_rgba=g_rgba, out_prim_id=g_ids, **kw)
        o_rgba = np.full((h, w, 4), 7.0, dtype=np.float32)
        o_ids = np.full((h, w, 2), -9, dtype=np.int32)
        o = oracle.render(buf, frame, w, h, sky=sky if case % 5 else None, mode=mode, spp=kw.get("spp", 1),
                          max_bounces=kw.get("max_bounces", 0), seed=kw.get("seed", 0), sample_offset=kw.get("sample_offset", 0),
                          flags=kw.get("flags", 0), tile=tile, shard=shard, rgba=o_rgba, ids=o_ids, want_t=False)
        what = (case, w, h, kw)
        assert _same(g_ids, o_ids), what
        assert _same(g_rgba, o_rgba), what
        assert g["rays"] == o["counters"]["rays"], what


@pytest.mark.parametrize("coord", [1e3, 1e4, 1e5])
@pytest.mark.parametrize("inst_scale", [1e-3, 1.0, 1e3])
def test_slivers_far_from_the_origin_under_scaled_instances(gpu_pass, oracle, coord, inst_scale):
    """The production kernel culls boxes against the closest hit times (1 + 2^-8) and hands a segment to
    the reference-order kernel when a second hit lies before the winner's own box entry.  That rests on
    a triangle's computed hit distance and its leaf-slot box's computed entry distance agreeing to
    2^-8 — in exact arithmetic the entry never exceeds the hit distance — which binary32 cancellation
    strains when a mesh sits 1e3 .. 1e5 away from its own origin: slivers of aspect 1e4 .. 3e5 there,
    under instances scaled 1e-3 .. 1e3, grazing and axis-aligned views from two distances.  Every id, t
    and radiance value must still equal the oracle's; the oracle reports the largest disagreement any
    accepted hit showed (1.4 x 2^-12 is the worst these scenes reach, 0.007 x 2^-12 the benchmark scenes)
    and the test fails when a scene comes within a factor of 4 of the slack."""
    from raytracing_amd import scenes
    w, h = 160, 90
    worst = 0.0
    for distance in (8.0, 3.0):
        for seed in range(4):
            buf, frame, sky = scenes.sliver_scene(w, h, seed, coord, inst_scale, distance=distance)
            gpu_pass.SetScene(buf)
            gpu_pass.SetSkyBox(sky)
            g = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, out_t=True, mode=T.MODE_PT, spp=3, max_bounces=5, seed=seed)
            o = oracle.render(buf, frame, w, h, sky=sky, mode=1, spp=3, max_bounces=5, seed=seed, nthreads=8)
            worst = max(worst, oracle.last_max_entry_gap())
            what = (coord, inst_scale, distance, seed, g["handover_pixels"])
            assert _same(g["ids"], o["ids"]), (_report("slivers", g, o), what)
            assert _same(g["t"], o["t"]), what
            assert _same(g["rgba"], o["rgba"]), (_report("slivers", g, o), what)
            assert g["rays"] == o["counters"]["rays"], what
            assert (o["ids"][..., 0] >= 0).mean() > 0.5
    assert worst < 2.0 ** -10, worst

"""rtamd_set_blas_build(RTAMD_BLAS_BUILD_DEVICE): the library's private trees over the meshes built on the GPU at
upload instead of by the host's SAH builder.  The hierarchy only orders the tests — the leaf-slot boxes and the
visiting order of the caller's trees (RayTracingShader.compute:293-345) decide hits and ties — so every output must
equal, bit for bit, what the host-built trees give (which the parity tests compare with the oracle)."""
import numpy as np
import pytest

from raytracing_amd import types as T

pytestmark = pytest.mark.gpu


def _render(p, frame, w, h, **kw):
    import torch
    rgba = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda:0")
    ids = torch.full((h, w, 2), -9, dtype=torch.int32, device="cuda:0")
    t = torch.zeros((h, w), dtype=torch.float32, device="cuda:0")
    r = p.Execute(frame, w, h, out_rgba=rgba, out_prim_id=ids, out_t=t, **kw)
    return rgba.cpu().numpy(), ids.cpu().numpy(), t.cpu().numpy(), r["rays"]


@pytest.fixture(autouse=True)
def _host_builder_afterwards(gpu_pass):
    """(the context is shared by the whole session: the tests that follow get the default builder back)"""
    yield
    gpu_pass.SetBlasBuild("auto")


def _scene(name, w, h):
    from raytracing_amd import scenes
    if name == "soup":
        return scenes.random_soup(w, h, seed=11, meshes=5, tris_per_mesh=2400, instances=30)
    if name == "spheres":
        return scenes.random_spheres(w, h, grid=6)
    if name == "mesh":
        return scenes.mesh_scene(w, h, target_tris=60_000)
    if name == "instanced":
        return scenes.instanced_scene(w, h, instances=40, tris_per_blas=3000)
    raise KeyError(name)


@pytest.mark.parametrize("scene", ["soup", "spheres", "mesh", "instanced"])
def test_device_built_trees_render_the_same_bits(gpu_pass, scene):
    w, h = 320, 180
    buf, frame, sky = _scene(scene, w, h)
    gpu_pass.SetSkyBox(sky)
    out = {}
    for who in ("host", "device"):
        gpu_pass.SetBlasBuild(who)
        gpu_pass.SetScene(buf)
        st = gpu_pass.UploadStats()
        assert (st["meshes_built_on_device"] > 0) == (who == "device"), st
        assert st["upload_ms"] >= st["tree_build_ms"] > 0
        out[who] = (_render(gpu_pass, frame, w, h, mode=T.MODE_REF),
                    _render(gpu_pass, frame, w, h, mode=T.MODE_PT, spp=4, max_bounces=5, seed=3))
    for a, b in zip(out["host"], out["device"]):
        for x, y in zip(a[:3], b[:3]):
            assert np.array_equal(x, y, equal_nan=True)
        assert a[3] == b[3]


def test_device_built_trees_equal_the_oracle(gpu_pass, oracle):
    from raytracing_amd import scenes
    w, h = 160, 90
    buf, frame, sky = scenes.random_soup(w, h, seed=4, meshes=3, tris_per_mesh=1500, instances=12)
    gpu_pass.SetBlasBuild("device")
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    assert gpu_pass.UploadStats()["meshes_built_on_device"] > 0
    got = _render(gpu_pass, frame, w, h, mode=T.MODE_PT, spp=2, max_bounces=4, seed=9)
    want = oracle.render(buf, frame, w, h, sky=sky, mode=T.MODE_PT, spp=2, max_bounces=4, seed=9, nthreads=8)
    assert np.array_equal(got[1], want["ids"])
    assert np.array_equal(got[2], want["t"], equal_nan=True)
    assert np.array_equal(got[0], want["rgba"], equal_nan=True)


@pytest.mark.parametrize("n_tris", [256, 257, 1023])
def test_odd_sizes_and_coinciding_triangles(gpu_pass, n_tris):
    """Meshes at the size from which the GPU builds, with an odd triangle count (the last leaf holds one triangle), and
    with many triangles on one spot (equal Morton codes: told apart by their sorted position)."""
    from raytracing_amd import host as H, scenes
    w, h = 200, 120
    rs = np.random.RandomState(n_tris)
    nv = n_tris
    pos = rs.normal(0, 1, size=(nv, 3)).astype(np.float32)
    idx = rs.randint(0, nv, size=(n_tris, 3)).astype(np.int32)
    idx[: n_tris // 3] = idx[0]                      # a third of the triangles are one and the same
    nrm = pos / np.maximum(np.linalg.norm(pos, axis=1, keepdims=True), 1e-6)
    sys = H.BuildAccelerateStructureSystem()
    b = sys.BakeBLAS(H.Mesh(pos, nrm.astype(np.float32), idx), 0)
    for k in range(3):
        sys.AddEntity(b, H.trs((2.5 * (k - 1), 0.0, 0.0), H.quat_axis_angle((0, 1, 0), 30.0 * k), (1, 1, 1)),
                      albedo=(0.8, 0.7, 0.6), specular=(0.1, 0.1, 0.1))
    buf = sys.OnUpdate()
    sys.close()
    frame = scenes.look_at_camera((0.0, 1.0, -8.0), (0.0, 0.0, 0.0), 45.0, w, h)
    gpu_pass.SetSkyBox(scenes.procedural_sky())
    out = {}
    for who in ("host", "device"):
        gpu_pass.SetBlasBuild(who)
        gpu_pass.SetScene(buf)
        out[who] = _render(gpu_pass, frame, w, h, mode=T.MODE_PT, spp=3, max_bounces=4, seed=1)
    assert gpu_pass.UploadStats()["meshes_built_on_device"] == 1
    for x, y in zip(out["host"][:3], out["device"][:3]):
        assert np.array_equal(x, y, equal_nan=True)
    assert (out["host"][1][..., 0] >= 0).any()


def test_unknown_mode_is_refused(gpu_pass):
    from raytracing_amd import backend
    with pytest.raises(backend.BackendError) as err:
        gpu_pass._check(backend.lib().rtamd_set_blas_build(gpu_pass._ctx, 7))
    assert err.value.code == T.ERR_INVALID_ARG

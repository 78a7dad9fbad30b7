"""RTAMD_FLAG_DEBUG_CHECKS: the bounds-checked variant of the traversal kernel.  On an intact scene it renders the
same bits as the production kernels; when a child word of the tree names a node, an instance or a triangle range
that does not exist (RTAMD_TEST_DAMAGE, applied for the checked launch only) the call returns RTAMD_ERR_BAD_INDEX
and follows nothing.  (The reference has no counterpart: its shader, RayTracingPass.cs:62-109, trusts the buffers.)"""
import os

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


@pytest.mark.parametrize("mode", ["ref", "pt"])
@pytest.mark.parametrize("scene", ["soup", "spheres"])
def test_checked_variant_renders_the_same_bits(gpu_pass, mode, scene):
    from raytracing_amd import scenes
    w, h = 320, 180
    buf, frame, sky = scenes.random_soup(w, h, seed=3) if scene == "soup" else scenes.random_spheres(w, h, grid=6)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    kw = dict(mode=T.MODE_REF) if mode == "ref" else dict(mode=T.MODE_PT, spp=4, max_bounces=5, seed=23)
    want = _render(gpu_pass, frame, w, h, **kw)
    got = _render(gpu_pass, frame, w, h, flags=T.FLAG_DEBUG_CHECKS, **kw)
    for a, b in zip(want[:3], got[:3]):
        assert np.array_equal(a, b, equal_nan=True)
    assert want[3] == got[3]


@pytest.mark.parametrize("what", ["node", "inst", "tri"])
@pytest.mark.parametrize("mode", ["ref", "pt"])
def test_damaged_tree_is_reported_not_followed(gpu_pass, what, mode):
    from raytracing_amd import backend, scenes
    w, h = 320, 180
    buf, frame, sky = scenes.random_soup(w, h, seed=3)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    kw = dict(mode=T.MODE_REF) if mode == "ref" else dict(mode=T.MODE_PT, spp=2, max_bounces=4, seed=5)
    want = _render(gpu_pass, frame, w, h, **kw)
    os.environ["RTAMD_TEST_DAMAGE"] = what
    try:
        with pytest.raises(backend.BackendError) as err:
            _render(gpu_pass, frame, w, h, flags=T.FLAG_DEBUG_CHECKS, **kw)
        assert err.value.code == T.ERR_BAD_INDEX
    finally:
        del os.environ["RTAMD_TEST_DAMAGE"]
    # the word was put back behind the checked launch: the production kernels render the scene as before
    got = _render(gpu_pass, frame, w, h, **kw)
    for a, b in zip(want[:3], got[:3]):
        assert np.array_equal(a, b, equal_nan=True)


def test_checked_variant_is_refused_where_it_does_not_exist(gpu_pass):
    from raytracing_amd import backend, scenes
    w, h = 64, 64
    buf, frame, sky = scenes.few_spheres(w, h)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(sky)
    with pytest.raises(backend.BackendError) as err:
        _render(gpu_pass, frame, w, h, mode=T.MODE_PT, spp=1, max_bounces=2, flags=T.FLAG_DEBUG_CHECKS | T.FLAG_DIELECTRICS)
    assert err.value.code == T.ERR_INVALID_ARG

"""The reference's own sample scene and camera (Assets/Scenes/SubScenes/RayTracingSubScene.unity:151,
219-221,305-307,416-418,567-569; Assets/Scenes/SampleScene.unity:638-640,666-667; mesh
Assets/Models/peter.dae) as committed fixtures: the meshes as the COLLADA loader reads them, the
flattened buffers of the scene, and the oracle's renders of it (tests/golden/make_golden.py wrote
them where the reference checkout exists).  The reference holds no rendered image of this scene,
so the expected outputs are the oracle's: parity with the reference itself stays unpinned."""
import json
import os

import numpy as np
import pytest

from raytracing_amd import host as H, scenes, types as T

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REFERENCE_MODELS = "/root/reference/Assets/Models"


@pytest.fixture(scope="module")
def dump():
    return H.SceneBuffers.load_file(os.path.join(GOLDEN, "reference_scene.rtscene"))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "reference_scene_640x360.npz"), allow_pickle=False)


def test_committed_model_arrays_are_what_the_loader_reads():
    if not os.path.exists(os.path.join(REFERENCE_MODELS, "peter.dae")):
        pytest.skip("no reference checkout here")
    for name, nv, nt in (("peter", 20013, 40018), ("bunny", 2503, 4968), ("cow", 2930, 5856)):   # peter.dae:56,78 bunny.dae:42,64 cow.dae:42,64
        a = scenes.load_model(name)                       # committed arrays
        b = scenes.load_model(name, REFERENCE_MODELS)     # the .dae itself
        assert a.positions.shape == (nv, 3) and a.indices.shape == (nt, 3)
        assert a.positions.tobytes() == b.positions.tobytes() and a.normals.tobytes() == b.normals.tobytes()
        assert a.indices.tobytes() == b.indices.tobytes()


def test_scene_built_from_the_model_arrays_equals_the_dump(dump):
    buf, frame, _ = scenes.reference_scene(None, 640, 360)
    for f in H.SceneBuffers.FIELDS:
        assert getattr(buf, f).tobytes() == getattr(dump, f).tobytes(), f
    assert len(dump.objects) == 4 and dump.n_blas == 2 and len(dump.triangles) == 40018 + 2


def test_bunny_bvh_dump(oracle):
    nodes = np.load(os.path.join(GOLDEN, "bvh_bunny.npy"))
    assert nodes.dtype == T.NODE
    bunny = scenes.load_model("bunny")
    aabbs, pts = bunny.triangle_aabbs()
    assert H.BoundingVolumeHierarchy.Build(pts, aabbs).tobytes() == nodes.tobytes()
    assert oracle.check_integrity(nodes, 4968) == 0
    leaves = nodes[1:][nodes[1:]["Flags"] != 0]["Data"].ravel()
    assert sorted(leaves[leaves >= 0].tolist()) == list(range(4968))       # every triangle exactly once


def test_oracle_renders_of_the_dump_match_the_fixture(oracle, dump, golden):
    summary = json.load(open(os.path.join(GOLDEN, "summary.json")))["reference_scene_640x360"]
    sky = scenes.procedural_sky()
    ref = oracle.render(dump, golden["frame"], 640, 360, sky=sky, mode=0, nthreads=8)
    assert np.array_equal(ref["ids"], golden["ids"]) and np.array_equal(ref["t"], golden["t"])
    assert np.array_equal(ref["rgba"], golden["ref_rgba"])
    assert ref["counters"] == summary["ref_counters"]
    pw, ph = summary["pt_size"]
    pt = oracle.render(dump, golden["pt_frame"], pw, ph, sky=sky, mode=1, want_ids=False, want_t=False, nthreads=8, **summary["pt"])
    assert np.array_equal(pt["rgba"], golden["pt_rgba"])
    assert pt["counters"] == summary["pt_counters"]
    # the four objects are all in view: peter (0) and the three walls
    assert set(np.unique(golden["ids"][..., 0]).tolist()) == {-1, 0, 1, 2, 3}
    assert 0.2 < (golden["ids"][..., 0] >= 0).mean() < 0.5


def test_oracle_traversal_equals_brute_force_on_the_reference_scene(oracle, dump):
    w, h = 128, 72
    _, frame, _ = scenes.reference_scene(None, w, h)
    o = oracle.render(dump, frame, w, h, mode=0, want_rgba=False)
    ids, t, ties = oracle.bruteforce_primary(dump, frame, w, h)
    clean = ties <= 1
    assert clean.mean() > 0.99
    assert np.array_equal(o["ids"][clean], ids[clean]) and np.array_equal(o["t"][clean], t[clean])
    assert (ids[..., 0] == 0).sum() > 200            # peter is hit


@pytest.mark.gpu
@pytest.mark.parametrize("ref_order", [False, True])
def test_gpu_renders_of_the_reference_scene_match_the_fixture(gpu_pass, dump, golden, ref_order):
    summary = json.load(open(os.path.join(GOLDEN, "summary.json")))["reference_scene_640x360"]
    sky = scenes.procedural_sky()
    gpu_pass.SetScene(dump)
    gpu_pass.SetSkyBox(sky)
    flags = T.FLAG_REFERENCE_ORDER if ref_order else 0
    g = gpu_pass.Execute(golden["frame"], 640, 360, out_rgba=True, out_prim_id=True, out_t=True, mode=T.MODE_REF, flags=flags)
    assert np.array_equal(g["ids"], golden["ids"])
    assert np.array_equal(g["t"], golden["t"])
    assert np.array_equal(g["rgba"], golden["ref_rgba"])
    assert g["rays"] == summary["ref_counters"]["rays"]
    pw, ph = summary["pt_size"]
    p = gpu_pass.Execute(golden["pt_frame"], pw, ph, out_rgba=True, mode=T.MODE_PT, flags=flags, **summary["pt"])
    assert np.array_equal(p["rgba"], golden["pt_rgba"])
    assert p["rays"] == summary["pt_counters"]["rays"]
    # the plain reference dispatch (no extension block) gives the same image
    out = np.zeros((360, 640, 4), dtype=np.float32)
    gpu_pass.ExecuteDefault(golden["frame"], 640, 360, out)
    assert np.array_equal(out, golden["ref_rgba"])

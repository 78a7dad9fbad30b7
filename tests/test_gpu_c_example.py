"""examples/render_scene.c: the C ABI used from plain C (no Python, no torch in the process) on the reference's own
sample scene (Assets/Scenes/SubScenes/RayTracingSubScene.unity flattened into tests/golden/reference_scene.rtscene,
sample camera of Assets/Scenes/SampleScene.unity:638-640,666-667).  Its image must be, bit for bit, the one the
Python binding renders and the one the golden fixture holds."""
import os
import subprocess

import numpy as np
import pytest

from raytracing_amd import types as T

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "_build", "render_scene")
SCENE = os.path.join(ROOT, "tests", "golden", "reference_scene.rtscene")


def _run(tmp_path, *args):
    out = str(tmp_path / "frame.ppm")
    r = subprocess.run([EXE, SCENE, out] + [str(a) for a in args], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return out, r.stdout


def test_c_program_renders_the_reference_image(gpu_pass, tmp_path):
    from raytracing_amd import host as H, scenes
    assert os.path.isfile(EXE), "python -m raytracing_amd.build compiles examples/render_scene.c"
    w, h = 640, 360
    out, text = _run(tmp_path, w, h)
    assert "reference image" in text and "rtamd 0.3 gfx950" in text
    rgba = np.fromfile(out + ".rgba", dtype=np.float32).reshape(h, w, 4)
    ids = np.fromfile(out + ".ids", dtype=np.int32).reshape(h, w, 2)
    buf = H.SceneBuffers.load_file(SCENE)
    _, frame, _ = scenes.reference_scene(None, w, h)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(None)          # (the C program uploads no sky: misses are black)
    r = gpu_pass.Execute(frame, w, h, out_rgba=True, out_prim_id=True, mode=T.MODE_REF)
    assert np.array_equal(ids, r["ids"])
    differ = ~((rgba == r["rgba"]) | (np.isnan(rgba) & np.isnan(r["rgba"])))
    assert not differ.any(), (int(differ.any(axis=2).sum()), rgba[differ.any(axis=2)][:3], r["rgba"][differ.any(axis=2)][:3], r["handover_pixels"])
    golden = np.load(os.path.join(ROOT, "tests", "golden", "reference_scene_640x360.npz"), allow_pickle=False)
    assert np.array_equal(ids, golden["ids"])
    # the image file: 8-bit, top row first
    with open(out, "rb") as f:
        assert f.readline().strip() == b"P6"
    assert (ids[..., 0] >= 0).mean() > 0.2          # (the sample camera sees the mesh and the walls on a quarter of the frame)


def test_c_program_path_traces(gpu_pass, tmp_path):
    from raytracing_amd import host as H, scenes
    w, h = 320, 180
    out, text = _run(tmp_path, w, h, 4)
    assert "path traced" in text
    rgba = np.fromfile(out + ".rgba", dtype=np.float32).reshape(h, w, 4)
    buf = H.SceneBuffers.load_file(SCENE)
    _, frame, _ = scenes.reference_scene(None, w, h)
    gpu_pass.SetScene(buf)
    gpu_pass.SetSkyBox(None)
    r = gpu_pass.Execute(frame, w, h, out_rgba=True, mode=T.MODE_PT, spp=4, max_bounces=8, seed=1)
    assert np.array_equal(rgba, r["rgba"], equal_nan=True)

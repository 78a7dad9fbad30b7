import os
import sys

import pytest

# torch brings its own copy of the HIP runtime: imported before librtamd.so is loaded, both use that one copy;
# the other way round torch finds a second runtime in the process and sees no GPU.  Some GPU tests hand torch
# tensors to the library, so every run of any subset of the tests imports torch first.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def scene_few():
    from raytracing_amd import scenes
    return scenes.few_spheres(64, 36)


@pytest.fixture(scope="session")
def scene_random_small():
    from raytracing_amd import scenes
    return scenes.random_spheres(96, 54, grid=6)


@pytest.fixture(scope="session")
def scene_mesh_small():
    from raytracing_amd import scenes
    return scenes.mesh_scene(96, 54, target_tris=20000, procedural=True)


@pytest.fixture(scope="session")
def gpu_pass():
    from raytracing_amd import backend
    p = backend.RayTracingPass(0)
    yield p
    p.Dispose()

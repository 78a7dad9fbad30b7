"""Host-side scene producers: camera matrices, matrix inverse, mesh sources, the per-frame
flattening (BuildAccelerateStructureSystem.OnUpdate) and the scene file."""
import ctypes as C
import os

import numpy as np
import pytest

from raytracing_amd import host as H, scenes, types as T

MODELS = "/root/reference/Assets/Models"


def _m(a):
    return np.asarray(a, dtype=np.float64).reshape(4, 4).T      # column-major storage -> math matrix


def test_inverse_and_trs():
    m = H.trs((1, 2, 3), H.quat_axis_angle((1, 2, 3), 40), (2, 3, 4))
    inv = H.inverse4x4(m)
    assert np.allclose(_m(m) @ _m(inv), np.eye(4), atol=1e-5)
    with pytest.raises(H.HostError):
        H.inverse4x4(np.zeros(16))


def test_unity_camera_convention():
    # identity rotation: the camera looks down world +Z although camera space looks down -Z
    f = H.camera((1, 2, 3), (0, 0, 0, 1), 60.0, 16 / 9)
    c2w = _m(f["CameraToWorld"][0])
    assert np.allclose(c2w[:3, 3], [1, 2, 3])
    assert np.allclose(c2w[:3, :3], np.diag([1, 1, -1]))
    ip = _m(f["CameraInverseProjection"][0])
    d = ip @ np.array([0, 0, 0, 1.0])
    d = c2w[:3, :3] @ d[:3]
    assert np.allclose(d / np.linalg.norm(d), [0, 0, 1], atol=1e-6)
    # top-right corner of the image maps to +x,+y at tan(fov/2)
    d = ip @ np.array([1, 1, 0, 1.0])
    assert np.isclose(d[1] / -d[2], np.tan(np.radians(30)), rtol=1e-5)
    assert np.isclose(d[0] / -d[2], np.tan(np.radians(30)) * 16 / 9, rtol=1e-5)
    assert f["PixelOffset"][0].tolist() == [0, 0]                 # RayTracingPass.cs:76


def test_generated_meshes_face_outward():
    for mesh in (H.Mesh.uv_sphere(16, 8, 0.5), H.Mesh.cube()):
        p = mesh.positions[mesh.indices]
        n = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
        assert (np.einsum("ij,ij->i", n, p.mean(axis=1)) > 0).all()
    assert len(H.Mesh.uv_sphere(32, 16, 0.5).indices) == 960
    q = H.Mesh.quad()
    assert q.indices.tolist() == [[0, 3, 1], [3, 0, 2]] and (q.normals == [0, 0, -1]).all()


def test_subdivide_quadruples_triangles():
    m = H.Mesh.uv_sphere(8, 4, 0.5)
    s = m.subdivide(2)
    assert len(s.indices) == 16 * len(m.indices)
    area = lambda x: np.linalg.norm(np.cross(x.positions[x.indices][:, 1] - x.positions[x.indices][:, 0],
                                             x.positions[x.indices][:, 2] - x.positions[x.indices][:, 0]), axis=1).sum()
    assert np.isclose(area(m), area(s), rtol=1e-5)
    assert np.allclose(np.linalg.norm(s.normals, axis=1), 1, atol=1e-5)


@pytest.mark.skipif(not os.path.exists(MODELS), reason="reference models not present")
def test_collada_loader_on_the_reference_models():
    # triangle counts from the <triangles count=...> attributes (bunny.dae:64, cow.dae:64, peter.dae:78)
    for name, tris in (("bunny", 4968), ("cow", 5856), ("peter", 40018)):
        m = H.Mesh.load_dae(os.path.join(MODELS, name + ".dae"))
        assert len(m.indices) == tris
        assert m.indices.max() == len(m.positions) - 1 and m.indices.min() == 0
        assert np.allclose(np.linalg.norm(m.normals, axis=1), 1, atol=1e-3)
        # Z_UP asset rotated to Y-up: closed scanned meshes face outward on average
        p = m.positions[m.indices]
        n = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
        assert (np.einsum("ij,ij->i", n, m.normals[m.indices[:, 0]]) > 0).mean() > 0.99
    assert len(H.Mesh.load_dae(os.path.join(MODELS, "bunny.dae")).positions) == 2503      # bunny.dae:44


def test_on_update_flattening(oracle):
    sys = H.BuildAccelerateStructureSystem()
    sph = sys.BakeBLAS(H.Mesh.uv_sphere(8, 4, 0.5))
    quad = sys.BakeBLAS(H.Mesh.quad())
    unused = sys.BakeBLAS(H.Mesh.cube())                       # baked but never instanced
    sys.AddEntity(quad, H.trs((0, 0, 0)))
    sys.AddEntity(sph, H.trs((1, 0, 0)), albedo=(0.1, 0.2, 0.3))
    sys.AddEntity(sph, H.trs((2, 0, 0), s=(2, 2, 2)), emission=(1, 2, 3))
    b = sys.OnUpdate()
    assert unused == 2 and b.n_blas == 2                       # BLAS index by first use (:70-83)
    assert b.objects["BLASIndex"].tolist() == [0, 1, 1]
    assert b.blas_offsets.tolist() == [0, 3] and b.triangles_offsets.tolist() == [0, 2] and b.vertices_offsets.tolist() == [0, 4]
    assert len(b.tlas) == 3 + 128                              # BodyCount + 64 (TopLevelAccelerateStructure.cs:17) + room for the first levels
    assert np.allclose(b.objects["albedo"][1], [0.1, 0.2, 0.3, 1])
    assert np.allclose(b.objects["emission"][2], [1, 2, 3, 1])
    for o in b.objects:
        assert np.allclose(_m(o["LocalToWorld"]) @ _m(o["WorldToLocal"]), np.eye(4), atol=1e-5)
    assert oracle.check_integrity(b.tlas, 3) == 0
    # world AABB of the scaled sphere (PrepareBuildTLASJob :215-232): centre (2,0,0), radius 1
    leaf = b.tlas[b.tlas["Flags"] != 0][0]
    slot = leaf["Data"].tolist().index(2)
    assert np.isclose(leaf["Lx"][slot], 1.0, atol=1e-5) and np.isclose(leaf["Hx"][slot], 3.0, atol=1e-5)
    # moving an object and updating again changes only TLAS + objects
    sys.SetLocalToWorld(1, H.trs((5, 0, 0)))
    b2 = sys.OnUpdate()
    assert b2.blas.tobytes() == b.blas.tobytes() and b2.objects.tobytes() != b.objects.tobytes()
    sys.close()


def test_empty_world_is_an_error():
    sys = H.BuildAccelerateStructureSystem()
    with pytest.raises(H.HostError):
        sys.OnUpdate()                                         # ObjectsCount == 0 (:53)
    sys.close()


def test_scene_file_round_trip(tmp_path, scene_few):
    buf, _, _ = scene_few
    hb = T.HostBuffers()
    keep = []
    for name, cname, cnt in (("tlas", "tlas", "n_tlas_nodes"), ("blas", "blas", "n_blas_nodes"), ("vertices", "vertices", "n_vertices"),
                             ("triangles", "triangles", "n_triangles"), ("objects", "objects", "n_objects")):
        arr = getattr(buf, name)
        keep.append(arr)
        setattr(hb, cname, arr.ctypes.data)
        setattr(hb, cnt, len(arr))
    for name in ("blas_offsets", "vertices_offsets", "triangles_offsets"):
        setattr(hb, name, getattr(buf, name).ctypes.data)
    hb.n_blas = buf.n_blas
    path = str(tmp_path / "scene.rtamd").encode()
    assert H.lib().rtamd_host_scene_save(C.byref(hb), path) == 0
    s = H.lib().rtamd_host_scene_create()
    assert H.lib().rtamd_host_scene_load(s, path) == 0
    out = T.HostBuffers()
    assert H.lib().rtamd_host_scene_buffers(s, C.byref(out)) == 0
    assert out.n_triangles == len(buf.triangles) and out.n_objects == len(buf.objects)
    got = (C.c_char * buf.blas.nbytes).from_address(out.blas)
    assert bytes(got) == buf.blas.tobytes()
    H.lib().rtamd_host_scene_destroy(s)
    assert H.lib().rtamd_host_scene_load(H.lib().rtamd_host_scene_create(), b"/nonexistent/file") < 0


def test_config_scenes_have_the_documented_shape():
    buf, frame, sky = scenes.random_spheres(64, 36)
    assert len(buf.objects) == 485 and buf.n_blas == 2
    assert sky.shape == (128, 256, 4)


def test_tlas_builder_choice_and_the_reference_builders_limit(oracle):
    """The reference's TLAS builder overruns its 64-entry level arrays for some inputs of
    more than about 1 300 objects; the producer refuses those instead, and offers the SAH
    builder for any count."""
    from raytracing_amd import scenes
    with pytest.raises(H.HostError, match="64-entry"):
        scenes.instanced_scene(32, 18, instances=1600, tris_per_blas=600)
    buf, _, _ = scenes.instanced_scene(32, 18, instances=1600, tris_per_blas=600, tlas_tree=1)
    assert len(buf.objects) == 1601
    assert oracle.check_integrity(buf.tlas, len(buf.objects)) == 0
    buf, _, _ = scenes.instanced_scene(32, 18, instances=1250, tris_per_blas=600)
    assert oracle.check_integrity(buf.tlas, len(buf.objects)) == 0

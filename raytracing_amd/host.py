"""Host side (CPU): thin ctypes layer over librtamd_host.so — meshes, images, camera, the flattened-scene
container and file, the SAH tree writer — and over the restated reference producers, which live in a
library of their own (tests/producers -> librtamd_testhost.so: fixture infrastructure, loaded only when a
producer class is used; the package's own helpers never need it).

Class and method names of the producers mirror the reference's C# host code so that tests read like
tests of that code:

  BoundingVolumeHierarchy.Build / .ScheduleBuildJobs   Assets/Script/BVH/BoundingVolumeHierarchyBuilder.cs:482-558
  BLASBaker-style bake, entities, OnUpdate              Assets/Editor/RayTracing/Authoring/RayTraceableBaker.cs:75-149,
                                                        Assets/Script/RayTracing/System/BuildAccelerateStructureSystem.cs:41-163
"""
import ctypes as C
import os

import numpy as np

from . import types as T

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librtamd_host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run `python -m raytracing_amd.build` first")
        L = C.CDLL(_LIB_PATH)
        L.rtamd_host_last_error.restype = C.c_char_p
        L.rtamd_host_scene_create.restype = C.c_void_p
        L.rtamd_host_scene_destroy.argtypes = [C.c_void_p]
        L.rtamd_host_scene_buffers.argtypes = [C.c_void_p, C.POINTER(T.HostBuffers)]
        L.rtamd_host_bvh_build_sah.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.rtamd_host_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.rtamd_host_inverse4x4.argtypes = [C.c_void_p, C.c_void_p]
        L.rtamd_host_trs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rtamd_host_mesh_free.argtypes = [C.POINTER(T.HostMesh)]
        L.rtamd_host_mesh_uv_sphere.argtypes = [C.c_int32, C.c_int32, C.c_float, C.POINTER(T.HostMesh)]
        L.rtamd_host_mesh_quad.argtypes = [C.POINTER(T.HostMesh)]
        L.rtamd_host_mesh_cube.argtypes = [C.POINTER(T.HostMesh)]
        L.rtamd_host_mesh_load_dae.argtypes = [C.c_char_p, C.POINTER(T.HostMesh)]
        L.rtamd_host_mesh_subdivide.argtypes = [C.POINTER(T.HostMesh), C.c_int32, C.POINTER(T.HostMesh)]
        L.rtamd_host_scene_save.argtypes = [C.POINTER(T.HostBuffers), C.c_char_p]
        L.rtamd_host_scene_load.argtypes = [C.c_void_p, C.c_char_p]
        _lib = L
    return _lib


_PRODUCERS_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "producers", "_build",
                               "librtamd_testhost.so")
_producers = None


def producers():
    """The restated reference producers (BVH builder, bake, per-frame flattening): test and fixture
    infrastructure, not part of the package's own library."""
    global _producers
    if _producers is None:
        lib()           # its scene container, SAH writer and error text are what the producers build on
        if not os.path.exists(_PRODUCERS_PATH):
            raise RuntimeError(f"{_PRODUCERS_PATH} is missing: the reference's scene producers are built from tests/producers "
                               "by `python -m raytracing_amd.build`")
        L = C.CDLL(_PRODUCERS_PATH)
        L.rtamd_host_bake_blas.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
        L.rtamd_host_add_entity.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rtamd_host_set_entity_transform.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.rtamd_host_scene_update.argtypes = [C.c_void_p]
        L.rtamd_host_scene_set_tlas_builder.argtypes = [C.c_void_p, C.c_int32]
        L.rtamd_host_bvh_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        L.rtamd_host_bvh_build_jobs.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
        _producers = L
    return _producers


class HostError(RuntimeError):
    pass


def _check(rc, what):
    if rc < 0:
        raise HostError(f"{what}: {T.ERRORS.get(rc, rc)}: {lib().rtamd_host_last_error().decode()}")
    return rc


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} floats, got {a.size}")
    return a


class Mesh:
    """positions (n,3) f32, normals (n,3) f32, indices (m,3) i32."""

    def __init__(self, positions, normals, indices):
        self.positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        self.normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1, 3)

    @staticmethod
    def _take(hm):
        nv, nt = hm.n_vertices, hm.n_triangles
        m = Mesh(np.ctypeslib.as_array(hm.positions, (nv * 3,)).copy(),
                 np.ctypeslib.as_array(hm.normals, (nv * 3,)).copy(),
                 np.ctypeslib.as_array(hm.indices, (nt * 3,)).copy())
        lib().rtamd_host_mesh_free(C.byref(hm))
        return m

    def _as_host(self):
        hm = T.HostMesh()
        hm.positions = self.positions.ctypes.data_as(C.POINTER(C.c_float))
        hm.normals = self.normals.ctypes.data_as(C.POINTER(C.c_float))
        hm.indices = self.indices.ctypes.data_as(C.POINTER(C.c_int32))
        hm.n_vertices = len(self.positions)
        hm.n_triangles = len(self.indices)
        return hm

    @staticmethod
    def uv_sphere(segments=32, rings=16, radius=0.5):
        hm = T.HostMesh()
        _check(lib().rtamd_host_mesh_uv_sphere(segments, rings, radius, C.byref(hm)), "uv_sphere")
        return Mesh._take(hm)

    @staticmethod
    def quad():
        hm = T.HostMesh()
        _check(lib().rtamd_host_mesh_quad(C.byref(hm)), "quad")
        return Mesh._take(hm)

    @staticmethod
    def cube():
        hm = T.HostMesh()
        _check(lib().rtamd_host_mesh_cube(C.byref(hm)), "cube")
        return Mesh._take(hm)

    @staticmethod
    def load_dae(path):
        hm = T.HostMesh()
        _check(lib().rtamd_host_mesh_load_dae(path.encode(), C.byref(hm)), "load_dae")
        return Mesh._take(hm)

    def subdivide(self, rounds):
        src = self._as_host()
        hm = T.HostMesh()
        _check(lib().rtamd_host_mesh_subdivide(C.byref(src), rounds, C.byref(hm)), "subdivide")
        return Mesh._take(hm)

    def triangle_aabbs(self):
        """Per-triangle AABB and centre as the baker computes them (RayTraceableBaker.cs:116-126)."""
        p = self.positions[self.indices]          # (m,3,3)
        aabbs = np.zeros(len(self.indices), dtype=T.AABB)
        aabbs["Min"] = p.min(axis=1)
        aabbs["Max"] = p.max(axis=1)
        pts = np.zeros(len(self.indices), dtype=T.POINT_AND_INDEX)
        pts["Position"] = (aabbs["Max"] + aabbs["Min"]) * np.float32(0.5)
        pts["Index"] = np.arange(len(self.indices), dtype=np.int32)
        return aabbs, pts


class BoundingVolumeHierarchy:
    """The 4-wide BVH builder (BoundingVolumeHierarchyBuilder.cs)."""

    @staticmethod
    def Build(points, aabbs):
        """Single-threaded median build (:529-558). Returns nodes[:nodeCount]; `points` is permuted."""
        n = len(aabbs)
        nodes = np.zeros(n + 64, dtype=T.NODE)
        cnt = C.c_int32(0)
        _check(producers().rtamd_host_bvh_build(_ptr(points), _ptr(aabbs), n, _ptr(nodes), len(nodes), C.byref(cnt)), "Build")
        return nodes[:cnt.value].copy()

    @staticmethod
    def ScheduleBuildJobs(points, aabbs):
        """The three-job build used for the TLAS (:482-527). Returns (nodes, branchCount).  The node
        array is BodyCount + 128 long: the reference's BodyCount + 64 (TopLevelAccelerateStructure.cs:17)
        overflows when the first levels alone take 86 nodes."""
        n = len(aabbs)
        nodes = np.zeros(n + 128, dtype=T.NODE)
        bc = C.c_int32(0)
        _check(producers().rtamd_host_bvh_build_jobs(_ptr(points), _ptr(aabbs), n, _ptr(nodes), len(nodes), C.byref(bc)), "ScheduleBuildJobs")
        return nodes, bc.value

    @staticmethod
    def BuildSah(aabbs):
        n = len(aabbs)
        nodes = np.zeros(n + 64, dtype=T.NODE)
        cnt = C.c_int32(0)
        _check(lib().rtamd_host_bvh_build_sah(_ptr(aabbs), n, _ptr(nodes), len(nodes), C.byref(cnt)), "BuildSah")
        return nodes[:cnt.value].copy()


class SceneBuffers:
    """The eight buffers BuildAccelerateStructureSystem uploads (numpy copies)."""

    FIELDS = ("tlas", "blas_offsets", "blas", "vertices_offsets", "vertices", "triangles_offsets", "triangles", "objects")

    def __init__(self, tlas, blas_offsets, blas, vertices_offsets, vertices, triangles_offsets, triangles, objects,
                 tlas_branch_count=0):
        self.tlas = np.ascontiguousarray(tlas, dtype=T.NODE)
        self.blas_offsets = np.ascontiguousarray(blas_offsets, dtype=np.int32)
        self.blas = np.ascontiguousarray(blas, dtype=T.NODE)
        self.vertices_offsets = np.ascontiguousarray(vertices_offsets, dtype=np.int32)
        self.vertices = np.ascontiguousarray(vertices, dtype=T.VERTEX)
        self.triangles_offsets = np.ascontiguousarray(triangles_offsets, dtype=np.int32)
        self.triangles = np.ascontiguousarray(triangles, dtype=T.INT3)
        self.objects = np.ascontiguousarray(objects, dtype=T.OBJECT)
        self.tlas_branch_count = tlas_branch_count

    @property
    def n_blas(self):
        return len(self.blas_offsets)

    def nbytes(self):
        return sum(getattr(self, f).nbytes for f in self.FIELDS)

    def blas_slice(self, b):
        """(nodes, triangle count) of BLAS b."""
        n0 = self.blas_offsets[b]
        n1 = self.blas_offsets[b + 1] if b + 1 < self.n_blas else len(self.blas)
        t0 = self.triangles_offsets[b]
        t1 = self.triangles_offsets[b + 1] if b + 1 < self.n_blas else len(self.triangles)
        return self.blas[n0:n1], int(t1 - t0)

    def save(self, path):
        np.savez_compressed(path, **{f: getattr(self, f) for f in self.FIELDS})

    def _as_host(self):
        hb = T.HostBuffers()
        hb.tlas, hb.n_tlas_nodes = self.tlas.ctypes.data, len(self.tlas)
        hb.blas_offsets, hb.blas, hb.n_blas_nodes = self.blas_offsets.ctypes.data, self.blas.ctypes.data, len(self.blas)
        hb.vertices_offsets, hb.vertices, hb.n_vertices = self.vertices_offsets.ctypes.data, self.vertices.ctypes.data, len(self.vertices)
        hb.triangles_offsets, hb.triangles, hb.n_triangles = self.triangles_offsets.ctypes.data, self.triangles.ctypes.data, len(self.triangles)
        hb.n_blas, hb.objects, hb.n_objects = self.n_blas, self.objects.ctypes.data, len(self.objects)
        hb.tlas_branch_count = self.tlas_branch_count
        return hb

    def save_file(self, path):
        """Flattened-scene file (rtamd_host_scene_save): the eight arrays behind a small header."""
        hb = self._as_host()
        _check(lib().rtamd_host_scene_save(C.byref(hb), os.fsencode(path)), "scene_save")

    @staticmethod
    def load_file(path):
        """Read a flattened-scene file back (rtamd_host_scene_load)."""
        s = lib().rtamd_host_scene_create()
        try:
            _check(lib().rtamd_host_scene_load(s, os.fsencode(path)), "scene_load")
            hb = T.HostBuffers()
            _check(lib().rtamd_host_scene_buffers(s, C.byref(hb)), "buffers")
            return _buffers_from_host(hb)
        finally:
            lib().rtamd_host_scene_destroy(s)

    @staticmethod
    def load(path):
        z = np.load(path, allow_pickle=False)
        return SceneBuffers(*[z[f] for f in SceneBuffers.FIELDS])


def _buffers_from_host(hb):
    """Numpy copies of the arrays a rtamd_host_buffers record points at."""
    def arr(ptr, n, dt):
        buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt, count=n).copy()
    return SceneBuffers(arr(hb.tlas, hb.n_tlas_nodes, T.NODE), arr(hb.blas_offsets, hb.n_blas, np.int32),
                        arr(hb.blas, hb.n_blas_nodes, T.NODE), arr(hb.vertices_offsets, hb.n_blas, np.int32),
                        arr(hb.vertices, hb.n_vertices, T.VERTEX), arr(hb.triangles_offsets, hb.n_blas, np.int32),
                        arr(hb.triangles, hb.n_triangles, T.INT3), arr(hb.objects, hb.n_objects, T.OBJECT),
                        hb.tlas_branch_count)


class BuildAccelerateStructureSystem:
    """Entities + the per-frame update that flattens them (BuildAccelerateStructureSystem.cs)."""

    MEDIAN, SAH = 0, 1

    def __init__(self, tlas_tree=MEDIAN):
        self._s = lib().rtamd_host_scene_create()
        if not self._s:
            raise MemoryError("rtamd_host_scene_create")
        self.ObjectsCount = 0
        if tlas_tree != self.MEDIAN:
            _check(producers().rtamd_host_scene_set_tlas_builder(self._s, tlas_tree), "set_tlas_builder")

    def close(self):
        if self._s:
            lib().rtamd_host_scene_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def BakeBLAS(self, mesh, tree=MEDIAN):
        """BLASBaker.BuildBLASBlobAssetReference: returns a blob handle; reuse it to instance the mesh."""
        return _check(producers().rtamd_host_bake_blas(self._s, _ptr(mesh.positions), _ptr(mesh.normals), len(mesh.positions),
                                                 _ptr(mesh.indices), len(mesh.indices), tree), "BakeBLAS")

    def AddEntity(self, blas, local_to_world, albedo=(1, 1, 1, 1), specular=(0, 0, 0, 1), emission=(0, 0, 0, 1)):
        def c4(v):
            v = list(v)
            return _f32(v + [1.0] * (4 - len(v)), 4)
        e = _check(producers().rtamd_host_add_entity(self._s, blas, _ptr(_f32(local_to_world, 16)), _ptr(c4(albedo)),
                                               _ptr(c4(specular)), _ptr(c4(emission))), "AddEntity")
        self.ObjectsCount = e + 1
        return e

    def SetLocalToWorld(self, entity, local_to_world):
        _check(producers().rtamd_host_set_entity_transform(self._s, entity, _ptr(_f32(local_to_world, 16))), "SetLocalToWorld")

    def OnUpdate(self):
        """Returns SceneBuffers (copies)."""
        _check(producers().rtamd_host_scene_update(self._s), "OnUpdate")
        hb = T.HostBuffers()
        _check(lib().rtamd_host_scene_buffers(self._s, C.byref(hb)), "buffers")
        return _buffers_from_host(hb)


def trs(t=(0, 0, 0), q=(0, 0, 0, 1), s=(1, 1, 1)):
    out = np.zeros(16, dtype=np.float32)
    lib().rtamd_host_trs(_ptr(_f32(t, 3)), _ptr(_f32(q, 4)), _ptr(_f32(s, 3)), _ptr(out))
    return out


def inverse4x4(m):
    out = np.zeros(16, dtype=np.float32)
    if lib().rtamd_host_inverse4x4(_ptr(_f32(m, 16)), _ptr(out)) != 0:
        raise HostError("singular matrix")
    return out


def quat_axis_angle(axis, degrees):
    a = np.asarray(axis, dtype=np.float64)
    a = a / np.linalg.norm(a)
    h = np.radians(degrees) / 2
    return np.array([*(a * np.sin(h)), np.cos(h)], dtype=np.float32)


def quat_look_rotation(forward, up=(0, 1, 0)):
    """Unity's Quaternion.LookRotation: +Z of the object points along `forward`."""
    f = np.asarray(forward, dtype=np.float64)
    f = f / np.linalg.norm(f)
    r = np.cross(np.asarray(up, dtype=np.float64), f)
    r = r / np.linalg.norm(r)
    u = np.cross(f, r)
    m = np.array([r, u, f]).T          # columns = right, up, forward
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
    return np.asarray(q, dtype=np.float32)


def camera(position, rotation_xyzw, fov_y_deg, aspect, z_near=0.3, z_far=1000.0):
    """FrameCBuffer (numpy record) for a Unity camera (RayTracingPass.cs:72-79)."""
    out = np.zeros(1, dtype=T.FRAME)
    lib().rtamd_host_camera(_ptr(_f32(position, 3)), _ptr(_f32(rotation_xyzw, 4)), fov_y_deg, aspect, z_near, z_far, _ptr(out))
    return out


def load_image(path, flip_y=True):
    """HDR image (.exr / .hdr) as float32 (h, w, 4).  flip_y=True gives Unity texture order
    (row 0 = bottom of the picture), the order RayTracingPass.SetSkyBox expects."""
    L = lib()
    L.rtamd_host_image_load.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rtamd_host_image_free.argtypes = [C.POINTER(C.c_float)]
    p = C.POINTER(C.c_float)()
    w, h = C.c_int32(), C.c_int32()
    _check(L.rtamd_host_image_load(os.fsencode(path), 1 if flip_y else 0, C.byref(p), C.byref(w), C.byref(h)), "image_load")
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    finally:
        L.rtamd_host_image_free(p)


def save_image(path, rgba, flip_y=True):
    """Write float32 (h, w, 4) as .exr (float, ZIP), .hdr (RGBE) or .ppm (8 bits, clamped)."""
    a = np.ascontiguousarray(rgba, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("expected an (h, w, 4) array")
    L = lib()
    L.rtamd_host_image_save.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    _check(L.rtamd_host_image_save(os.fsencode(path), _ptr(a), a.shape[1], a.shape[0], 1 if flip_y else 0), "image_save")

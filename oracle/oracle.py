"""ctypes wrapper of the CPU oracle (oracle/rt_oracle.cpp).

TEST INFRASTRUCTURE ONLY — imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py; never by the raytracing_amd package.  PARITY UNPINNED: the
reference holds no golden vectors and cannot be compiled here, so the oracle restates the
shader text with declared float semantics (see the header of rt_oracle.cpp).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "_build", "librt_oracle.so")
_SO_RECIPROCAL = os.path.join(_DIR, "_build", "librt_oracle_reciprocal.so")      # divisions as reciprocal multiplications (counting only)
_lib = None
_lib_reciprocal = None


class _Scene(C.Structure):
    _fields_ = [("tlas", C.c_void_p), ("nTlas", C.c_int32), ("blasOffsets", C.c_void_p), ("blas", C.c_void_p),
                ("nBlasNodes", C.c_int32), ("vtxOffsets", C.c_void_p), ("verts", C.c_void_p), ("nVerts", C.c_int32),
                ("triOffsets", C.c_void_p), ("tris", C.c_void_p), ("nTris", C.c_int32), ("nBlas", C.c_int32),
                ("objects", C.c_void_p), ("nObjects", C.c_int32), ("sky", C.c_void_p), ("skyW", C.c_int32),
                ("skyH", C.c_int32)]


class _Params(C.Structure):
    _fields_ = [("mode", C.c_int32), ("spp", C.c_int32), ("max_bounces", C.c_int32), ("seed", C.c_uint64),
                ("sample_offset", C.c_int32), ("tile_x", C.c_int32), ("tile_y", C.c_int32), ("tile_w", C.c_int32),
                ("tile_h", C.c_int32), ("shard_rank", C.c_int32), ("shard_count", C.c_int32), ("flags", C.c_uint32),
                ("nthreads", C.c_int32)]


class _Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes", C.c_uint64), ("tris", C.c_uint64), ("instances", C.c_uint64),
                ("hits", C.c_uint64), ("max_blas_stack", C.c_uint64), ("max_tlas_stack", C.c_uint64),
                ("overflow", C.c_uint64)]


def build():
    src = os.path.join(_DIR, "rt_oracle.cpp")
    if any(not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src) for so in (_SO, _SO_RECIPROCAL)):
        subprocess.run(["make", "-C", _DIR], check=True)
    return _SO


def lib_reciprocal():
    """The build with RT_ORACLE_RECIPROCAL_DIVISIONS: same entry points, reciprocal multiplications
    where the shader text has a `/` by a constant or per-frame value."""
    global _lib_reciprocal
    if _lib_reciprocal is None:
        if not os.path.exists(_SO_RECIPROCAL):
            build()
        _lib_reciprocal = C.CDLL(_SO_RECIPROCAL)
    return _lib_reciprocal


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.oracle_acos.restype = C.c_float
        L.oracle_acos.argtypes = [C.c_float]
        L.oracle_atan2.restype = C.c_float
        L.oracle_atan2.argtypes = [C.c_float, C.c_float]
        L.oracle_sincos2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_rng.restype = C.c_float
        L.oracle_rng.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_check_integrity.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _scene(b, sky):
    s = _Scene()
    s.tlas, s.nTlas = _p(b.tlas), len(b.tlas)
    s.blasOffsets = _p(b.blas_offsets)
    s.blas, s.nBlasNodes = _p(b.blas), len(b.blas)
    s.vtxOffsets = _p(b.vertices_offsets)
    s.verts, s.nVerts = _p(b.vertices), len(b.vertices)
    s.triOffsets = _p(b.triangles_offsets)
    s.tris, s.nTris = _p(b.triangles), len(b.triangles)
    s.nBlas = len(b.blas_offsets)
    s.objects, s.nObjects = _p(b.objects), len(b.objects)
    if sky is not None:
        s.sky, s.skyH, s.skyW = _p(sky), sky.shape[0], sky.shape[1]
    return s


def render(buffers, frame, width, height, sky=None, mode=0, spp=1, max_bounces=8, seed=0, sample_offset=0,
           tile=None, shard=None, flags=0, nthreads=1, want_rgba=True, want_ids=True, want_t=True,
           rgba=None, ids=None, t=None, reciprocal=False):
    """Returns dict(rgba, ids, t, counters, overflow).  reciprocal=True renders with the shader's divisions
    read as multiplications by rounded reciprocals instead of the declared true divisions."""
    sky = None if sky is None else np.ascontiguousarray(sky, dtype=np.float32)
    sc = _scene(buffers, sky)
    p = _Params()
    p.mode, p.spp, p.max_bounces, p.seed, p.sample_offset, p.flags, p.nthreads = mode, spp, max_bounces, seed, sample_offset, flags, nthreads
    if tile:
        p.tile_x, p.tile_y, p.tile_w, p.tile_h = tile
    if shard:
        p.shard_rank, p.shard_count = shard
    if rgba is None and want_rgba:
        rgba = np.zeros((height, width, 4), dtype=np.float32)
    if ids is None and want_ids:
        ids = np.full((height, width, 2), -2, dtype=np.int32)
    if t is None and want_t:
        t = np.zeros((height, width), dtype=np.float32)
    cnt = _Counters()
    frame = np.ascontiguousarray(frame)
    rc = (lib_reciprocal() if reciprocal else lib()).oracle_render(C.byref(sc), _p(frame), C.byref(p), width, height, _p(rgba), _p(ids), _p(t), C.byref(cnt))
    counters = {k: int(getattr(cnt, k)) for k, _ in cnt._fields_}
    return dict(rgba=rgba, ids=ids, t=t, counters=counters, overflow=rc != 0)


def last_max_entry_gap():
    """Largest (box entry - hit distance) / hit distance over the accepted hits of the last render():
    rounding only (a triangle lies inside its leaf-slot box); diagnostic for distance-culling kernels."""
    f = lib().oracle_last_max_entry_gap
    f.restype = C.c_float
    return float(f())


def bruteforce_primary(buffers, frame, width, height):
    sc = _scene(buffers, None)
    ids = np.zeros((height, width, 2), dtype=np.int32)
    t = np.zeros((height, width), dtype=np.float32)
    ties = np.zeros((height, width), dtype=np.int32)
    frame = np.ascontiguousarray(frame)
    lib().oracle_bruteforce_primary(C.byref(sc), _p(frame), width, height, _p(ids), _p(t), _p(ties))
    return ids, t, ties


def check_integrity(nodes, n_prims):
    nodes = np.ascontiguousarray(nodes)
    return lib().oracle_check_integrity(_p(nodes), len(nodes), n_prims)


def struct_sizes():
    out = np.zeros(5, dtype=np.int32)
    lib().oracle_struct_sizes(_p(out))
    return out.tolist()

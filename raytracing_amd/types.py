"""Wire structs of the C ABI as numpy dtypes and ctypes structures.

Byte layouts follow the reference (Assets/Script/Common/RayTraceMaterial.cs:6-35,
Assets/Script/BVH/BoundingVolumeHierarchyBuilder.cs:31-37) and include/rtamd.h.
"""
import ctypes as C

import numpy as np

NODE = np.dtype([("Lx", "<f4", 4), ("Hx", "<f4", 4), ("Ly", "<f4", 4), ("Hy", "<f4", 4),
                 ("Lz", "<f4", 4), ("Hz", "<f4", 4), ("Data", "<i4", 4), ("Flags", "<i4")])
VERTEX = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3)])
INT3 = np.dtype([("x", "<i4"), ("y", "<i4"), ("z", "<i4")])
OBJECT = np.dtype([("BLASIndex", "<i4"), ("albedo", "<f4", 4), ("specular", "<f4", 4), ("emission", "<f4", 4),
                   ("WorldToLocal", "<f4", 16), ("LocalToWorld", "<f4", 16)])
FRAME = np.dtype([("CameraToWorld", "<f4", 16), ("CameraInverseProjection", "<f4", 16),
                  ("PixelOffset", "<f4", 2), ("Seed", "<f4"), ("ReflectionNum", "<i4")])
AABB = np.dtype([("Min", "<f4", 3), ("Max", "<f4", 3)])
POINT_AND_INDEX = np.dtype([("Position", "<f4", 3), ("Index", "<i4")])

assert NODE.itemsize == 116 and VERTEX.itemsize == 24 and INT3.itemsize == 12
assert OBJECT.itemsize == 180 and FRAME.itemsize == 144

MODE_REF, MODE_PT = 0, 1
FLAG_ACCUMULATE, FLAG_NO_JITTER, FLAG_REFERENCE_ORDER, FLAG_OUTPUT_SUM, FLAG_COUNTERS, FLAG_SHARD_SAMPLES, FLAG_DIELECTRICS = 1, 2, 4, 8, 16, 32, 64
FLAG_DEBUG_CHECKS = 128
MATERIAL_DIELECTRIC = 2.0

ERR_INVALID_ARG, ERR_BAD_INDEX, ERR_OUT_OF_MEMORY, ERR_HIP, ERR_STACK_OVERFLOW, ERR_NO_SCENE, ERR_NO_DEVICE = -1, -2, -3, -4, -5, -6, -7
ERRORS = {0: "OK", -1: "INVALID_ARG", -2: "BAD_INDEX", -3: "OUT_OF_MEMORY", -4: "HIP", -5: "STACK_OVERFLOW",
          -6: "NO_SCENE", -7: "NO_DEVICE"}


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes", C.c_uint64), ("tris", C.c_uint64), ("instances", C.c_uint64),
                ("hits", C.c_uint64), ("max_blas_stack", C.c_uint64), ("max_tlas_stack", C.c_uint64),
                ("reserved", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class ExtParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("mode", C.c_int32), ("spp", C.c_int32), ("max_bounces", C.c_int32),
                ("seed", C.c_uint64), ("sample_offset", C.c_int32),
                ("tile_x", C.c_int32), ("tile_y", C.c_int32), ("tile_w", C.c_int32), ("tile_h", C.c_int32),
                ("shard_rank", C.c_int32), ("shard_count", C.c_int32), ("flags", C.c_uint32),
                ("out_t", C.c_void_p), ("out_counters", C.POINTER(Counters))]


class HostBuffers(C.Structure):
    _fields_ = [("tlas", C.c_void_p), ("n_tlas_nodes", C.c_int32),
                ("blas_offsets", C.c_void_p),
                ("blas", C.c_void_p), ("n_blas_nodes", C.c_int32),
                ("vertices_offsets", C.c_void_p),
                ("vertices", C.c_void_p), ("n_vertices", C.c_int32),
                ("triangles_offsets", C.c_void_p),
                ("triangles", C.c_void_p), ("n_triangles", C.c_int32),
                ("n_blas", C.c_int32),
                ("objects", C.c_void_p), ("n_objects", C.c_int32),
                ("tlas_branch_count", C.c_int32)]


class HostMesh(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("n_vertices", C.c_int32),
                ("indices", C.POINTER(C.c_int32)), ("n_triangles", C.c_int32)]

"""GPU backend: ctypes binding of librtamd.so (the C ABI in include/rtamd.h) and a
`RayTracingPass` class shaped like the reference's render feature
(Assets/Script/RenderPass/RayTracingPass.cs:14-147).

There is no CPU fallback: without the HIP library or without a GPU every call raises.
"""
import ctypes as C
import os

import numpy as np

from . import types as T

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librtamd.so")
_lib = None

EXPORTS = ["rtamd_ctx_create", "rtamd_ctx_create_multi", "rtamd_multi_info", "rtamd_multi_exchange_note", "rtamd_ctx_destroy", "rtamd_last_error_string", "rtamd_scene_upload",
           "rtamd_scene_refresh_objects", "rtamd_sky_upload", "rtamd_render", "rtamd_render_noalloc", "rtamd_render_begin", "rtamd_render_wait", "rtamd_accum_download", "rtamd_accum_upload", "rtamd_accum_reset",
           "rtamd_set_stream", "rtamd_set_tlas_build", "rtamd_set_blas_build", "rtamd_last_upload_stats", "rtamd_tlas_info", "rtamd_last_render_stats", "rtamd_last_trace_kernel_ms", "rtamd_last_handover_stats", "rtamd_last_guard_stats", "rtamd_composite_add", "rtamd_present", "rtamd_version"]


class BackendError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rtamd error {code} ({T.ERRORS.get(code, '?')}): {msg}")
        self.code = code


def lib():
    """Load librtamd.so (raises if it has not been built)."""
    global _lib
    if _lib is None:
        path = os.environ.get("RTAMD_LIB", _LIB_PATH)       # (experiments: another build of the same library)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m raytracing_amd.build` (needs hipcc)")
        L = C.CDLL(path)
        L.rtamd_version.restype = C.c_char_p
        L.rtamd_last_error_string.restype = C.c_char_p
        L.rtamd_last_error_string.argtypes = [C.c_void_p]
        L.rtamd_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.rtamd_ctx_destroy.argtypes = [C.c_void_p]
        L.rtamd_ctx_create_multi.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]
        L.rtamd_multi_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        L.rtamd_multi_exchange_note.restype = C.c_char_p
        L.rtamd_multi_exchange_note.argtypes = [C.c_void_p]
        L.rtamd_scene_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_int32, C.c_void_p, C.c_int32]
        L.rtamd_scene_refresh_objects.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.rtamd_sky_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.rtamd_render.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(T.ExtParams), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.rtamd_render_noalloc.argtypes = L.rtamd_render.argtypes
        L.rtamd_render_begin.argtypes = L.rtamd_render.argtypes
        L.rtamd_render_wait.argtypes = [C.c_void_p]
        L.rtamd_accum_download.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.rtamd_accum_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.rtamd_accum_reset.argtypes = [C.c_void_p]
        L.rtamd_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.rtamd_set_tlas_build.argtypes = [C.c_void_p, C.c_int32]
        L.rtamd_set_blas_build.argtypes = [C.c_void_p, C.c_int32]
        L.rtamd_last_upload_stats.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.rtamd_tlas_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.rtamd_last_render_stats.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
        L.rtamd_last_handover_stats.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.rtamd_last_trace_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.rtamd_last_guard_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.rtamd_composite_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        _lib = L
    return _lib


def source_sha():
    """The fingerprint of the sources the loaded librtamd.so was built from (see build.source_sha)."""
    v = lib().rtamd_version().decode()
    return v.split("src:")[1] if "src:" in v else "unstamped"


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    # torch tensor (device or host)
    return C.c_void_p(a.data_ptr())


class RayTraceSettings:
    """RayTracingPass.RayTraceSettings (RayTracingPass.cs:23-32) plus the extension parameters."""

    def __init__(self, ReflectionNum=1, randomSeed=323424, RTDownScaling=1.0, mode=T.MODE_REF, spp=1, max_bounces=8,
                 seed=None, flags=0):
        self.ReflectionNum = ReflectionNum
        self.randomSeed = randomSeed
        self.RTDownScaling = RTDownScaling
        self.mode = mode
        self.spp = spp
        self.max_bounces = max_bounces
        self.seed = randomSeed if seed is None else seed
        self.flags = flags


class RayTracingPass:
    """Owns one GPU context.  `SetScene` = the buffer bindings of Execute (:83-97),
    `Execute` = the dispatch (:101)."""

    def __init__(self, device=0, settings=None, devices=None):
        """device: one GPU ordinal; devices: a list of ordinals for one context over several GPUs
        (rtamd_ctx_create_multi; the frame is assembled on devices[0])."""
        self.settings = settings or RayTraceSettings()
        self._ctx = C.c_void_p()
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*devices)
            rc = lib().rtamd_ctx_create_multi(arr, len(devices), C.byref(self._ctx))
            device = devices[0] if len(devices) else 0
        else:
            rc = lib().rtamd_ctx_create(device, C.byref(self._ctx))
        if rc != 0:
            raise BackendError(rc, lib().rtamd_last_error_string(None).decode())
        self.device = device

    def MultiInfo(self):
        n, ranks, ms = C.c_int32(0), C.c_int32(0), C.c_float(0)
        self._check(lib().rtamd_multi_info(self._ctx, C.byref(n), C.byref(ranks), C.byref(ms)))
        return {"devices": n.value, "rccl_ranks": ranks.value, "exchange_ms": ms.value,
                "note": lib().rtamd_multi_exchange_note(self._ctx).decode()}

    def Dispose(self):
        if getattr(self, "_ctx", None):
            lib().rtamd_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise BackendError(rc, lib().rtamd_last_error_string(self._ctx).decode())

    def SetScene(self, b):
        """Upload the eight reference buffers (a host.SceneBuffers)."""
        self._check(lib().rtamd_scene_upload(
            self._ctx, _ptr(b.tlas), len(b.tlas), _ptr(b.blas_offsets), _ptr(b.blas), len(b.blas),
            _ptr(b.vertices_offsets), _ptr(b.vertices), len(b.vertices), _ptr(b.triangles_offsets),
            _ptr(b.triangles), len(b.triangles), b.n_blas, _ptr(b.objects), len(b.objects)))

    def RefreshObjects(self, b):
        self._check(lib().rtamd_scene_refresh_objects(self._ctx, _ptr(b.tlas), len(b.tlas), _ptr(b.objects), len(b.objects)))

    def SetSkyBox(self, rgba):
        """rgba: (h, w, 4) float32, or None for a black sky."""
        if rgba is None:
            self._check(lib().rtamd_sky_upload(self._ctx, None, 0, 0))
            return
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        h, w, _ = rgba.shape
        self._check(lib().rtamd_sky_upload(self._ctx, _ptr(rgba), w, h))

    def SetStream(self, stream_ptr):
        self._check(lib().rtamd_set_stream(self._ctx, C.c_void_p(stream_ptr) if stream_ptr else None))

    def SetTlasBuild(self, mode):
        """'auto' | 'host' | 'device': who rebuilds the library's private tree over the instances
        (the reference's per-frame TLAS build, BuildAccelerateStructureSystem.cs:56-66)."""
        self._check(lib().rtamd_set_tlas_build(self._ctx, {"auto": 0, "host": 1, "device": 2}[mode]))

    def SetBlasBuild(self, mode):
        """'auto' | 'host' | 'device': who builds the library's private trees over the meshes at the next SetScene
        (the reference bakes its own at edit time, RayTraceableBaker.cs:75-149)."""
        self._check(lib().rtamd_set_blas_build(self._ctx, {"auto": 0, "host": 1, "device": 2}[mode]))

    def UploadStats(self):
        ms, tree, dev = C.c_float(0), C.c_float(0), C.c_int32(0)
        self._check(lib().rtamd_last_upload_stats(self._ctx, C.byref(ms), C.byref(tree), C.byref(dev)))
        return {"upload_ms": ms.value, "tree_build_ms": tree.value, "meshes_built_on_device": dev.value}

    def TlasInfo(self):
        n, lv, dev = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(lib().rtamd_tlas_info(self._ctx, C.byref(n), C.byref(lv), C.byref(dev)))
        return {"nodes": n.value, "levels": lv.value, "built_on_device": bool(dev.value)}

    def GuardStats(self):
        """The culling guard after the last Execute (rtamd_last_guard_stats)."""
        h, m, r = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._check(lib().rtamd_last_guard_stats(self._ctx, C.byref(h), C.byref(m), C.byref(r)))
        return {"hits_near_slack": h.value, "demoted_meshes": m.value, "rerendered": bool(r.value)}

    def DownloadAccumulation(self, width, height):
        """The running sum (h, w, 4): rgb sums and the sample count per pixel."""
        out = np.zeros((height, width, 4), dtype=np.float32)
        self._check(lib().rtamd_accum_download(self._ctx, _ptr(out), width, height))
        return out

    def UploadAccumulation(self, sums):
        sums = np.ascontiguousarray(sums, dtype=np.float32)
        self._check(lib().rtamd_accum_upload(self._ctx, _ptr(sums), sums.shape[1], sums.shape[0]))

    def ResetAccumulation(self):
        self._check(lib().rtamd_accum_reset(self._ctx))

    def ext(self, **kw):
        s = self.settings
        e = T.ExtParams()
        e.struct_size = C.sizeof(T.ExtParams)
        e.mode, e.spp, e.max_bounces, e.seed, e.flags = s.mode, s.spp, s.max_bounces, s.seed, s.flags
        for k, v in kw.items():
            setattr(e, k, v)
        return e

    def ExecuteNoAlloc(self, frame, width, height, out_rgba=None, out_prim_id=None, out_t=None, **ext_kw):
        """Execute through rtamd_render_noalloc: outputs must be device memory (torch tensors or
        raw device pointers); the call allocates nothing on the GPU."""
        return self.Execute(frame, width, height, out_rgba=out_rgba, out_prim_id=out_prim_id, out_t=out_t,
                            _entry="rtamd_render_noalloc", **ext_kw)

    def Execute(self, frame, width, height, out_rgba=None, out_prim_id=None, out_t=None, counters=False,
                _entry="rtamd_render", **ext_kw):
        """Render one frame.

        frame: numpy FRAME record.  Outputs may be numpy arrays, torch tensors (host or
        device) or None; when an output is True a numpy array is allocated and returned.
        Returns dict(rgba=, ids=, t=, counters=, kernel_ms=, rays=, trace_ms=, handover_ms=, handover_pixels=);
        kernel_ms includes handover_ms; trace_ms is the path-tracing kernel alone.
        """
        if out_rgba is True:
            out_rgba = np.zeros((height, width, 4), dtype=np.float32)
        if out_prim_id is True:
            out_prim_id = np.zeros((height, width, 2), dtype=np.int32)
        if out_t is True:
            out_t = np.zeros((height, width), dtype=np.float32)
        e = self.ext(**ext_kw)
        cnt = T.Counters()
        if counters:
            e.flags |= T.FLAG_COUNTERS
            e.out_counters = C.pointer(cnt)
        if out_t is not None:
            e.out_t = _ptr(out_t)
        frame = np.ascontiguousarray(frame, dtype=T.FRAME)
        rc = getattr(lib(), _entry)(self._ctx, _ptr(frame), C.byref(e), width, height, _ptr(out_rgba), _ptr(out_prim_id))
        ms, rays = C.c_float(0), C.c_uint64(0)
        lib().rtamd_last_render_stats(self._ctx, C.byref(ms), C.byref(rays))
        hms, hpix = C.c_float(0), C.c_int32(0)
        lib().rtamd_last_handover_stats(self._ctx, C.byref(hms), C.byref(hpix))
        tms = C.c_float(0)
        lib().rtamd_last_trace_kernel_ms(self._ctx, C.byref(tms))
        self._check(rc)
        return dict(rgba=out_rgba, ids=out_prim_id, t=out_t, counters=cnt.as_dict() if counters else None,
                    kernel_ms=ms.value, rays=int(rays.value), trace_ms=tms.value, handover_ms=hms.value, handover_pixels=int(hpix.value))

    def ExecuteBegin(self, frame, width, height, out_rgba=None, out_prim_id=None, out_t=None, **ext_kw):
        """rtamd_render_begin: queue a frame and return before it is complete (device-memory outputs; at most two
        frames in flight).  ExecuteWait completes the oldest frame begun and returns its statistics."""
        e = self.ext(**ext_kw)
        if out_t is not None:
            e.out_t = _ptr(out_t)
        frame = np.ascontiguousarray(frame, dtype=T.FRAME)
        self._check(lib().rtamd_render_begin(self._ctx, _ptr(frame), C.byref(e), width, height, _ptr(out_rgba), _ptr(out_prim_id)))

    def ExecuteWait(self):
        rc = lib().rtamd_render_wait(self._ctx)
        ms, rays = C.c_float(0), C.c_uint64(0)
        lib().rtamd_last_render_stats(self._ctx, C.byref(ms), C.byref(rays))
        hms, hpix = C.c_float(0), C.c_int32(0)
        lib().rtamd_last_handover_stats(self._ctx, C.byref(hms), C.byref(hpix))
        tms = C.c_float(0)
        lib().rtamd_last_trace_kernel_ms(self._ctx, C.byref(tms))
        self._check(rc)
        return dict(kernel_ms=ms.value, rays=int(rays.value), trace_ms=tms.value, handover_ms=hms.value, handover_pixels=int(hpix.value))

    def ExecuteDefault(self, frame, width, height, out_rgba, out_prim_id=None):
        """The plain reference dispatch: no extension block (REF mode)."""
        frame = np.ascontiguousarray(frame, dtype=T.FRAME)
        self._check(lib().rtamd_render(self._ctx, _ptr(frame), None, width, height, _ptr(out_rgba), _ptr(out_prim_id)))

    def CompositeAdd(self, src_rgba, dst_rgba8):
        h, w = src_rgba.shape[0], src_rgba.shape[1]
        self._check(lib().rtamd_composite_add(self._ctx, _ptr(src_rgba), _ptr(dst_rgba8), w, h))

    def Present(self, src_rgba, dst_rgba8, tonemap=0, exposure=1.0, gamma=1.0):
        """Tone-mapped 8-bit copy of an HDR framebuffer (tonemap 0 = the reference's clamp, 1 = Reinhard)."""
        h, w = src_rgba.shape[0], src_rgba.shape[1]
        L = lib()
        L.rtamd_present.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float]
        self._check(L.rtamd_present(self._ctx, _ptr(src_rgba), _ptr(dst_rgba8), w, h, tonemap, exposure, gamma))

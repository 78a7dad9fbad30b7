// librtamd: host side of the C ABI declared in include/rtamd.h — context, scene upload
// and validation, re-packing for the GPU, kernel launches, timing.
//
// Boundary replaced: CustomRenderPass.Execute (Assets/Script/RenderPass/RayTracingPass.cs:
// 62-109) and the ComputeBuffer plumbing of BuildAccelerateStructureSystem.OnUpdate
// (Assets/Script/RayTracing/System/BuildAccelerateStructureSystem.cs:119-158).

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "kernel_faithful.h"
#include "kernel_fast.h"
#include "kernel_pool.h"
#include "multi_gpu.h"
#include "rtamd.h"

static_assert(sizeof(rtamd_node) == 116, "Node stride");
static_assert(sizeof(rtamd_vertex) == 24, "Vertex stride");
static_assert(sizeof(rtamd_int3) == 12, "int3 stride");
static_assert(sizeof(rtamd_object_property) == 180, "PerObjectProperty stride");
static_assert(sizeof(rtamd_frame_cbuffer) == 144, "FrameCBuffer size");

namespace {

thread_local std::string g_createError;

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        if (count <= n && p) return hipSuccess;
        release();
        hipError_t e = hipMalloc((void**)&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) n = count; else p = nullptr;
        return e;
    }
    hipError_t upload(const T* src, size_t count, hipStream_t s) {
        hipError_t e = alloc(count);
        if (e != hipSuccess || !count) return e;
        return hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s);
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

// ---- working memory of the allocation-free entry point (rtamd_render_noalloc) ----------------
// One fixed array per GPU, held from library load to process exit; every call on a GPU uses that
// GPU's copy.  Layout (byte offsets, each a multiple of 256):
constexpr int64_t kNaMaxPixels = 3840ll * 2160ll;   // largest frame
constexpr int32_t kNaMaxTiles = 8192;               // 32x32 tiles of the largest frame (120 x 68 = 8160)
constexpr int32_t kNaRedoCap = 65536;               // pixels the production kernel may hand to the reference-order kernel
constexpr int32_t kNaSampleCap = 262144;            // samples (float4) of one piece of the hand-over pass; also the largest spp
constexpr size_t kNaOffCounters = 0;                // rt::DevCounters (64 B)
constexpr size_t kNaOffRayCount = 64;               // ray counter, work-queue head (2 x 8 B)
constexpr size_t kNaOffErrorFlag = 80;
constexpr size_t kNaOffRedoCount = 84;
constexpr size_t kCtrlBytes = 256;                  // the block of small words in front (what a render clears and reads back)
constexpr size_t kNaOffPathCount = 88;              // paths handed to the reference-order kernel: this batch, whole render, the batch in the other half of the buffers (3 x 4 B)
constexpr size_t kNaOffGuardCount = 100;            // hits that came within a quarter of the culling slack (RT_GUARD_FACTOR)
constexpr size_t kCtrlUsed = 104;
constexpr size_t kNaOffRedoMask = kCtrlBytes;       // one bit per pixel
constexpr size_t kNaOffRedoList = kNaOffRedoMask + (size_t)(kNaMaxPixels / 32) * 4;
constexpr size_t kNaOffSuperTiles = kNaOffRedoList + (size_t)kNaRedoCap * 4;
constexpr size_t kNaOffSamples = kNaOffSuperTiles + (size_t)kNaMaxTiles * 4;
constexpr size_t kNaOffSpill = kNaOffSamples + (size_t)kNaSampleCap * 16;   // overflow of the LDS traversal stacks: [entry][lane of the grid]
constexpr size_t kNaSpillWords = (size_t)16 * 256 * 1024;                   // 16 entries for each lane of 256 workgroups of 1024 (a larger grid gets fewer entries per lane)
constexpr size_t kNaBytes = kNaOffSpill + kNaSpillWords * 4;                // 22 303 488 bytes
static_assert(kNaOffRedoList % 256 == 0 && kNaOffSuperTiles % 256 == 0 && kNaOffSamples % 256 == 0 && kNaOffSpill % 256 == 0, "scratch pieces are 256-byte aligned");

}  // namespace

static __device__ __align__(256) unsigned char scratch[kNaBytes];

namespace {
// Device address of `scratch` on each GPU (the array's name is only a host-side handle: every GPU
// holds its own copy at its own address, found once per device with hipGetSymbolAddress).
constexpr int kMaxDevices = 64;
std::mutex g_scratchLock;
unsigned char* g_scratchPtr[kMaxDevices] = {};

hipError_t scratchForDevice(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> hold(g_scratchLock);
    if (!g_scratchPtr[device]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));      // the current device's copy
        if (e != hipSuccess) return e;
        g_scratchPtr[device] = (unsigned char*)p;
    }
    *out = g_scratchPtr[device];
    return hipSuccess;
}
}  // namespace

struct rtamd_ctx {
    int device = 0;
    unsigned char* naScratch = nullptr;   // this GPU's copy of `scratch`
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    std::string error;

    // reference-layout copies (used by the reference-order kernel)
    DevBuf<rtamd_node> tlas, blas;
    DevBuf<int32_t> blasOffsets, vtxOffsets, triOffsets;
    DevBuf<rtamd_vertex> verts;
    DevBuf<rtamd_int3> tris;
    DevBuf<rtamd_object_property> objects;
    int nTlas = 0, nBlasNodes = 0, nVerts = 0, nTris = 0, nBlas = 0, nObjects = 0;
    bool haveScene = false;

    // host copies of what the re-pack needs again on an object refresh
    std::vector<int32_t> hBlasOffsets, hTriOffsets, hVtxOffsets;
    std::vector<int32_t> hBlasNodeCount, hBlasTriCount, hBlasVertCount;

    // GPU layout (production kernel)
    rt::FastSceneHost fast;
    rt::PoolHost pool;                    // path records and sample buffer of kPool

    DevBuf<float4> sky;
    int skyW = 0, skyH = 0;

    DevBuf<float4> accum;
    int accumW = 0, accumH = 0;

    DevBuf<uchar4> tmp8;         // 8-bit staging of rtamd_composite_add / rtamd_present for host buffers
    DevBuf<float4> outRgba;      // staging when the caller passes host pointers
    DevBuf<int32_t> outIds;
    DevBuf<float> outT;
    // The small words every render clears and reads back — counters, ray count + work-queue head, error flag,
    // hand-over count — sit in one 256-byte block (laid out like the head of `scratch`), so a render clears
    // them with one fill and reads them with one copy into pinned host memory.
    DevBuf<unsigned char> ctrl;
    unsigned char* hostCtrl = nullptr;    // pinned, kCtrlBytes

    DevBuf<uint32_t> redoMask;            // pixels handed from the production kernel to the reference-order kernel
    DevBuf<int32_t> redoList;
    DevBuf<float4> redoSamples;
    int32_t lastRedoPixels = 0;
    int32_t lastGuardHits = 0, lastGuardRerendered = 0, demotedMeshes = 0;      // the culling guard: see rtamd_last_guard_stats
    float lastKernelMs = 0.f, lastHandoverMs = 0.f, lastTraceMs = 0.f;
    uint64_t lastRays = 0;
    float lastUploadMs = 0.f;      // wall-clock time of the last rtamd_scene_upload (validation, re-pack, tree builds, copies)

    // Frames that return before they are complete (rtamd_render_begin / rtamd_render_wait): up to two in flight,
    // each with a working set of its own (the control block, the hand-over mask and list) and what
    // rtamd_render_wait needs to finish the frame.
    struct FrameSlot {
        DevBuf<unsigned char> ctrl;
        unsigned char* hostCtrl = nullptr;       // pinned
        DevBuf<uint32_t> redoMask;
        DevBuf<int32_t> redoList;
        hipEvent_t evStart = nullptr, evStop = nullptr, evDone = nullptr;
        rt::RenderArgs a;
        rtamd_frame_cbuffer frame;
        rtamd_ext_params e;
        int32_t width = 0, height = 0;
        float* outRgba = nullptr;
        int32_t* outIds = nullptr;
        bool usedPool = false;
    };
    FrameSlot slots[2];
    int asyncBegun = 0, asyncInFlight = 0;        // frames begun so far / not yet waited for (0 .. 2)

    // A context over several GPUs (rtamd_ctx_create_multi) owns one ordinary context per device and
    // nothing else on the GPUs; subs[0] is the device the finished frame is assembled on.
    std::vector<rtamd_ctx*> subs;
    std::vector<std::unique_ptr<rt::DeviceWorker>> workers;   // host thread of subs[d], d >= 1 (subs[0] runs on the caller's thread)
    std::vector<ncclComm_t> comms;                    // per sub, when RCCL carries the exchange
    int rcclRanks = 0;
    std::string exchangeNote;                         // why RCCL is not in use, if it is not
    std::vector<DevBuf<unsigned char>> piece, recv;   // per sub: its packed tiles (on its device) / where they land (on subs[0]'s device)
    float lastExchangeMs = 0.f;
};

namespace {

int fail(rtamd_ctx* c, int code, const std::string& msg) {
    if (c) c->error = msg; else g_createError = msg;
    return code;
}
// Scene, sky and running-sum changes and blocking renders are for a context with no frame in flight.
#define NO_FRAME_IN_FLIGHT(c, what)                                                                                                  \
    do {                                                                                                                             \
        if ((c)->asyncInFlight > 0)                                                                                                  \
            return fail(c, RTAMD_ERR_INVALID_ARG, std::string(what) + ": frames begun with rtamd_render_begin are in flight, call rtamd_render_wait first"); \
    } while (0)
int failHip(rtamd_ctx* c, hipError_t e, const char* what) {
    return fail(c, e == hipErrorOutOfMemory ? RTAMD_ERR_OUT_OF_MEMORY : RTAMD_ERR_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(c, expr)                                   \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return failHip(c, _e, #expr); \
    } while (0)

// No C++ exception crosses the C boundary: host-side allocation failures (vectors, strings, builder
// threads) come back as status codes like everything else.
template <typename F>
int guarded(rtamd_ctx* c, F body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        try { return fail(c, RTAMD_ERR_OUT_OF_MEMORY, "host allocation failed"); } catch (...) { return RTAMD_ERR_OUT_OF_MEMORY; }
    } catch (const std::exception& e) {
        try { return fail(c, RTAMD_ERR_HIP, std::string("internal error: ") + e.what()); } catch (...) { return RTAMD_ERR_HIP; }
    } catch (...) {
        try { return fail(c, RTAMD_ERR_HIP, "internal error"); } catch (...) { return RTAMD_ERR_HIP; }
    }
}

bool isDevicePointer(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// Lanes left idle between the items of a hand-over list (log2): as many as keep the launch within
// the number of waves the GPU holds at once, so a short list runs one path per wave.
int listSpreadFor(size_t items) {
    const size_t residentLanes = (size_t)256 * 32 * 64;      // 256 CUs x 8 waves per SIMD x 4 SIMDs, 64 lanes each
    int spread = 0;
    if (const char* s = getenv("RTAMD_LIST_SPREAD")) return atoi(s) < 0 ? 0 : (atoi(s) > 6 ? 6 : atoi(s));
    while (spread < 6 && (items << (spread + 1)) <= residentLanes) spread++;
    return spread;
}

// Walk the nodes reachable from node 1 of one tree and check every index the kernels
// will follow.  leafLimit = number of primitives (objects for the TLAS).
int validateTree(rtamd_ctx* c, const rtamd_node* nodes, int nNodes, int leafLimit, const char* what) {
    if (nNodes < 2) return fail(c, RTAMD_ERR_BAD_INDEX, std::string(what) + ": fewer than 2 nodes");
    std::vector<uint8_t> seen((size_t)nNodes, 0);
    std::vector<int> st;
    st.push_back(1);
    seen[1] = 1;
    while (!st.empty()) {
        int ni = st.back();
        st.pop_back();
        const rtamd_node& n = nodes[ni];
        for (int i = 0; i < 4; i++) {
            int d = n.Data[i];
            if (n.Flags == 0) {
                if (d == 0) continue;
                if (d < 0 || d >= nNodes) return fail(c, RTAMD_ERR_BAD_INDEX, std::string(what) + ": child index out of range");
                if (seen[d]) return fail(c, RTAMD_ERR_BAD_INDEX, std::string(what) + ": node reachable twice");
                seen[d] = 1;
                st.push_back(d);
            } else {
                if (d == -1) continue;
                if (d < 0 || d >= leafLimit) return fail(c, RTAMD_ERR_BAD_INDEX, std::string(what) + ": leaf index out of range");
            }
        }
    }
    return RTAMD_OK;
}

}  // namespace

// A context over several GPUs forwards every call to its per-device contexts (defined further down).
namespace multi {
template <typename F> int forEach(rtamd_ctx* c, F call);
int render(rtamd_ctx* c, const rtamd_frame_cbuffer* frame, const rtamd_ext_params* ext, int32_t width, int32_t height,
           float* out_rgba, int32_t* out_prim_id);
}  // namespace multi

extern "C" {

#ifndef RTAMD_SOURCE_SHA
#define RTAMD_SOURCE_SHA "unstamped"
#endif
// (the fingerprint of the sources this library was built from: raytracing_amd/build.py, source_sha)
const char* rtamd_version(void) { return "rtamd 0.3 gfx950 src:" RTAMD_SOURCE_SHA; }

const char* rtamd_last_error_string(const rtamd_ctx* ctx) {
    return ctx ? ctx->error.c_str() : g_createError.c_str();
}

int rtamd_ctx_create(int device_ordinal, rtamd_ctx** out_ctx) {
    return guarded(nullptr, [&]() -> int {
    if (!out_ctx) return fail(nullptr, RTAMD_ERR_INVALID_ARG, "out_ctx is NULL");
    *out_ctx = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, RTAMD_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    }
    if (device_ordinal < 0 || device_ordinal >= count) return fail(nullptr, RTAMD_ERR_INVALID_ARG, "device ordinal out of range");
    rtamd_ctx* c = new (std::nothrow) rtamd_ctx();
    if (!c) return fail(nullptr, RTAMD_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->device = device_ordinal;
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess || (e = scratchForDevice(device_ordinal, &c->naScratch)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->evStart)) != hipSuccess || (e = hipEventCreate(&c->evStop)) != hipSuccess ||
        (e = c->ctrl.alloc(kCtrlBytes)) != hipSuccess || (e = hipHostMalloc((void**)&c->hostCtrl, kCtrlBytes, hipHostMallocDefault)) != hipSuccess) {
        int rc = failHip(nullptr, e, "context creation");
        rtamd_ctx_destroy(c);
        return rc;
    }
    c->stream = c->ownStream;
    if ((e = hipDeviceGetAttribute(&c->fast.numCU, hipDeviceAttributeMultiprocessorCount, device_ordinal)) != hipSuccess || c->fast.numCU <= 0) {
        int rc = failHip(nullptr, e == hipSuccess ? hipErrorInvalidDevice : e, "compute-unit count of the device");
        rtamd_ctx_destroy(c);
        return rc;
    }
    c->fast.poolPathCap = rt::poolMaxPathsForCurrentDevice();
    c->fast.hSuper.reserve((size_t)kNaMaxTiles);     // the tile list of a launch never grows the heap after this
    *out_ctx = c;
    return RTAMD_OK;
    });
}

int rtamd_ctx_destroy(rtamd_ctx* c) {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (!c->subs.empty()) {
        c->workers.clear();        // (joins the per-device host threads)
        for (ncclComm_t comm : c->comms) if (comm) (void)rt::rccl().CommDestroy(comm);
        for (size_t d = 0; d < c->subs.size(); d++) {
            (void)hipSetDevice(c->subs[d]->device);
            if (d < c->piece.size()) c->piece[d].release();
        }
        (void)hipSetDevice(c->subs[0]->device);
        for (auto& b : c->recv) b.release();
        for (rtamd_ctx* sub : c->subs) (void)rtamd_ctx_destroy(sub);
        delete c;
        return RTAMD_OK;
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->tlas.release(); c->blas.release(); c->blasOffsets.release(); c->vtxOffsets.release(); c->triOffsets.release();
    c->verts.release(); c->tris.release(); c->objects.release(); c->sky.release(); c->accum.release();
    c->tmp8.release(); c->outRgba.release(); c->outIds.release(); c->outT.release(); c->ctrl.release();
    c->redoMask.release(); c->redoList.release(); c->redoSamples.release();
    if (c->pool.side) (void)hipStreamSynchronize(c->pool.side);
    for (rtamd_ctx::FrameSlot& S : c->slots) {
        S.ctrl.release(); S.redoMask.release(); S.redoList.release();
        if (S.hostCtrl) (void)hipHostFree(S.hostCtrl);
        if (S.evStart) (void)hipEventDestroy(S.evStart);
        if (S.evStop) (void)hipEventDestroy(S.evStop);
        if (S.evDone) (void)hipEventDestroy(S.evDone);
    }
    if (c->hostCtrl) (void)hipHostFree(c->hostCtrl);
    c->fast.release();
    c->pool.release();
    if (c->evStart) (void)hipEventDestroy(c->evStart);
    if (c->evStop) (void)hipEventDestroy(c->evStop);
    if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    delete c;
    return RTAMD_OK;
}

int rtamd_set_stream(rtamd_ctx* c, void* s) {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    NO_FRAME_IN_FLIGHT(c, "rtamd_set_stream");
    if (!c->subs.empty()) return fail(c, RTAMD_ERR_INVALID_ARG, "a context over several GPUs runs every device on a stream of its own");
    c->stream = s ? (hipStream_t)s : c->ownStream;
    return RTAMD_OK;
}

int rtamd_set_tlas_build(rtamd_ctx* c, int32_t mode) {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (mode < RTAMD_TLAS_BUILD_AUTO || mode > RTAMD_TLAS_BUILD_DEVICE) return fail(c, RTAMD_ERR_INVALID_ARG, "unknown TLAS build mode");
    if (!c->subs.empty()) return multi::forEach(c, [&](rtamd_ctx* sub) { return rtamd_set_tlas_build(sub, mode); });
    c->fast.tlasBuildMode = mode;
    return RTAMD_OK;
}

int rtamd_set_blas_build(rtamd_ctx* c, int32_t mode) {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (mode < RTAMD_BLAS_BUILD_AUTO || mode > RTAMD_BLAS_BUILD_DEVICE) return fail(c, RTAMD_ERR_INVALID_ARG, "unknown BLAS build mode");
    if (!c->subs.empty()) return multi::forEach(c, [&](rtamd_ctx* sub) { return rtamd_set_blas_build(sub, mode); });
    c->fast.blasBuildMode = mode;
    return RTAMD_OK;
}

int rtamd_last_upload_stats(const rtamd_ctx* c, float* upload_ms, float* tree_build_ms, int32_t* meshes_built_on_device) {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (!c->subs.empty()) return rtamd_last_upload_stats(c->subs[0], upload_ms, tree_build_ms, meshes_built_on_device);
    if (upload_ms) *upload_ms = c->lastUploadMs;
    if (tree_build_ms) *tree_build_ms = c->fast.blasBuildMs;
    if (meshes_built_on_device) *meshes_built_on_device = c->fast.blasBuiltOnDevice;
    return RTAMD_OK;
}

int rtamd_tlas_info(const rtamd_ctx* c, int32_t* nodes, int32_t* levels, int32_t* built_on_device) {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (!c->subs.empty()) return rtamd_tlas_info(c->subs[0], nodes, levels, built_on_device);
    if (!c->haveScene) return RTAMD_ERR_NO_SCENE;
    if (nodes) *nodes = c->fast.nTlasNodes;
    if (levels) *levels = c->fast.tlasDepth;
    if (built_on_device) *built_on_device = c->fast.tlasBuiltOnDevice ? 1 : 0;
    return RTAMD_OK;
}

static int uploadTlasAndObjects(rtamd_ctx* c, const rtamd_node* tlas, int nTlas, const rtamd_object_property* objects,
                                int nObjects) {
    int rc = validateTree(c, tlas, nTlas, nObjects, "TLAS");
    if (rc) return rc;
    for (int o = 0; o < nObjects; o++)
        if (objects[o].BLASIndex < 0 || objects[o].BLASIndex >= c->nBlas)
            return fail(c, RTAMD_ERR_BAD_INDEX, "object BLASIndex out of range");
    c->haveScene = false;      // from here on device state changes; a call rejected above leaves the last good scene in place
    HIP_TRY(c, c->tlas.upload(tlas, nTlas, c->stream));
    HIP_TRY(c, c->objects.upload(objects, nObjects, c->stream));
    c->nTlas = nTlas;
    c->nObjects = nObjects;
    std::string err;
    hipError_t e = c->fast.buildTop(tlas, nTlas, objects, nObjects, c->tlas.p, c->objects.p, c->stream, err);
    if (e != hipSuccess) return failHip(c, e, "TLAS re-pack");
    if (!err.empty()) return fail(c, RTAMD_ERR_BAD_INDEX, err);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTAMD_OK;
}

int rtamd_scene_upload(rtamd_ctx* c, const rtamd_node* tlas, int32_t nTlas, const int32_t* blasOffsets,
                       const rtamd_node* blas, int32_t nBlasNodes, const int32_t* vtxOffsets,
                       const rtamd_vertex* verts, int32_t nVerts, const int32_t* triOffsets, const rtamd_int3* tris,
                       int32_t nTris, int32_t nBlas, const rtamd_object_property* objects, int32_t nObjects) {
    return guarded(c, [&]() -> int {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (!c->subs.empty()) {
        // the scene is replicated: every device re-packs and uploads it, all at once
        const int rc = multi::forEach(c, [&](rtamd_ctx* sub) {
            return rtamd_scene_upload(sub, tlas, nTlas, blasOffsets, blas, nBlasNodes, vtxOffsets, verts, nVerts, triOffsets, tris, nTris, nBlas, objects, nObjects);
        });
        c->haveScene = rc == RTAMD_OK;
        return rc;
    }
    NO_FRAME_IN_FLIGHT(c, "rtamd_scene_upload");
    if (!tlas || !blasOffsets || !blas || !vtxOffsets || !verts || !triOffsets || !tris || !objects)
        return fail(c, RTAMD_ERR_INVALID_ARG, "NULL scene buffer");
    if (nTlas < 2 || nBlasNodes < 2 || nVerts < 1 || nTris < 1 || nBlas < 1 || nObjects < 1)
        return fail(c, RTAMD_ERR_INVALID_ARG, "empty scene buffer (the reference skips the dispatch when ObjectsCount == 0)");
    HIP_TRY(c, hipSetDevice(c->device));
    c->haveScene = false;
    const auto uploadBegan = std::chrono::steady_clock::now();
    // offsets: exclusive prefix sums, BuildAccelerateStructureSystem.cs:105-117
    std::vector<int32_t> nodeCnt(nBlas), triCnt(nBlas), vertCnt(nBlas);
    for (int b = 0; b < nBlas; b++) {
        int no = blasOffsets[b], to = triOffsets[b], vo = vtxOffsets[b];
        int ne = b + 1 < nBlas ? blasOffsets[b + 1] : nBlasNodes;
        int te = b + 1 < nBlas ? triOffsets[b + 1] : nTris;
        int ve = b + 1 < nBlas ? vtxOffsets[b + 1] : nVerts;
        if (no < 0 || ne > nBlasNodes || ne - no < 2 || to < 0 || te > nTris || te <= to || vo < 0 || ve > nVerts || ve <= vo)
            return fail(c, RTAMD_ERR_BAD_INDEX, "BLAS offsets are not increasing prefix sums inside the buffers");
        nodeCnt[b] = ne - no; triCnt[b] = te - to; vertCnt[b] = ve - vo;
        int rc = validateTree(c, blas + no, ne - no, te - to, "BLAS");
        if (rc) return rc;
        for (int t = to; t < te; t++) {
            const rtamd_int3& tr = tris[t];
            if (tr.x < 0 || tr.y < 0 || tr.z < 0 || tr.x >= ve - vo || tr.y >= ve - vo || tr.z >= ve - vo)
                return fail(c, RTAMD_ERR_BAD_INDEX, "triangle vertex index out of range");
        }
    }
    const float validateMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - uploadBegan).count();
    c->nBlas = nBlas;
    c->hBlasOffsets.assign(blasOffsets, blasOffsets + nBlas);
    c->hTriOffsets.assign(triOffsets, triOffsets + nBlas);
    c->hVtxOffsets.assign(vtxOffsets, vtxOffsets + nBlas);
    c->hBlasNodeCount = nodeCnt; c->hBlasTriCount = triCnt; c->hBlasVertCount = vertCnt;

    HIP_TRY(c, c->blasOffsets.upload(blasOffsets, nBlas, c->stream));
    HIP_TRY(c, c->vtxOffsets.upload(vtxOffsets, nBlas, c->stream));
    HIP_TRY(c, c->triOffsets.upload(triOffsets, nBlas, c->stream));
    HIP_TRY(c, c->blas.upload(blas, nBlasNodes, c->stream));
    HIP_TRY(c, c->verts.upload(verts, nVerts, c->stream));
    HIP_TRY(c, c->tris.upload(tris, nTris, c->stream));
    c->nBlasNodes = nBlasNodes; c->nVerts = nVerts; c->nTris = nTris;

    c->demotedMeshes = 0;
    std::string err;
    const float rawCopyMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - uploadBegan).count() - validateMs;
    hipError_t e = c->fast.buildBottom(blasOffsets, blas, nodeCnt.data(), vtxOffsets, verts, triOffsets, tris,
                                       triCnt.data(), nBlas, nObjects, c->stream, err);
    if (e != hipSuccess) return failHip(c, e, "BLAS re-pack");
    if (!err.empty()) return fail(c, RTAMD_ERR_BAD_INDEX, err);

    int rc = uploadTlasAndObjects(c, tlas, nTlas, objects, nObjects);
    if (rc) return rc;
    c->haveScene = true;
    c->lastUploadMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - uploadBegan).count();
    if (getenv("RTAMD_VERBOSE"))
        fprintf(stderr, "[rtamd] upload: %.1f ms in all; validation %.1f ms, copies of the caller's arrays %.1f ms\n", c->lastUploadMs, validateMs, rawCopyMs);
    return RTAMD_OK;
    });
}

int rtamd_scene_refresh_objects(rtamd_ctx* c, const rtamd_node* tlas, int32_t nTlas,
                                const rtamd_object_property* objects, int32_t nObjects) {
    return guarded(c, [&]() -> int {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (!c->subs.empty()) return multi::forEach(c, [&](rtamd_ctx* sub) { return rtamd_scene_refresh_objects(sub, tlas, nTlas, objects, nObjects); });
    NO_FRAME_IN_FLIGHT(c, "rtamd_scene_refresh_objects");
    if (!c->haveScene) return fail(c, RTAMD_ERR_NO_SCENE, "refresh before a scene upload");
    if (!tlas || !objects || nTlas < 2) return fail(c, RTAMD_ERR_INVALID_ARG, "NULL / empty buffer");
    if (nObjects != c->nObjects) return fail(c, RTAMD_ERR_INVALID_ARG, "object count differs from the uploaded scene");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = uploadTlasAndObjects(c, tlas, nTlas, objects, nObjects);
    if (rc) return rc;
    c->haveScene = true;
    return RTAMD_OK;
    });
}

int rtamd_sky_upload(rtamd_ctx* c, const float* rgba, int32_t w, int32_t h) {
    return guarded(c, [&]() -> int {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (!c->subs.empty()) return multi::forEach(c, [&](rtamd_ctx* sub) { return rtamd_sky_upload(sub, rgba, w, h); });
    NO_FRAME_IN_FLIGHT(c, "rtamd_sky_upload");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!rgba || w <= 0 || h <= 0) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->sky.release(); c->skyW = c->skyH = 0;
        return RTAMD_OK;
    }
    HIP_TRY(c, c->sky.upload((const float4*)rgba, (size_t)w * h, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->skyW = w; c->skyH = h;
    return RTAMD_OK;
    });
}

int rtamd_accum_reset(rtamd_ctx* c) {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (!c->subs.empty()) return multi::forEach(c, [&](rtamd_ctx* sub) { return rtamd_accum_reset(sub); });
    NO_FRAME_IN_FLIGHT(c, "rtamd_accum_reset");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->accum.p) HIP_TRY(c, hipMemsetAsync(c->accum.p, 0, c->accum.n * sizeof(float4), c->stream));
    return RTAMD_OK;
}

}  // extern "C"

// One render.  noAlloc: working memory comes from this GPU's `scratch` array instead of the
// context's growable buffers, outputs must be device memory, and nothing is allocated or freed.
static int renderCore(rtamd_ctx* c, const rtamd_frame_cbuffer* frame, const rtamd_ext_params* ext, int32_t width,
                      int32_t height, float* out_rgba, int32_t* out_prim_id, const bool noAlloc, const bool afterGuard = false) {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (!frame) return fail(c, RTAMD_ERR_INVALID_ARG, "frame is NULL");
    if (width <= 0 || height <= 0 || width > (1 << 20) || height > (1 << 20) || (int64_t)width * height > (1ll << 30)) return fail(c, RTAMD_ERR_INVALID_ARG, "bad image size");
    if (!c->haveScene) return fail(c, RTAMD_ERR_NO_SCENE, "render before a scene upload");
    if (!afterGuard) NO_FRAME_IN_FLIGHT(c, "rtamd_render");
    rtamd_ext_params e;
    std::memset(&e, 0, sizeof(e));
    e.struct_size = sizeof(e); e.mode = RTAMD_MODE_REF; e.spp = 1;
    if (ext) {
        if (ext->struct_size != sizeof(rtamd_ext_params)) return fail(c, RTAMD_ERR_INVALID_ARG, "ext.struct_size mismatch");
        e = *ext;
    }
    if (e.mode != RTAMD_MODE_REF && e.mode != RTAMD_MODE_PT) return fail(c, RTAMD_ERR_INVALID_ARG, "unknown mode");
    if (e.mode == RTAMD_MODE_PT && (e.spp < 1 || e.max_bounces < 0 || e.sample_offset < 0))
        return fail(c, RTAMD_ERR_INVALID_ARG, "PT mode needs spp >= 1, max_bounces >= 0, sample_offset >= 0");
    if (e.shard_count > 1 && (e.shard_rank < 0 || e.shard_rank >= e.shard_count)) return fail(c, RTAMD_ERR_INVALID_ARG, "shard_rank out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t npix = (size_t)width * height;
    if (noAlloc) {
        const int64_t tiles = (int64_t)((width + 31) / 32) * (int64_t)((height + 31) / 32);
        if ((int64_t)npix > kNaMaxPixels || tiles > kNaMaxTiles)
            return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_render_noalloc: frame larger than 3840x2160 pixels (or more than 8192 tiles of 32x32)");
        if (e.mode == RTAMD_MODE_PT && e.spp > kNaSampleCap) return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_render_noalloc: spp above 262144");
        if ((out_rgba && !isDevicePointer(out_rgba)) || (out_prim_id && !isDevicePointer(out_prim_id)) || (e.out_t && !isDevicePointer(e.out_t)))
            return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_render_noalloc: outputs must be device memory (a host buffer needs a staging allocation)");
        if ((e.flags & RTAMD_FLAG_ACCUMULATE) && e.mode == RTAMD_MODE_PT && (c->accumW != width || c->accumH != height || !c->accum.p))
            return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_render_noalloc: no running sum of this size yet (the first accumulating frame must go through rtamd_render)");
    }
    // working memory of this call
    unsigned char* const na = c->naScratch;
    unsigned char* const ctrl = noAlloc ? na : c->ctrl.p;
    rt::DevCounters* const dCounters = (rt::DevCounters*)(ctrl + kNaOffCounters);
    unsigned long long* const dRayCount = (unsigned long long*)(ctrl + kNaOffRayCount);
    int32_t* const dErrorFlag = (int32_t*)(ctrl + kNaOffErrorFlag);
    int32_t* const dRedoCount = (int32_t*)(ctrl + kNaOffRedoCount);
    const rt::LaunchScratch lent = {(int32_t*)(na + kNaOffSuperTiles), kNaMaxTiles, (int32_t*)(na + kNaOffSpill), kNaSpillWords};

    rt::RenderArgs a;
    std::memset(&a, 0, sizeof(a));
    a.cb = *frame;
    a.width = width; a.height = height;
    a.mode = e.mode; a.spp = e.mode == RTAMD_MODE_PT ? e.spp : 1; a.maxBounces = e.max_bounces;
    a.seedHi = (uint32_t)(e.seed >> 32); a.seedLo = (uint32_t)e.seed;
    a.sampleOffset = e.sample_offset;
    a.tileX = 0; a.tileY = 0; a.tileW = width; a.tileH = height;
    bool partial = e.shard_count > 1;
    if (e.tile_w > 0 && e.tile_h > 0) {
        int x0 = e.tile_x < 0 ? 0 : e.tile_x, y0 = e.tile_y < 0 ? 0 : e.tile_y;
        int x1 = e.tile_x + e.tile_w > width ? width : e.tile_x + e.tile_w;
        int y1 = e.tile_y + e.tile_h > height ? height : e.tile_y + e.tile_h;
        a.tileX = x0; a.tileY = y0; a.tileW = x1 > x0 ? x1 - x0 : 0; a.tileH = y1 > y0 ? y1 - y0 : 0;
        partial = true;
    }
    a.shardRank = e.shard_rank; a.shardCount = e.shard_count;
    a.flags = e.flags;
    if ((e.flags & RTAMD_FLAG_COUNTERS) && !e.out_counters) return fail(c, RTAMD_ERR_INVALID_ARG, "RTAMD_FLAG_COUNTERS needs ext.out_counters");
    if ((e.flags & RTAMD_FLAG_DEBUG_CHECKS) && (e.flags & (RTAMD_FLAG_DIELECTRICS | RTAMD_FLAG_REFERENCE_ORDER | RTAMD_FLAG_COUNTERS)))
        return fail(c, RTAMD_ERR_INVALID_ARG, "RTAMD_FLAG_DEBUG_CHECKS is a variant of the production kernel without glass: not together with DIELECTRICS, REFERENCE_ORDER or COUNTERS");
    bool wantCounters = (e.flags & RTAMD_FLAG_COUNTERS) != 0;
    bool faithful = (e.flags & RTAMD_FLAG_REFERENCE_ORDER) || wantCounters;

    // outputs: device pointers are written in place, host pointers go through staging
    bool rgbaDev = isDevicePointer(out_rgba), idsDev = isDevicePointer(out_prim_id), tDev = isDevicePointer(e.out_t);
    if (out_rgba && !rgbaDev) {
        HIP_TRY(c, c->outRgba.alloc(npix));
        if (partial) HIP_TRY(c, hipMemcpyAsync(c->outRgba.p, out_rgba, npix * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    }
    if (out_prim_id && !idsDev) {
        HIP_TRY(c, c->outIds.alloc(npix * 2));
        if (partial) HIP_TRY(c, hipMemcpyAsync(c->outIds.p, out_prim_id, npix * 2 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    }
    if (e.out_t && !tDev) {
        HIP_TRY(c, c->outT.alloc(npix));
        if (partial) HIP_TRY(c, hipMemcpyAsync(c->outT.p, e.out_t, npix * sizeof(float), hipMemcpyHostToDevice, c->stream));
    }
    a.outRgba = out_rgba ? (rgbaDev ? (float4*)out_rgba : c->outRgba.p) : nullptr;
    a.outIds = out_prim_id ? (idsDev ? out_prim_id : c->outIds.p) : nullptr;
    a.outT = e.out_t ? (tDev ? e.out_t : c->outT.p) : nullptr;
    if (e.mode == RTAMD_MODE_PT && (e.flags & RTAMD_FLAG_ACCUMULATE)) {
        if (c->accumW != width || c->accumH != height || !c->accum.p) {
            HIP_TRY(c, c->accum.alloc(npix));
            HIP_TRY(c, hipMemsetAsync(c->accum.p, 0, npix * sizeof(float4), c->stream));
            c->accumW = width; c->accumH = height;
        }
        a.accum = c->accum.p;
    }
    a.counters = wantCounters ? dCounters : nullptr;
    a.rayCount = dRayCount;
    a.errorFlag = dErrorFlag;

    rt::Sky sky;
    sky.texels = c->sky.p; sky.w = c->skyW; sky.h = c->skyH;
    sky.invW = c->skyW > 0 ? 1.0f / (float)c->skyW : 0.f; sky.invH = c->skyH > 0 ? 1.0f / (float)c->skyH : 0.f;

    int32_t redo = 0;
    float ms = 0.f, handoverMs = 0.f;
    bool usedPool = false;
    const size_t maskWords = (npix + 31) / 32;
    if (noAlloc) {
        // the array keeps the last call's contents: everything that must start at zero is cleared here
        a.redoMask = (uint32_t*)(na + kNaOffRedoMask); a.redoList = (int32_t*)(na + kNaOffRedoList);
        a.redoCap = (int32_t)npix < kNaRedoCap ? (int32_t)npix : kNaRedoCap;
    } else {
        HIP_TRY(c, c->redoMask.alloc(maskWords));
        HIP_TRY(c, c->redoList.alloc(npix));
        a.redoMask = c->redoMask.p; a.redoList = c->redoList.p; a.redoCap = (int32_t)npix;
    }
    a.redoCount = dRedoCount;
    if (noAlloc) HIP_TRY(c, hipMemsetAsync(ctrl, 0, kCtrlBytes + maskWords * sizeof(uint32_t), c->stream));      // (the mask follows the block)
    else {
        HIP_TRY(c, hipMemsetAsync(a.redoMask, 0, maskWords * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMemsetAsync(ctrl, 0, kCtrlBytes, c->stream));
    }
    rt::RefScene rsc;
    rsc.tlas = c->tlas.p; rsc.blasOffsets = c->blasOffsets.p; rsc.blas = c->blas.p; rsc.vtxOffsets = c->vtxOffsets.p;
    rsc.verts = c->verts.p; rsc.triOffsets = c->triOffsets.p; rsc.tris = c->tris.p; rsc.objects = c->objects.p;
    rsc.guardCount = faithful ? nullptr : (int32_t*)(ctrl + kNaOffGuardCount);
    rsc.guardBlas = faithful ? nullptr : c->fast.guardBlas.p;
    HIP_TRY(c, hipEventRecord(c->evStart, c->stream));
    if (a.tileW > 0 && a.tileH > 0) {
        if (faithful) {
            dim3 grid((width + 7) / 8, (height + 7) / 8);
            hipLaunchKernelGGL(rt::kFaithful, grid, dim3(64), 0, c->stream, rsc, sky, a, (const int32_t*)nullptr, 0, (float4*)nullptr, 0);
        } else {
            // (the path pool needs record and sample buffers sized to the frame: without allocations PT mode
            // runs one path per lane, which gives the same image bit for bit)
            usedPool = a.mode == RTAMD_MODE_PT && rt::fastLaunchConfig().pool && !noAlloc && !(a.flags & RTAMD_FLAG_DEBUG_CHECKS);
            hipError_t le = usedPool ? rt::launchPool(c->fast, c->pool, sky, a, c->stream, rsc, (int32_t*)(ctrl + kNaOffPathCount), (int32_t*)(ctrl + kNaOffPathCount + 8), (int32_t*)(ctrl + kNaOffPathCount + 4))
                                     : rt::launchFast(c->fast, sky, a, c->stream, noAlloc ? &lent : nullptr);
            if (le != hipSuccess) return failHip(c, le, "production kernel launch");
        }
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipEventRecord(c->evStop, c->stream));
    // one read-back of the block; with nothing handed over it is final, and the caller's host buffers (if any)
    // can be filled behind it before the one synchronisation of the call
    const bool hostOutputs = (out_rgba && !rgbaDev) || (out_prim_id && !idsDev) || (e.out_t && !tDev);
    auto copyOutputs = [&]() -> hipError_t {
        hipError_t ce = hipSuccess;
        if (out_rgba && !rgbaDev) ce = hipMemcpyAsync(out_rgba, c->outRgba.p, npix * sizeof(float4), hipMemcpyDeviceToHost, c->stream);
        if (ce == hipSuccess && out_prim_id && !idsDev) ce = hipMemcpyAsync(out_prim_id, c->outIds.p, npix * 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
        if (ce == hipSuccess && e.out_t && !tDev) ce = hipMemcpyAsync(e.out_t, c->outT.p, npix * sizeof(float), hipMemcpyDeviceToHost, c->stream);
        return ce;
    };
    HIP_TRY(c, hipMemcpyAsync(c->hostCtrl, ctrl, kCtrlUsed, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::memcpy(&redo, c->hostCtrl + kNaOffRedoCount, sizeof(redo));
    HIP_TRY(c, hipEventElapsedTime(&ms, c->evStart, c->evStop));
    c->lastTraceMs = usedPool ? c->pool.tracedMs() : ms;
    if (redo > 0) {
        // pixels the production kernel handed over (a ray with a zero direction component, or
        // a stack deeper than its LDS stack): the reference-order kernel renders exactly those
        if (redo > a.redoCap) redo = a.redoCap;       // (more than the list holds: the error flag says so below)
        float ms2 = 0.f;
        HIP_TRY(c, hipEventRecord(c->evStart, c->stream));
        if (a.mode == RTAMD_MODE_PT) {
            // one thread per (pixel, sample), then an in-order sum per pixel; in pieces of the list
            // when the sample buffer is the fixed one
            float4* samples = (float4*)(na + kNaOffSamples);
            int32_t piece = redo;
            if (noAlloc) piece = kNaSampleCap / a.spp;
            else {
                HIP_TRY(c, c->redoSamples.alloc((size_t)redo * (size_t)a.spp));
                samples = c->redoSamples.p;
            }
            for (int32_t at = 0; at < redo; at += piece) {
                const int32_t cnt = redo - at < piece ? redo - at : piece;
                const size_t n = (size_t)cnt * (size_t)a.spp;
                const int spread = listSpreadFor(n);
                hipLaunchKernelGGL(rt::kFaithful, dim3((unsigned)(((n << spread) + 63) / 64)), dim3(64), 0, c->stream, rsc, sky, a,
                                   (const int32_t*)a.redoList + at, cnt, samples, spread);
                HIP_TRY(c, hipGetLastError());
                hipLaunchKernelGGL(rt::kSumListedSamples, dim3((cnt + 63) / 64), dim3(64), 0, c->stream, a,
                                   (const int32_t*)a.redoList + at, cnt, (const float4*)samples);
            }
        } else {
            const int spread = listSpreadFor((size_t)redo);
            hipLaunchKernelGGL(rt::kFaithful, dim3((unsigned)((((size_t)redo << spread) + 63) / 64)), dim3(64), 0, c->stream, rsc, sky, a,
                               (const int32_t*)a.redoList, redo, (float4*)nullptr, spread);
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->evStop, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->hostCtrl, ctrl, kCtrlUsed, hipMemcpyDeviceToHost, c->stream));      // counters and flags now include the pass
        if (hostOutputs) HIP_TRY(c, copyOutputs());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipEventElapsedTime(&ms2, c->evStart, c->evStop));
        ms += ms2;
        handoverMs = ms2;
    } else if (hostOutputs) {
        HIP_TRY(c, copyOutputs());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    rt::DevCounters hc;
    unsigned long long hr[2];
    int32_t herr;
    std::memcpy(&hc, c->hostCtrl + kNaOffCounters, sizeof(hc));
    std::memcpy(hr, c->hostCtrl + kNaOffRayCount, sizeof(hr));
    std::memcpy(&herr, c->hostCtrl + kNaOffErrorFlag, sizeof(herr));
    {
        // pixels of the hand-over pass plus, in PT mode, the single paths the pool kernel passed on batch by batch
        int32_t paths = 0;
        std::memcpy(&paths, c->hostCtrl + kNaOffPathCount + 4, sizeof(paths));
        c->lastRedoPixels = redo + (usedPool ? paths : 0);
    }
    c->lastHandoverMs = handoverMs;
    {
        // The culling guard (RT_GUARD_FACTOR, kernel_fast.h).  The rays it fired on were handed over one by one and are
        // exact.  A single such hit in a billion rays is the tail of ordinary rounding (a hit a hair's breadth from the
        // ray's origin) and changes nothing else; a mesh that shows them at a rate -- at least kGuardMinHits in the frame
        // and more than one per million ray segments -- may also have hidden a hit from the kernel unseen: every instance
        // of it is switched to the reference-order kernel (until the next scene upload) and the frame is rendered once
        // more.  A frame that adds to the running sum is not rendered again (its samples are in the sum already): there
        // the switch takes effect from the next frame on.
        int32_t guardHits = 0;
        std::memcpy(&guardHits, c->hostCtrl + kNaOffGuardCount, sizeof(guardHits));
        if (!afterGuard) { c->lastGuardHits = guardHits; c->lastGuardRerendered = 0; }
        if (guardHits > 0 && !afterGuard && !faithful) {
            constexpr int32_t kGuardMinHits = 4;
            std::vector<int32_t> seen((size_t)c->fast.nBlasOfScene, 0);
            HIP_TRY(c, hipMemcpyAsync(seen.data(), c->fast.guardBlas.p, seen.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->fast.guardBlas.p, 0, seen.size() * sizeof(int32_t), c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            bool newlyMarked = false;
            for (size_t b = 0; b < seen.size(); b++)
                if (!c->fast.hDemoted[b] && seen[b] >= kGuardMinHits && (double)seen[b] > 1e-6 * (double)hr[0]) {
                    c->fast.hDemoted[b] = 1;
                    c->demotedMeshes++;
                    newlyMarked = true;
                }
            if (newlyMarked) HIP_TRY(c, c->fast.applyDemotions(c->tlas.p, c->objects.p, c->stream));
            if (newlyMarked && getenv("RTAMD_VERBOSE"))
                fprintf(stderr, "[rtamd] culling guard: %d hits within a quarter of the slack; %d of %d meshes now go to the reference-order kernel\n", guardHits,
                        c->demotedMeshes, c->fast.nBlasOfScene);
            const bool addsToSum = e.mode == RTAMD_MODE_PT && (e.flags & RTAMD_FLAG_ACCUMULATE);
            if (newlyMarked && !addsToSum) {
                const int rc = renderCore(c, frame, ext, width, height, out_rgba, out_prim_id, noAlloc, true);
                c->lastGuardRerendered = 1;
                return rc;
            }
        }
    }
    if (getenv("RTAMD_VERBOSE"))
        fprintf(stderr, "[rtamd] render %dx%d mode %d: %.3f ms, %d pixels handed to the reference-order kernel; nodes %zu (BLAS %zu), depth BLAS %d TLAS %d, instances %d\n",
                width, height, a.mode, ms, redo, c->fast.hNodes.size(), c->fast.nBlasNodes, c->fast.maxDepth, c->fast.tlasDepth, c->fast.nInst);
    c->lastKernelMs = ms;
    c->lastRays = hr[0];
    if (wantCounters) {
        rtamd_counters* oc = e.out_counters;
        oc->rays = hc.rays; oc->nodes = hc.nodes; oc->tris = hc.tris; oc->instances = hc.instances; oc->hits = hc.hits;
        oc->max_blas_stack = hc.maxBlasStack; oc->max_tlas_stack = hc.maxTlasStack; oc->reserved = hc.overflow;
    }
    if (herr & 8) return fail(c, RTAMD_ERR_BAD_INDEX, "bounds-checked kernel: an index outside the scene's arrays was about to be followed (device memory of the scene damaged, or a fault of the library)");
    if (herr & 1) return fail(c, RTAMD_ERR_STACK_OVERFLOW, "traversal stack overflow (scene deeper than the shader's stacks allow)");
    if (herr & 4) return fail(c, RTAMD_ERR_OUT_OF_MEMORY, noAlloc ? "rtamd_render_noalloc: more than 65536 pixels needed the reference-order kernel; render this frame with rtamd_render"
                                                                    : "hand-over list overflow");
    return RTAMD_OK;
}

// ---------------------------------------------------------------- several GPUs behind one context
namespace multi {

// Run `call` on every per-device context, each on its own host thread (the calls block; the threads live as long
// as the context); the first failure is reported with that device's message.
template <typename F>
int forEach(rtamd_ctx* c, F call) {
    const size_t n = c->subs.size();
    std::vector<int> rc(n, RTAMD_OK);
    for (size_t d = 1; d < n; d++) c->workers[d]->post([&call, c, d]() -> int { return call(c->subs[d]); });
    try { rc[0] = call(c->subs[0]); } catch (...) { rc[0] = -1000; }
    for (size_t d = 1; d < n; d++) rc[d] = c->workers[d]->wait();
    for (size_t d = 0; d < n; d++) {
        if (rc[d] == -1000) return fail(c, RTAMD_ERR_HIP, "device " + std::to_string(c->subs[d]->device) + ": internal error (exception on the device's host thread)");
        if (rc[d] != RTAMD_OK) return fail(c, rc[d], "device " + std::to_string(c->subs[d]->device) + ": " + c->subs[d]->error);
    }
    return RTAMD_OK;
}

// RCCL failed at run time: give its communicators up and let peer copies carry the exchange from here on.
static void dropRccl(rtamd_ctx* c, const std::string& why) {
    rt::RcclApi& api = rt::rccl();
    for (ncclComm_t comm : c->comms)
        if (comm) (void)(api.CommAbort ? api.CommAbort(comm) : api.CommDestroy(comm));
    c->comms.clear();
    c->rcclRanks = 0;
    c->exchangeNote = why + "; peer copies carry the exchange from that frame on";
    if (getenv("RTAMD_VERBOSE")) fprintf(stderr, "[rtamd] %s\n", c->exchangeNote.c_str());
}

#define RCCL_TRY(c, expr)                                                                                 \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess) return fail(c, RTAMD_ERR_HIP, std::string(#expr) + ": " + rt::rccl().GetErrorString(r_)); \
    } while (0)

// Move device d's packed piece (bytes) to the receive buffer on the first device.
static int gatherPieces(rtamd_ctx* c, const std::vector<size_t>& bytes) {
    const size_t n = c->subs.size();
    rtamd_ctx* root = c->subs[0];
    if (!c->comms.empty()) {
        rt::RcclApi& api = rt::rccl();
        ncclResult_t r = getenv("RTAMD_RCCL_FAIL") ? ncclInternalError /* (tests: as if the first call had failed) */ : api.GroupStart();
        if (r == ncclSuccess) {
            for (size_t d = 1; d < n && r == ncclSuccess; d++) {
                if (!bytes[d]) continue;
                r = api.Recv(c->recv[d].p, bytes[d], ncclUint8, (int)d, c->comms[0], root->stream);
                if (r == ncclSuccess) r = api.Send(c->piece[d].p, bytes[d], ncclUint8, 0, c->comms[d], c->subs[d]->stream);
            }
            const ncclResult_t e = api.GroupEnd();
            if (r == ncclSuccess) r = e;
        }
        if (r == ncclSuccess) return RTAMD_OK;
        dropRccl(c, std::string("ncclSend / ncclRecv failed at run time (") + api.GetErrorString(r) + ")");
    }
    for (size_t d = 1; d < n; d++) {
        if (!bytes[d]) continue;
        HIP_TRY(c, hipSetDevice(c->subs[d]->device));
        HIP_TRY(c, hipStreamSynchronize(c->subs[d]->stream));          // the piece is packed
        HIP_TRY(c, hipSetDevice(root->device));
        HIP_TRY(c, hipMemcpyPeerAsync(c->recv[d].p, root->device, c->piece[d].p, c->subs[d]->device, bytes[d], root->stream));
    }
    return RTAMD_OK;
}

template <typename T>
static int packOn(rtamd_ctx* c, size_t d, const T* frame, size_t offset, int w, int h, uint32_t nTiles) {
    rtamd_ctx* sub = c->subs[d];
    HIP_TRY(c, hipSetDevice(sub->device));
    const uint32_t n = nTiles * 1024u;
    hipLaunchKernelGGL((rt::kPackShard<T>), dim3((n + 255) / 256), dim3(256), 0, sub->stream, frame, (T*)(c->piece[d].p + offset), w, h, (int)d,
                       (int)c->subs.size(), nTiles);
    HIP_TRY(c, hipGetLastError());
    return RTAMD_OK;
}
template <typename T>
static int unpackOn(rtamd_ctx* c, size_t d, T* frame, size_t offset, int w, int h, uint32_t nTiles, const rt::RenderArgs& rect) {
    rtamd_ctx* root = c->subs[0];
    const uint32_t n = nTiles * 1024u;
    hipLaunchKernelGGL((rt::kUnpackShard<T>), dim3((n + 255) / 256), dim3(256), 0, root->stream, (const T*)(c->recv[d].p + offset), frame, w, h, (int)d,
                       (int)c->subs.size(), nTiles, rect.tileX, rect.tileY, rect.tileW, rect.tileH);
    HIP_TRY(c, hipGetLastError());
    return RTAMD_OK;
}

int render(rtamd_ctx* c, const rtamd_frame_cbuffer* frame, const rtamd_ext_params* ext, int32_t width, int32_t height,
           float* out_rgba, int32_t* out_prim_id) {
    if (!frame) return fail(c, RTAMD_ERR_INVALID_ARG, "frame is NULL");
    if (width <= 0 || height <= 0 || width > (1 << 20) || height > (1 << 20) || (int64_t)width * height > (1ll << 30)) return fail(c, RTAMD_ERR_INVALID_ARG, "bad image size");
    if (!c->haveScene) return fail(c, RTAMD_ERR_NO_SCENE, "render before a scene upload");
    rtamd_ext_params e;
    std::memset(&e, 0, sizeof(e));
    e.struct_size = sizeof(e); e.mode = RTAMD_MODE_REF; e.spp = 1;
    if (ext) {
        if (ext->struct_size != sizeof(rtamd_ext_params)) return fail(c, RTAMD_ERR_INVALID_ARG, "ext.struct_size mismatch");
        e = *ext;
    }
    if (e.shard_count > 1) return fail(c, RTAMD_ERR_INVALID_ARG, "a context over several GPUs shards the frame itself: leave shard_count at 0");
    const int n = (int)c->subs.size();
    const bool bySamples = (e.flags & RTAMD_FLAG_SHARD_SAMPLES) && e.mode == RTAMD_MODE_PT && n > 1;
    if (bySamples && (e.flags & RTAMD_FLAG_ACCUMULATE)) return fail(c, RTAMD_ERR_INVALID_ARG, "RTAMD_FLAG_ACCUMULATE needs tile sharding on a context over several GPUs");
    if (bySamples && e.spp < n) return fail(c, RTAMD_ERR_INVALID_ARG, "sample sharding needs at least one sample per GPU");
    if (e.mode == RTAMD_MODE_PT && (e.spp < 1 || e.max_bounces < 0 || e.sample_offset < 0))
        return fail(c, RTAMD_ERR_INVALID_ARG, "PT mode needs spp >= 1, max_bounces >= 0, sample_offset >= 0");
    const uint32_t callerFlags = e.flags;
    e.flags &= ~(uint32_t)RTAMD_FLAG_SHARD_SAMPLES;
    const size_t npix = (size_t)width * height;
    rtamd_ctx* root = c->subs[0];
    // rendered rectangle, as renderCore clips it
    rt::RenderArgs rect;
    std::memset(&rect, 0, sizeof(rect));
    rect.tileX = 0; rect.tileY = 0; rect.tileW = width; rect.tileH = height;
    bool partial = false;
    if (e.tile_w > 0 && e.tile_h > 0) {
        int x0 = e.tile_x < 0 ? 0 : e.tile_x, y0 = e.tile_y < 0 ? 0 : e.tile_y;
        int x1 = e.tile_x + e.tile_w > width ? width : e.tile_x + e.tile_w;
        int y1 = e.tile_y + e.tile_h > height ? height : e.tile_y + e.tile_h;
        rect.tileX = x0; rect.tileY = y0; rect.tileW = x1 > x0 ? x1 - x0 : 0; rect.tileH = y1 > y0 ? y1 - y0 : 0;
        partial = true;
    }
    float* const callerT = e.out_t;
    rtamd_counters* const callerCounters = e.out_counters;

    // per-device frames: every device renders into buffers on its own card (the staging buffers of its
    // context); the first device's are the ones the frame is assembled in
    for (int d = 0; d < n; d++) {
        rtamd_ctx* sub = c->subs[d];
        HIP_TRY(c, hipSetDevice(sub->device));
        const bool wantIds = out_prim_id && (d == 0 || !bySamples), wantT = callerT && (d == 0 || !bySamples);
        if (out_rgba) HIP_TRY(c, sub->outRgba.alloc(npix));
        if (wantIds) HIP_TRY(c, sub->outIds.alloc(npix * 2));
        if (wantT) HIP_TRY(c, sub->outT.alloc(npix));
        if (d == 0 && partial) {
            // pixels outside the rectangle keep the caller's values
            if (out_rgba) HIP_TRY(c, hipMemcpyAsync(sub->outRgba.p, out_rgba, npix * sizeof(float4), hipMemcpyDefault, sub->stream));
            if (out_prim_id) HIP_TRY(c, hipMemcpyAsync(sub->outIds.p, out_prim_id, npix * 2 * sizeof(int32_t), hipMemcpyDefault, sub->stream));
            if (callerT) HIP_TRY(c, hipMemcpyAsync(sub->outT.p, callerT, npix * sizeof(float), hipMemcpyDefault, sub->stream));
        }
        if (d > 0 && bySamples && out_rgba) HIP_TRY(c, hipMemsetAsync(sub->outRgba.p, 0, npix * sizeof(float4), sub->stream));   // un-rendered pixels add nothing
    }
    std::vector<rtamd_counters> counters((size_t)n);
    std::memset(counters.data(), 0, sizeof(rtamd_counters) * (size_t)n);
    const int rc = forEach(c, [&](rtamd_ctx* sub) -> int {
        int d = 0;
        while (c->subs[d] != sub) d++;
        rtamd_ext_params ed = e;
        const bool wantIds = out_prim_id && (d == 0 || !bySamples), wantT = callerT && (d == 0 || !bySamples);
        ed.out_t = wantT ? sub->outT.p : nullptr;
        ed.out_counters = callerCounters ? &counters[(size_t)d] : nullptr;
        if (bySamples) {
            const int base = e.spp / n, extra = e.spp % n;
            ed.spp = base + (d < extra ? 1 : 0);
            ed.sample_offset = e.sample_offset + d * base + (d < extra ? d : extra);
            ed.flags |= RTAMD_FLAG_OUTPUT_SUM;
        } else if (n > 1) {
            ed.shard_rank = d; ed.shard_count = n;
        }
        return renderCore(sub, frame, &ed, width, height, out_rgba ? (float*)sub->outRgba.p : nullptr, wantIds ? sub->outIds.p : nullptr, false);
    });
    if (rc != RTAMD_OK) return rc;

    // ---- the exchange step
    hipEvent_t ev0 = root->evStart, ev1 = root->evStop;
    HIP_TRY(c, hipSetDevice(root->device));
    HIP_TRY(c, hipEventRecord(ev0, root->stream));
    if (n > 1 && !bySamples) {
        // gather: rgba (16 B), ids (8 B), t (4 B) of a device's tiles go in one piece
        std::vector<size_t> bytes((size_t)n, 0);
        std::vector<uint32_t> tiles((size_t)n, 0);
        const size_t per = (out_rgba ? 16 : 0) + (out_prim_id ? 8 : 0) + (callerT ? 4 : 0);
        for (int d = 1; d < n; d++) {
            tiles[d] = rt::shardTileCount(width, height, d, n);
            bytes[d] = (size_t)tiles[d] * 1024 * per;
            if (!bytes[d]) continue;
            HIP_TRY(c, hipSetDevice(c->subs[d]->device));
            HIP_TRY(c, c->piece[d].alloc(bytes[d]));
            HIP_TRY(c, hipSetDevice(root->device));
            HIP_TRY(c, c->recv[d].alloc(bytes[d]));
            const size_t offIds = out_rgba ? (size_t)tiles[d] * 1024 * 16 : 0, offT = offIds + (out_prim_id ? (size_t)tiles[d] * 1024 * 8 : 0);
            int prc;
            if (out_rgba && (prc = packOn<float4>(c, d, c->subs[d]->outRgba.p, 0, width, height, tiles[d]))) return prc;
            if (out_prim_id && (prc = packOn<int2>(c, d, (const int2*)c->subs[d]->outIds.p, offIds, width, height, tiles[d]))) return prc;
            if (callerT && (prc = packOn<float>(c, d, c->subs[d]->outT.p, offT, width, height, tiles[d]))) return prc;
        }
        HIP_TRY(c, hipSetDevice(root->device));
        int grc = gatherPieces(c, bytes);
        if (grc) return grc;
        HIP_TRY(c, hipSetDevice(root->device));
        for (int d = 1; d < n; d++) {
            if (!bytes[d]) continue;
            const size_t offIds = out_rgba ? (size_t)tiles[d] * 1024 * 16 : 0, offT = offIds + (out_prim_id ? (size_t)tiles[d] * 1024 * 8 : 0);
            int prc;
            if (out_rgba && (prc = unpackOn<float4>(c, d, root->outRgba.p, 0, width, height, tiles[d], rect))) return prc;
            if (out_prim_id && (prc = unpackOn<int2>(c, d, (int2*)root->outIds.p, offIds, width, height, tiles[d], rect))) return prc;
            if (callerT && (prc = unpackOn<float>(c, d, root->outT.p, offT, width, height, tiles[d], rect))) return prc;
        }
    } else if (n > 1 && out_rgba) {
        // reduce the sum buffers onto the first device
        bool reduced = false;
        if (!c->comms.empty()) {
            rt::RcclApi& api = rt::rccl();
            ncclResult_t r = getenv("RTAMD_RCCL_FAIL") ? ncclInternalError : api.GroupStart();
            if (r == ncclSuccess) {
                for (int d = 0; d < n && r == ncclSuccess; d++)
                    r = api.Reduce(c->subs[d]->outRgba.p, d == 0 ? root->outRgba.p : nullptr, npix * 4, ncclFloat32, ncclSum, 0, c->comms[d], c->subs[d]->stream);
                const ncclResult_t e2 = api.GroupEnd();
                if (r == ncclSuccess) r = e2;
            }
            if (r == ncclSuccess) reduced = true;
            else dropRccl(c, std::string("ncclReduce failed at run time (") + api.GetErrorString(r) + ")");
        }
        if (!reduced) {
            HIP_TRY(c, c->recv[1].alloc(npix * sizeof(float4)));
            for (int d = 1; d < n; d++) {
                HIP_TRY(c, hipSetDevice(c->subs[d]->device));
                HIP_TRY(c, hipStreamSynchronize(c->subs[d]->stream));
                HIP_TRY(c, hipSetDevice(root->device));
                HIP_TRY(c, hipMemcpyPeerAsync(c->recv[1].p, root->device, c->subs[d]->outRgba.p, c->subs[d]->device, npix * sizeof(float4), root->stream));
                hipLaunchKernelGGL(rt::kAddSums, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, root->stream, root->outRgba.p, (const float4*)c->recv[1].p, npix);
                HIP_TRY(c, hipGetLastError());
            }
        }
        if (!(callerFlags & RTAMD_FLAG_OUTPUT_SUM)) {
            hipLaunchKernelGGL(rt::kSumsToMean, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, root->stream, root->outRgba.p, npix, width, rect.tileX, rect.tileY,
                               rect.tileW, rect.tileH);
            HIP_TRY(c, hipGetLastError());
        }
    } else if (bySamples && out_rgba && !(callerFlags & RTAMD_FLAG_OUTPUT_SUM)) {
        hipLaunchKernelGGL(rt::kSumsToMean, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, root->stream, root->outRgba.p, npix, width, rect.tileX, rect.tileY,
                           rect.tileW, rect.tileH);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipEventRecord(ev1, root->stream));
    // hand the assembled frame to the caller (host memory, or device memory on any GPU)
    if (out_rgba) HIP_TRY(c, hipMemcpyAsync(out_rgba, root->outRgba.p, npix * sizeof(float4), hipMemcpyDefault, root->stream));
    if (out_prim_id) HIP_TRY(c, hipMemcpyAsync(out_prim_id, root->outIds.p, npix * 2 * sizeof(int32_t), hipMemcpyDefault, root->stream));
    if (callerT) HIP_TRY(c, hipMemcpyAsync(callerT, root->outT.p, npix * sizeof(float), hipMemcpyDefault, root->stream));
    for (int d = n - 1; d >= 0; d--) {
        HIP_TRY(c, hipSetDevice(c->subs[d]->device));
        HIP_TRY(c, hipStreamSynchronize(c->subs[d]->stream));
    }
    float xms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&xms, ev0, ev1));
    c->lastExchangeMs = xms;
    // the slowest device sets the frame time; work adds up
    c->lastKernelMs = c->lastTraceMs = c->lastHandoverMs = 0.f;
    c->lastRays = 0; c->lastRedoPixels = 0;
    for (rtamd_ctx* sub : c->subs) {
        c->lastKernelMs = sub->lastKernelMs > c->lastKernelMs ? sub->lastKernelMs : c->lastKernelMs;
        c->lastTraceMs = sub->lastTraceMs > c->lastTraceMs ? sub->lastTraceMs : c->lastTraceMs;
        c->lastHandoverMs = sub->lastHandoverMs > c->lastHandoverMs ? sub->lastHandoverMs : c->lastHandoverMs;
        c->lastRays += sub->lastRays;
        c->lastRedoPixels += sub->lastRedoPixels;
    }
    if (callerCounters) {
        rtamd_counters t;
        std::memset(&t, 0, sizeof(t));
        for (const rtamd_counters& k : counters) {
            t.rays += k.rays; t.nodes += k.nodes; t.tris += k.tris; t.instances += k.instances; t.hits += k.hits; t.reserved += k.reserved;
            t.max_blas_stack = k.max_blas_stack > t.max_blas_stack ? k.max_blas_stack : t.max_blas_stack;
            t.max_tlas_stack = k.max_tlas_stack > t.max_tlas_stack ? k.max_tlas_stack : t.max_tlas_stack;
        }
        *callerCounters = t;
    }
    return RTAMD_OK;
}

}  // namespace multi

extern "C" {

int rtamd_ctx_create_multi(const int32_t* device_ordinals, int32_t n_devices, rtamd_ctx** out_ctx) {
    return guarded(nullptr, [&]() -> int {
    if (!out_ctx) return fail(nullptr, RTAMD_ERR_INVALID_ARG, "out_ctx is NULL");
    *out_ctx = nullptr;
    if (!device_ordinals || n_devices < 1 || n_devices > kMaxDevices) return fail(nullptr, RTAMD_ERR_INVALID_ARG, "device list is NULL, empty or longer than 64");
    rtamd_ctx* c = new (std::nothrow) rtamd_ctx();
    if (!c) return fail(nullptr, RTAMD_ERR_OUT_OF_MEMORY, "host allocation failed");
    c->device = device_ordinals[0];
    bool distinct = true;
    for (int d = 0; d < n_devices; d++) {
        rtamd_ctx* sub = nullptr;
        const int rc = rtamd_ctx_create(device_ordinals[d], &sub);
        if (rc != RTAMD_OK) {
            const std::string why = g_createError;
            for (rtamd_ctx* made : c->subs) (void)rtamd_ctx_destroy(made);
            delete c;
            return fail(nullptr, rc, "device " + std::to_string(device_ordinals[d]) + ": " + why);
        }
        c->subs.push_back(sub);
        for (int k = 0; k < d; k++) distinct = distinct && device_ordinals[k] != device_ordinals[d];
    }
    try {
        c->workers.resize((size_t)n_devices);
        for (int d = 1; d < n_devices; d++) c->workers[(size_t)d] = std::make_unique<rt::DeviceWorker>();
        c->piece.resize((size_t)n_devices);
        c->recv.resize((size_t)n_devices < 2 ? 2 : (size_t)n_devices);
    } catch (...) {
        c->workers.clear();
        for (rtamd_ctx* made : c->subs) (void)rtamd_ctx_destroy(made);
        delete c;
        return fail(nullptr, RTAMD_ERR_OUT_OF_MEMORY, "could not start a host thread per device");
    }
    // peer access for the copies that carry the exchange when RCCL does not
    for (int d = 1; d < n_devices; d++)
        if (device_ordinals[d] != device_ordinals[0]) {
            (void)hipSetDevice(device_ordinals[0]);
            if (hipDeviceEnablePeerAccess(device_ordinals[d], 0) != hipSuccess) (void)hipGetLastError();
            (void)hipSetDevice(device_ordinals[d]);
            if (hipDeviceEnablePeerAccess(device_ordinals[0], 0) != hipSuccess) (void)hipGetLastError();
        }
    // RCCL communicator over the devices (one rank per device, this process holds them all)
    const char* off = getenv("RTAMD_NO_RCCL");
    if (off && atoi(off) != 0) c->exchangeNote = "RCCL switched off by RTAMD_NO_RCCL";
    else if (!distinct) c->exchangeNote = "a device is listed twice: RCCL takes one rank per GPU, peer copies carry the exchange";
    else {
        std::string why;
        if (rt::rccl().load(why)) {
            std::vector<int> devs(device_ordinals, device_ordinals + n_devices);
            c->comms.assign((size_t)n_devices, nullptr);
            const ncclResult_t r = rt::rccl().CommInitAll(c->comms.data(), n_devices, devs.data());
            if (r != ncclSuccess) {
                c->exchangeNote = std::string("ncclCommInitAll: ") + rt::rccl().GetErrorString(r) + "; peer copies carry the exchange";
                c->comms.clear();
            } else if (rt::rccl().CommCount(c->comms[0], &c->rcclRanks) != ncclSuccess) c->rcclRanks = n_devices;
        } else c->exchangeNote = why + "; peer copies carry the exchange";
    }
    (void)hipSetDevice(device_ordinals[0]);
    *out_ctx = c;
    return RTAMD_OK;
    });
}

int rtamd_multi_info(const rtamd_ctx* c, int32_t* n_devices, int32_t* rccl_ranks, float* last_exchange_ms) {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (n_devices) *n_devices = c->subs.empty() ? 1 : (int32_t)c->subs.size();
    if (rccl_ranks) *rccl_ranks = c->rcclRanks;
    if (last_exchange_ms) *last_exchange_ms = c->lastExchangeMs;
    return RTAMD_OK;
}

const char* rtamd_multi_exchange_note(const rtamd_ctx* c) { return c ? c->exchangeNote.c_str() : ""; }

int rtamd_render(rtamd_ctx* c, const rtamd_frame_cbuffer* frame, const rtamd_ext_params* ext, int32_t width,
                 int32_t height, float* out_rgba, int32_t* out_prim_id) {
    return guarded(c, [&]() -> int {
    if (c && !c->subs.empty()) return multi::render(c, frame, ext, width, height, out_rgba, out_prim_id);
    return renderCore(c, frame, ext, width, height, out_rgba, out_prim_id, false);
    });
}

int rtamd_render_noalloc(rtamd_ctx* c, const rtamd_frame_cbuffer* frame, const rtamd_ext_params* ext, int32_t width,
                         int32_t height, float* out_rgba_device, int32_t* out_prim_id_device) {
    return guarded(c, [&]() -> int {
    if (c && !c->subs.empty()) return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_render_noalloc: not available on a context over several GPUs (the exchange buffers are sized to the frame)");
    return renderCore(c, frame, ext, width, height, out_rgba_device, out_prim_id_device, true);
    });
}

// ---- frames that return before they are complete ----------------------------------------------------------------
// rtamd_render_begin queues a frame and returns; rtamd_render_wait completes the oldest frame begun.  With two frames
// in flight the second frame's trace kernel fills the chip while the first one's last paths drain, and the first
// frame's hand-over paths and summation run on the context's second stream beside it (path tracing), and the host's
// part of a frame (read-back, the decision about a hand-over pass) overlaps the next frame's kernels (REF mode).
int rtamd_render_begin(rtamd_ctx* c, const rtamd_frame_cbuffer* frame, const rtamd_ext_params* ext, int32_t width, int32_t height,
                       float* out_rgba_device, int32_t* out_prim_id_device) {
    return guarded(c, [&]() -> int {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (!c->subs.empty()) return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_render_begin: not available on a context over several GPUs");
    if (!frame) return fail(c, RTAMD_ERR_INVALID_ARG, "frame is NULL");
    if (width <= 0 || height <= 0 || width > (1 << 20) || height > (1 << 20) || (int64_t)width * height > (1ll << 30)) return fail(c, RTAMD_ERR_INVALID_ARG, "bad image size");
    if (!c->haveScene) return fail(c, RTAMD_ERR_NO_SCENE, "render before a scene upload");
    if (c->asyncInFlight >= 2) return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_render_begin: two frames are in flight already, call rtamd_render_wait first");
    rtamd_ext_params e;
    std::memset(&e, 0, sizeof(e));
    e.struct_size = sizeof(e); e.mode = RTAMD_MODE_REF; e.spp = 1;
    if (ext) {
        if (ext->struct_size != sizeof(rtamd_ext_params)) return fail(c, RTAMD_ERR_INVALID_ARG, "ext.struct_size mismatch");
        e = *ext;
    }
    if (e.mode != RTAMD_MODE_REF && e.mode != RTAMD_MODE_PT) return fail(c, RTAMD_ERR_INVALID_ARG, "unknown mode");
    if (e.mode == RTAMD_MODE_PT && (e.spp < 1 || e.max_bounces < 0 || e.sample_offset < 0))
        return fail(c, RTAMD_ERR_INVALID_ARG, "PT mode needs spp >= 1, max_bounces >= 0, sample_offset >= 0");
    if (e.shard_count > 1 && (e.shard_rank < 0 || e.shard_rank >= e.shard_count)) return fail(c, RTAMD_ERR_INVALID_ARG, "shard_rank out of range");
    if (e.flags & (RTAMD_FLAG_COUNTERS | RTAMD_FLAG_REFERENCE_ORDER | RTAMD_FLAG_DEBUG_CHECKS)) return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_render_begin: counters, reference order and the bounds-checked variant are for rtamd_render");
    if ((out_rgba_device && !isDevicePointer(out_rgba_device)) || (out_prim_id_device && !isDevicePointer(out_prim_id_device)) || (e.out_t && !isDevicePointer(e.out_t)))
        return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_render_begin: outputs must be device memory");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t npix = (size_t)width * height;
    rtamd_ctx::FrameSlot& S = c->slots[c->asyncBegun & 1];
    if (!S.evDone) {
        HIP_TRY(c, hipEventCreate(&S.evStart));
        HIP_TRY(c, hipEventCreate(&S.evStop));
        HIP_TRY(c, hipEventCreateWithFlags(&S.evDone, hipEventDisableTiming));
        HIP_TRY(c, S.ctrl.alloc(kCtrlBytes));
        HIP_TRY(c, hipHostMalloc((void**)&S.hostCtrl, kCtrlBytes, hipHostMallocDefault));
    }
    const size_t maskWords = (npix + 31) / 32;
    HIP_TRY(c, S.redoMask.alloc(maskWords));
    HIP_TRY(c, S.redoList.alloc(npix));
    unsigned char* const ctrl = S.ctrl.p;

    rt::RenderArgs a;
    std::memset(&a, 0, sizeof(a));
    a.cb = *frame;
    a.width = width; a.height = height;
    a.mode = e.mode; a.spp = e.mode == RTAMD_MODE_PT ? e.spp : 1; a.maxBounces = e.max_bounces;
    a.seedHi = (uint32_t)(e.seed >> 32); a.seedLo = (uint32_t)e.seed;
    a.sampleOffset = e.sample_offset;
    a.tileX = 0; a.tileY = 0; a.tileW = width; a.tileH = height;
    if (e.tile_w > 0 && e.tile_h > 0) {
        int x0 = e.tile_x < 0 ? 0 : e.tile_x, y0 = e.tile_y < 0 ? 0 : e.tile_y;
        int x1 = e.tile_x + e.tile_w > width ? width : e.tile_x + e.tile_w;
        int y1 = e.tile_y + e.tile_h > height ? height : e.tile_y + e.tile_h;
        a.tileX = x0; a.tileY = y0; a.tileW = x1 > x0 ? x1 - x0 : 0; a.tileH = y1 > y0 ? y1 - y0 : 0;
    }
    a.shardRank = e.shard_rank; a.shardCount = e.shard_count;
    a.flags = e.flags;
    a.outRgba = (float4*)out_rgba_device; a.outIds = out_prim_id_device; a.outT = e.out_t;
    if (e.mode == RTAMD_MODE_PT && (e.flags & RTAMD_FLAG_ACCUMULATE)) {
        if (c->accumW != width || c->accumH != height || !c->accum.p) {
            if (c->asyncInFlight > 0) return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_render_begin: the running sum changes size while a frame is in flight");
            HIP_TRY(c, c->accum.alloc(npix));
            HIP_TRY(c, hipMemsetAsync(c->accum.p, 0, npix * sizeof(float4), c->stream));
            c->accumW = width; c->accumH = height;
        }
        a.accum = c->accum.p;
    }
    a.counters = nullptr;
    a.rayCount = (unsigned long long*)(ctrl + kNaOffRayCount);
    a.errorFlag = (int32_t*)(ctrl + kNaOffErrorFlag);
    a.redoMask = S.redoMask.p; a.redoList = S.redoList.p; a.redoCap = (int32_t)npix;
    a.redoCount = (int32_t*)(ctrl + kNaOffRedoCount);

    rt::Sky sky;
    sky.texels = c->sky.p; sky.w = c->skyW; sky.h = c->skyH;
    sky.invW = c->skyW > 0 ? 1.0f / (float)c->skyW : 0.f; sky.invH = c->skyH > 0 ? 1.0f / (float)c->skyH : 0.f;
    rt::RefScene rsc;
    rsc.tlas = c->tlas.p; rsc.blasOffsets = c->blasOffsets.p; rsc.blas = c->blas.p; rsc.vtxOffsets = c->vtxOffsets.p;
    rsc.verts = c->verts.p; rsc.triOffsets = c->triOffsets.p; rsc.tris = c->tris.p; rsc.objects = c->objects.p;
    rsc.guardCount = (int32_t*)(ctrl + kNaOffGuardCount); rsc.guardBlas = c->fast.guardBlas.p;

    HIP_TRY(c, hipMemsetAsync(a.redoMask, 0, maskWords * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(ctrl, 0, kCtrlBytes, c->stream));
    HIP_TRY(c, hipEventRecord(S.evStart, c->stream));
    hipStream_t last = c->stream;
    S.usedPool = false;
    c->fast.asyncInFlight = c->asyncInFlight;
    if (a.tileW > 0 && a.tileH > 0) {
        S.usedPool = a.mode == RTAMD_MODE_PT && rt::fastLaunchConfig().pool;
        hipError_t le = S.usedPool ? rt::launchPool(c->fast, c->pool, sky, a, c->stream, rsc, (int32_t*)(ctrl + kNaOffPathCount), (int32_t*)(ctrl + kNaOffPathCount + 8),
                                                    (int32_t*)(ctrl + kNaOffPathCount + 4), c->asyncBegun & 1, &last)
                                   : rt::launchFast(c->fast, sky, a, c->stream, nullptr);
        if (le == hipErrorNotSupported) return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_render_begin: the frame does not fit one batch of paths; render it with rtamd_render");
        if (le != hipSuccess) return failHip(c, le, "production kernel launch");
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipEventRecord(S.evStop, c->stream));
    HIP_TRY(c, hipMemcpyAsync(S.hostCtrl, ctrl, kCtrlUsed, hipMemcpyDeviceToHost, last));
    HIP_TRY(c, hipEventRecord(S.evDone, last));
    S.a = a; S.frame = *frame; S.e = e; S.width = width; S.height = height; S.outRgba = out_rgba_device; S.outIds = out_prim_id_device;
    c->asyncBegun++;
    c->asyncInFlight++;
    c->fast.asyncInFlight = c->asyncInFlight;
    return RTAMD_OK;
    });
}

int rtamd_render_wait(rtamd_ctx* c) {
    return guarded(c, [&]() -> int {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (c->asyncInFlight <= 0) return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_render_wait: no frame in flight");
    HIP_TRY(c, hipSetDevice(c->device));
    rtamd_ctx::FrameSlot& S = c->slots[(c->asyncBegun - c->asyncInFlight) & 1];
    auto done = [&]() { c->asyncInFlight--; c->fast.asyncInFlight = c->asyncInFlight; };
    hipError_t we = hipEventSynchronize(S.evDone);
    if (we != hipSuccess) { done(); return failHip(c, we, "waiting for the frame"); }
    const rt::RenderArgs& a = S.a;
    int32_t redo = 0;
    std::memcpy(&redo, S.hostCtrl + kNaOffRedoCount, sizeof(redo));
    float ms = 0.f, handoverMs = 0.f;
    (void)hipEventElapsedTime(&ms, S.evStart, S.evStop);
    if (S.usedPool) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, c->pool.asyncBegin[(c->asyncBegun - c->asyncInFlight) & 1], c->pool.asyncEnd[(c->asyncBegun - c->asyncInFlight) & 1]) == hipSuccess) c->lastTraceMs = t;
    } else c->lastTraceMs = ms;
    unsigned char* const ctrl = S.ctrl.p;
    if (redo > 0) {
        // pixels the production kernel handed over whole: the reference-order kernel renders exactly those, on the
        // caller's stream (behind whatever the next frame has queued there already)
        if (redo > a.redoCap) redo = a.redoCap;
        rt::Sky sky;
        sky.texels = c->sky.p; sky.w = c->skyW; sky.h = c->skyH;
        sky.invW = c->skyW > 0 ? 1.0f / (float)c->skyW : 0.f; sky.invH = c->skyH > 0 ? 1.0f / (float)c->skyH : 0.f;
        rt::RefScene rsc;
        rsc.tlas = c->tlas.p; rsc.blasOffsets = c->blasOffsets.p; rsc.blas = c->blas.p; rsc.vtxOffsets = c->vtxOffsets.p;
        rsc.verts = c->verts.p; rsc.triOffsets = c->triOffsets.p; rsc.tris = c->tris.p; rsc.objects = c->objects.p;
        rsc.guardCount = (int32_t*)(ctrl + kNaOffGuardCount); rsc.guardBlas = c->fast.guardBlas.p;
        auto handOver = [&]() -> int {
            HIP_TRY(c, hipEventRecord(c->evStart, c->stream));
            if (a.mode == RTAMD_MODE_PT) {
                HIP_TRY(c, c->redoSamples.alloc((size_t)redo * (size_t)a.spp));
                const size_t n = (size_t)redo * (size_t)a.spp;
                const int spread = listSpreadFor(n);
                hipLaunchKernelGGL(rt::kFaithful, dim3((unsigned)(((n << spread) + 63) / 64)), dim3(64), 0, c->stream, rsc, sky, a, (const int32_t*)a.redoList, redo,
                                   c->redoSamples.p, spread);
                HIP_TRY(c, hipGetLastError());
                hipLaunchKernelGGL(rt::kSumListedSamples, dim3((redo + 63) / 64), dim3(64), 0, c->stream, a, (const int32_t*)a.redoList, redo, (const float4*)c->redoSamples.p);
            } else {
                const int spread = listSpreadFor((size_t)redo);
                hipLaunchKernelGGL(rt::kFaithful, dim3((unsigned)((((size_t)redo << spread) + 63) / 64)), dim3(64), 0, c->stream, rsc, sky, a, (const int32_t*)a.redoList, redo,
                                   (float4*)nullptr, spread);
            }
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipEventRecord(c->evStop, c->stream));
            HIP_TRY(c, hipMemcpyAsync(S.hostCtrl, ctrl, kCtrlUsed, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipEventRecord(S.evDone, c->stream));
            HIP_TRY(c, hipEventSynchronize(S.evDone));
            HIP_TRY(c, hipEventElapsedTime(&handoverMs, c->evStart, c->evStop));
            return RTAMD_OK;
        };
        const int hrc = handOver();
        if (hrc != RTAMD_OK) { done(); return hrc; }
        ms += handoverMs;
    }
    unsigned long long hr[2];
    int32_t herr, paths = 0, guardHits = 0;
    std::memcpy(hr, S.hostCtrl + kNaOffRayCount, sizeof(hr));
    std::memcpy(&herr, S.hostCtrl + kNaOffErrorFlag, sizeof(herr));
    std::memcpy(&paths, S.hostCtrl + kNaOffPathCount + 4, sizeof(paths));
    std::memcpy(&guardHits, S.hostCtrl + kNaOffGuardCount, sizeof(guardHits));
    c->lastRedoPixels = redo + (S.usedPool ? paths : 0);
    c->lastHandoverMs = handoverMs;
    c->lastKernelMs = ms;
    c->lastRays = hr[0];
    c->lastGuardHits = guardHits; c->lastGuardRerendered = 0;
    done();
    if (guardHits > 0) {
        // the culling guard, as in rtamd_render: a newly marked mesh means this frame is rendered again, in a blocking
        // call, once the other frame in flight (rendered before the mark) has left the GPU; that frame's own
        // rtamd_render_wait looks at its own hits
        constexpr int32_t kGuardMinHits = 4;
        std::vector<int32_t> seen((size_t)c->fast.nBlasOfScene, 0);
        HIP_TRY(c, hipMemcpyAsync(seen.data(), c->fast.guardBlas.p, seen.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->fast.guardBlas.p, 0, seen.size() * sizeof(int32_t), c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        bool newlyMarked = false;
        for (size_t b = 0; b < seen.size(); b++)
            if (!c->fast.hDemoted[b] && seen[b] >= kGuardMinHits && (double)seen[b] > 1e-6 * (double)hr[0]) {
                c->fast.hDemoted[b] = 1;
                c->demotedMeshes++;
                newlyMarked = true;
            }
        if (newlyMarked) {
            if (c->pool.side) HIP_TRY(c, hipStreamSynchronize(c->pool.side));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, c->fast.applyDemotions(c->tlas.p, c->objects.p, c->stream));
            if (!(S.e.mode == RTAMD_MODE_PT && (S.e.flags & RTAMD_FLAG_ACCUMULATE))) {
                const int rc = renderCore(c, &S.frame, &S.e, S.width, S.height, S.outRgba, S.outIds, false, true);
                c->lastGuardRerendered = 1;
                return rc;
            }
        }
    }
    if (herr & 1) return fail(c, RTAMD_ERR_STACK_OVERFLOW, "traversal stack overflow (scene deeper than the shader's stacks allow)");
    if (herr & 4) return fail(c, RTAMD_ERR_OUT_OF_MEMORY, "hand-over list overflow");
    return RTAMD_OK;
    });
}

// The running sum of RTAMD_FLAG_ACCUMULATE frames (rgb sums and the sample count per pixel) out to / in from the
// host: what a progressive render needs to stop and go on later, together with the next sample_offset.
int rtamd_accum_download(rtamd_ctx* c, float* sums_host, int32_t width, int32_t height) {
    return guarded(c, [&]() -> int {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (!c->subs.empty()) return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_accum_download: on a context over several GPUs every device keeps the sums of its own tiles");
    NO_FRAME_IN_FLIGHT(c, "rtamd_accum_download");
    if (!sums_host || width <= 0 || height <= 0) return fail(c, RTAMD_ERR_INVALID_ARG, "NULL buffer or bad size");
    if (!c->accum.p || c->accumW != width || c->accumH != height) return fail(c, RTAMD_ERR_INVALID_ARG, "no running sum of this size");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(sums_host, c->accum.p, (size_t)width * height * sizeof(float4), hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTAMD_OK;
    });
}

int rtamd_accum_upload(rtamd_ctx* c, const float* sums_host, int32_t width, int32_t height) {
    return guarded(c, [&]() -> int {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (!c->subs.empty()) return fail(c, RTAMD_ERR_INVALID_ARG, "rtamd_accum_upload: on a context over several GPUs every device keeps the sums of its own tiles");
    NO_FRAME_IN_FLIGHT(c, "rtamd_accum_upload");
    if (!sums_host || width <= 0 || height <= 0 || (int64_t)width * height > (1ll << 30)) return fail(c, RTAMD_ERR_INVALID_ARG, "NULL buffer or bad size");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, c->accum.alloc((size_t)width * height));
    c->accumW = width; c->accumH = height;
    HIP_TRY(c, hipMemcpyAsync(c->accum.p, sums_host, (size_t)width * height * sizeof(float4), hipMemcpyDefault, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTAMD_OK;
    });
}

int rtamd_last_trace_kernel_ms(const rtamd_ctx* c, float* ms) {
    if (!c || !ms) return RTAMD_ERR_INVALID_ARG;
    *ms = c->lastTraceMs;
    return RTAMD_OK;
}

int rtamd_last_handover_stats(const rtamd_ctx* c, float* ms, int32_t* pixels) {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (ms) *ms = c->lastHandoverMs;
    if (pixels) *pixels = c->lastRedoPixels;
    return RTAMD_OK;
}

int rtamd_last_guard_stats(const rtamd_ctx* c, int32_t* hits_near_slack, int32_t* demoted_meshes, int32_t* rerendered) {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    int32_t hits = c->lastGuardHits, meshes = c->demotedMeshes, again = c->lastGuardRerendered;
    if (!c->subs.empty()) {
        hits = meshes = again = 0;
        for (const rtamd_ctx* sub : c->subs) {
            hits += sub->lastGuardHits;
            meshes = sub->demotedMeshes > meshes ? sub->demotedMeshes : meshes;
            again |= sub->lastGuardRerendered;
        }
    }
    if (hits_near_slack) *hits_near_slack = hits;
    if (demoted_meshes) *demoted_meshes = meshes;
    if (rerendered) *rerendered = again;
    return RTAMD_OK;
}

int rtamd_last_render_stats(const rtamd_ctx* c, float* ms, uint64_t* rays) {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (ms) *ms = c->lastKernelMs;
    if (rays) *rays = c->lastRays;
    return RTAMD_OK;
}

}  // extern "C"

namespace rt {
// AddShader: Blend SrcAlpha One, no depth (Assets/Shaders/AddShader.shader:11,43-46), onto an
// 8-bit UNORM target (RayTracingPass.cs:58).
__global__ void kCompositeAdd(const float4* src, uchar4* dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 s = src[i];
    uchar4 d = dst[i];
    float r = saturate((float)d.x / 255.0f + s.x * s.w);
    float g = saturate((float)d.y / 255.0f + s.y * s.w);
    float b = saturate((float)d.z / 255.0f + s.z * s.w);
    float al = saturate((float)d.w / 255.0f + s.w * s.w);
    dst[i] = make_uchar4((unsigned char)__builtin_floorf(r * 255.0f + 0.5f), (unsigned char)__builtin_floorf(g * 255.0f + 0.5f),
                         (unsigned char)__builtin_floorf(b * 255.0f + 0.5f), (unsigned char)__builtin_floorf(al * 255.0f + 0.5f));
}
}  // namespace rt

extern "C" int rtamd_composite_add(rtamd_ctx* c, const float* src, uint8_t* dst, int32_t width, int32_t height) {
    return guarded(c, [&]() -> int {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (!c->subs.empty()) return rtamd_composite_add(c->subs[0], src, dst, width, height);
    if (!src || !dst || width <= 0 || height <= 0) return fail(c, RTAMD_ERR_INVALID_ARG, "NULL buffer or bad size");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t n = (size_t)width * height;
    bool sDev = isDevicePointer(src), dDev = isDevicePointer(dst);
    if (sDev != dDev) return fail(c, RTAMD_ERR_INVALID_ARG, "src and dst must both be host or both be device memory");
    const float4* ds = (const float4*)src;
    uchar4* dd = (uchar4*)dst;
    DevBuf<uchar4>& tmp = c->tmp8;
    if (!sDev) {
        HIP_TRY(c, c->outRgba.upload((const float4*)src, n, c->stream));
        HIP_TRY(c, tmp.upload((const uchar4*)dst, n, c->stream));
        ds = c->outRgba.p; dd = tmp.p;
    }
    hipLaunchKernelGGL(rt::kCompositeAdd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ds, dd, n);
    HIP_TRY(c, hipGetLastError());
    if (!sDev) HIP_TRY(c, hipMemcpyAsync(dst, dd, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTAMD_OK;
    });
}

namespace rt {
__global__ void kPresent(const float4* src, uchar4* dst, size_t n, int tonemap, float exposure, float invGamma) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 s = src[i];
    float c[3] = {s.x * exposure, s.y * exposure, s.z * exposure};
    unsigned char q[3];
    for (int k = 0; k < 3; k++) {
        float x = c[k];
        if (tonemap == RTAMD_TONEMAP_REINHARD) x = x / (1.0f + x);
        x = saturate(x);
        if (invGamma != 1.0f) x = __builtin_powf(x, invGamma);
        q[k] = (unsigned char)__builtin_floorf(x * 255.0f + 0.5f);
    }
    dst[i] = make_uchar4(q[0], q[1], q[2], (unsigned char)__builtin_floorf(saturate(s.w) * 255.0f + 0.5f));
}
}  // namespace rt

extern "C" int rtamd_present(rtamd_ctx* c, const float* src, uint8_t* dst, int32_t width, int32_t height, int32_t tonemap,
                             float exposure, float gamma) {
    return guarded(c, [&]() -> int {
    if (!c) return RTAMD_ERR_INVALID_ARG;
    if (!c->subs.empty()) return rtamd_present(c->subs[0], src, dst, width, height, tonemap, exposure, gamma);
    if (!src || !dst || width <= 0 || height <= 0) return fail(c, RTAMD_ERR_INVALID_ARG, "NULL buffer or bad size");
    if (tonemap != RTAMD_TONEMAP_CLAMP && tonemap != RTAMD_TONEMAP_REINHARD) return fail(c, RTAMD_ERR_INVALID_ARG, "unknown tone map");
    if (!(exposure >= 0.0f) || !(gamma >= 0.0f)) return fail(c, RTAMD_ERR_INVALID_ARG, "exposure and gamma must be >= 0");
    HIP_TRY(c, hipSetDevice(c->device));
    size_t n = (size_t)width * height;
    bool sDev = isDevicePointer(src), dDev = isDevicePointer(dst);
    if (sDev != dDev) return fail(c, RTAMD_ERR_INVALID_ARG, "src and dst must both be host or both be device memory");
    const float4* ds = (const float4*)src;
    uchar4* dd = (uchar4*)dst;
    DevBuf<uchar4>& tmp = c->tmp8;
    if (!sDev) {
        HIP_TRY(c, c->outRgba.upload((const float4*)src, n, c->stream));
        HIP_TRY(c, tmp.alloc(n));
        ds = c->outRgba.p; dd = tmp.p;
    }
    const float invGamma = (gamma == 0.0f || gamma == 1.0f) ? 1.0f : 1.0f / gamma;
    hipLaunchKernelGGL(rt::kPresent, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, ds, dd, n, tonemap, exposure, invGamma);
    HIP_TRY(c, hipGetLastError());
    if (!sDev) HIP_TRY(c, hipMemcpyAsync(dst, dd, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RTAMD_OK;
    });
}

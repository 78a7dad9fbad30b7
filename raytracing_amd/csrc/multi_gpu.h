// Several GPUs behind one context: what the exchange step needs besides the per-device renders.
//
// The image partitions (RayTracingShader.compute:431-444 computes every pixel on its own), the
// scene is read-only and replicated on every GPU, and only the framebuffer is exchanged, once per
// frame:
//   tile sharding    device d renders the 32x32-pixel tiles t with t % N == d; its tiles are packed
//                    into one contiguous piece (1/N of the frame), sent to the first device and
//                    unpacked there — a gather, bit-identical to a one-GPU render;
//   sample sharding  device d renders its share of the samples of every pixel as raw sums; the sum
//                    buffers are reduced (SUM) onto the first device and divided by the sample
//                    count there — equal to a one-GPU render up to fp32 summation order.
// Transport is RCCL (ncclSend / ncclRecv in one group for the gather, ncclReduce for the sums) when
// librccl can be loaded and a communicator over the devices exists; otherwise (the same device listed
// twice, as in a one-GPU rehearsal; no librccl) peer copies on the first device's stream and an add
// kernel.  librccl is loaded on demand so that a one-GPU user never maps it.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace rt {

// One host thread per further device of a context over several GPUs, kept for the life of the context: a frame
// posts one job to each (the per-device calls block) and waits for all of them.
class DeviceWorker {
public:
    DeviceWorker() : th_([this] { run(); }) {}
    ~DeviceWorker() {
        { std::lock_guard<std::mutex> hold(m_); quit_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    DeviceWorker(const DeviceWorker&) = delete;
    DeviceWorker& operator=(const DeviceWorker&) = delete;
    void post(std::function<int()> job) {
        { std::lock_guard<std::mutex> hold(m_); job_ = std::move(job); pending_ = true; done_ = false; }
        cv_.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> hold(m_);
        cv_.wait(hold, [this] { return done_; });
        return rc_;
    }

private:
    void run() {
        for (;;) {
            std::function<int()> job;
            {
                std::unique_lock<std::mutex> hold(m_);
                cv_.wait(hold, [this] { return pending_ || quit_; });
                if (quit_ && !pending_) return;
                job = std::move(job_);
                pending_ = false;
            }
            int rc;
            try { rc = job(); } catch (...) { rc = -1000; }      // (reported as an internal error by the caller)
            { std::lock_guard<std::mutex> hold(m_); rc_ = rc; done_ = true; }
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int()> job_;
    bool pending_ = false, quit_ = false, done_ = true;
    int rc_ = 0;
    std::thread th_;      // (last: starts when everything above exists)
};

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;       // optional: used to drop communicators after a failed call
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;

    bool load(std::string& why) {
        if (handle) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!handle) { why = std::string("librccl not loadable: ") + dlerror(); return false; }
        bool ok = true;
        auto sym = [&](const char* n) { void* p = dlsym(handle, n); if (!p) { ok = false; why = std::string("librccl lacks ") + n; } return p; };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommAbort = (decltype(CommAbort))dlsym(handle, "ncclCommAbort");
        CommCount = (decltype(CommCount))sym("ncclCommCount");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        Reduce = (decltype(Reduce))sym("ncclReduce");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(handle); handle = nullptr; }
        return ok;
    }
};

inline RcclApi& rccl() {
    static RcclApi api;
    return api;
}

// ---- tile shards as contiguous pieces ------------------------------------------------------

// Pixel (x, y) of the n-th tile owned by shard `rank` (tiles t = rank, rank + count, ...), `within` in
// [0, 1024): row-major inside the 32x32 tile.
__device__ __forceinline__ bool shardPixel(int width, int height, int rank, int count, uint32_t n, uint32_t within, int& x, int& y) {
    const int tilesX = (width + 31) / 32;
    const int tile = rank + (int)n * count;
    x = (tile % tilesX) * 32 + (int)(within & 31u);
    y = (tile / tilesX) * 32 + (int)(within >> 5);
    return x < width && y < height;
}

// frame -> piece: the pixels of this shard's tiles, tile after tile.  T is one pixel's payload.
template <typename T>
__global__ void kPackShard(const T* frame, T* piece, int width, int height, int rank, int count, uint32_t nTiles) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nTiles * 1024u) return;
    int x, y;
    if (shardPixel(width, height, rank, count, i >> 10, i & 1023u, x, y)) piece[i] = frame[(size_t)y * width + x];
}

// piece -> frame, only inside the rendered rectangle (pixels outside it keep what the frame holds)
template <typename T>
__global__ void kUnpackShard(const T* piece, T* frame, int width, int height, int rank, int count, uint32_t nTiles, int rx, int ry, int rw, int rh) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nTiles * 1024u) return;
    int x, y;
    if (shardPixel(width, height, rank, count, i >> 10, i & 1023u, x, y) && x >= rx && y >= ry && x < rx + rw && y < ry + rh)
        frame[(size_t)y * width + x] = piece[i];
}

// sum buffers: dst += src (rgb sums and the sample count in w), then optionally the mean
__global__ void kAddSums(float4* dst, const float4* src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = dst[i];
    const float4 b = src[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    dst[i] = a;
}
__global__ void kSumsToMean(float4* buf, size_t n, int width, int rx, int ry, int rw, int rh) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % (size_t)width), y = (int)(i / (size_t)width);
    if (x < rx || y < ry || x >= rx + rw || y >= ry + rh) return;
    float4 a = buf[i];
    if (a.w > 0.0f) buf[i] = make_float4(a.x / a.w, a.y / a.w, a.z / a.w, 1.0f);
}

inline uint32_t shardTileCount(int width, int height, int rank, int count) {
    const int tiles = ((width + 31) / 32) * ((height + 31) / 32);
    return rank < tiles ? (uint32_t)((tiles - rank + count - 1) / count) : 0u;
}

}  // namespace rt

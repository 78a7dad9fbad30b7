// Streaming-read ceilings of the memory levels a BVH lives in: every workgroup reads a working set of
// a given size over and over with 16-byte loads, so the set settles in the smallest level that holds
// it — 0.2 MB (the bench scene: L2 of every XCD), 84 MB (the 1M-triangle scene: Infinity Cache),
// 4 GB (HBM).  Two access shapes: "stream" (consecutive lanes read consecutive 16-byte pieces: the
// best the level can do) and "gather" (every lane reads seven 16-byte pieces of its own random
// 128-byte line: the shape of an inner-node fetch of the traversal kernel).
// Prints one JSON object.  Build: hipcc --offload-arch=gfx950 -O3 tools/l2_stream.hip -o tools/bin/l2_stream
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every thread reads `count` pieces, a whole grid apart, wrapping around the working set (so a set
// smaller than the grid is read by many workgroups at once)
__global__ __launch_bounds__(256) void kStream(const float4* __restrict__ src, size_t n16, int count, float4* sink) {
    float4 acc = make_float4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) % n16;
    const size_t step = stride % n16;
    for (int k = 0; k < count; k++) {
        const float4 v = src[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        i += step;
        if (i >= n16) i -= n16;
    }
    if (acc.x == 12345.678f) sink[0] = acc;      // never true: keeps the loads alive
}

// every lane walks its own pseudo-random sequence of 128-byte lines and reads 7 x 16 bytes of each
__global__ __launch_bounds__(256) void kGather(const float4* __restrict__ src, uint32_t nLines, int steps, float4* sink) {
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int k = 0; k < steps; k++) {
        s = s * 747796405u + 2891336453u;
        const uint32_t line = (uint32_t)(((uint64_t)(s >> 4) * nLines) >> 28);
        const float4* p = src + (size_t)line * 8;
#pragma unroll
        for (int c = 0; c < 7; c++) { const float4 v = p[c]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    if (acc.x == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
    const double sizesMB[3] = {0.2, 84.0, 4096.0};
    const char* names[3] = {"l2_0.2MB", "mall_84MB", "hbm_4GB"};
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float4* sink = nullptr;
    CHECK(hipMalloc((void**)&sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d", prop.name, cus, prop.clockRate / 1000);
    for (int k = 0; k < 3; k++) {
        const size_t bytes = ((size_t)(sizesMB[k] * 1024.0 * 1024.0) / 4096) * 4096;
        float4* buf = nullptr;
        CHECK(hipMalloc((void**)&buf, bytes));
        CHECK(hipMemset(buf, 0, bytes));
        const size_t n16 = bytes / 16;
        const int grid = cus * 8;
        const int rounds = (int)(400e9 / ((double)grid * 256.0 * 16.0));     // pieces per thread: 400 GB per timed launch
        double best = 0.0;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kStream, dim3(grid), dim3(256), 0, 0, buf, n16, rounds, sink);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double gbs = (double)grid * 256.0 * 16.0 * rounds / (ms * 1e-3) / 1e9;
            if (rep > 0 && gbs > best) best = gbs;
        }
        const uint32_t nLines = (uint32_t)(bytes / 128);
        const int steps = 2048;
        double bestG = 0.0;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kGather, dim3(grid), dim3(256), 0, 0, buf, nLines, steps, sink);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double gbs = (double)grid * 256.0 * steps * 112.0 / (ms * 1e-3) / 1e9;
            if (rep > 0 && gbs > bestG) bestG = gbs;
        }
        printf(", \"%s\": {\"bytes\": %zu, \"stream_GBps\": %.1f, \"gather_node_GBps\": %.1f, \"gather_nodes_per_s\": %.4g}", names[k], bytes, best, bestG,
               bestG * 1e9 / 112.0);
        CHECK(hipFree(buf));
    }
    printf("}\n");
    return 0;
}

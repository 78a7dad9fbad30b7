// GPU-side scene layout of the production path: record types shared by the traversal kernel
// and the device-side tree builder (see kernel_fast.h for what each record holds).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace rt {

struct alignas(128) WNode {
    float4 lx, hx, ly, hy, lz, hz;
    int4 child;
    int4 pad;
};
struct alignas(64) FTri { float4 a, b, c, d; };
struct alignas(16) FNrm { float4 a, b, c; };                            // n0.xyz n1.x | n1.yz n2.xy | n2.z
struct alignas(32) FInst { float4 r0, r1, r2; float4 bxy; float blz, bhz; int32_t root, object; float4 pad; };   // TLAS slot box as (lo, hi) pairs: bxy = lo.x, hi.x, lo.y, hi.y;
                                                                                                                   // object | kDemotedBit; pad.x = bits of the BLAS index
struct alignas(32) FShade { float4 r0, r1, r2; float4 albedo, specular, emission; };

// child / stack words: >= 0 node index; < 0: ~payload, payload bit 30 set = instance rank,
// clear = (first triangle << 2 | count-1).  kEmptyChild never reaches the stack.
constexpr int32_t kEmptyChild = (int32_t)0x80000001;
// Inside a kernel a node word >= 0 with this bit set names a slot of the workgroup's LDS copy of the hot nodes
// instead of an index into the node array: the child words of the copied nodes are rewritten that way when the copy
// is made, so that a node step tells where its node lives from the word itself (node indices stay below 2^29).
constexpr int32_t kHotNodeBit = 1 << 29;
constexpr uint32_t kInstanceBit = 0x40000000u;
// FInst::object bit: the instance's mesh is one on which the distance culling of the production kernels was seen
// to come close to its slack (see RT_GUARD_FACTOR): every ray that enters the instance is handed to the
// reference-order kernel.
constexpr int32_t kDemotedBit = 1 << 30;

struct FastScene {
    const WNode* nodes;
    const FInst* inst;
    const FShade* shade;
    const FTri* tris;
    const FNrm* nrms;            // by triangle rank
    const int32_t* triOrig;      // by triangle rank: the caller's triangle index
    const int32_t* superTiles;   // the 32x32 pixel tiles this launch owns: tile column | tile row << 16
    int32_t nSuperTiles;
    int32_t tlasRoot;            // index of the TLAS root in nodes
    // nodes kept in LDS by every workgroup: [0, hotTlas) and [blasBase, blasBase + hotBlas)
    int32_t hotTlas, blasBase, hotBlas;
    // Every box bound of the caller's trees, sorted, per tree and axis (REF mode and the primary-id pass settle rays
    // with a zero direction component against it, see planesClear in kernel_fast.h): planeRange[(tree * 3 + axis) * 2]
    // = first value and count in `planes`; tree = BLAS index, or planeTlas for the top-level tree; count < 0: no table
    const float* planes;
    const int32_t* planeRange;
    int32_t planeTlas;
    int32_t nNodes, nInst, nTriRecords, nTriangles;      // entries in use of nodes / inst (and shade) / tris / nrms (and triOrig): what the bounds-checked kernel variant compares with
    int32_t* stackSpill;         // overflow of the LDS traversal stacks: [spillMax][threads of the grid]
    int32_t spillMax;            // overflow entries per lane (0: none, a deeper stack hands its pixel over at once)
};

template <typename T>
struct DBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t ensure(size_t count) {
        if (p && count <= n) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        hipError_t e = hipMalloc((void**)&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    hipError_t put(const std::vector<T>& v, hipStream_t s) {
        hipError_t e = ensure(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace rt

// Device-side construction of the production kernel's trees (interface: tlas_build.h): the top-level tree over the
// instances, per frame, and — where the caller asks for it — the bottom-level tree of a mesh at upload.  The kernels
// are shared; what a leaf is (one instance / one triangle, or two that are siblings in the radix tree) and what a
// finished node looks like (WNode / WideTree::Node) are policies (InstLeaves, TriLeaves; WNodeOut, WideOut).
//
// Steps, all on the caller's stream:
//   kTlasInit       bounds of the centroids := empty, parent of the root := none
//   kTlasBounds     min / max of the box centroids (wave reduction + ordered-integer atomics)
//   kTlasMorton     63-bit Morton code of each centroid (21 bits per axis), value = instance position
//   radix sort      rocPRIM, 64-bit keys with 32-bit values (stable, so the result is deterministic)
//   kTlasHierarchy  binary radix tree over the sorted codes (Karras, "Maximizing parallelism in the
//                   construction of BVHs, octrees and k-d trees", 2012); equal codes are told apart
//                   by their sorted position
//   kTlasRefit      boxes of the binary nodes, bottom up; the second thread to reach a node goes on
//   kTlasCollapse   one workgroup turns the binary tree into 4-wide nodes level by level (a node takes
//                   a binary node's two children and opens the inner child with the largest surface
//                   area until it has four) and numbers them breadth first with a block-wide scan
// HBM traffic is a few hundred bytes per instance; the steps are launch-latency bound below ~10^5
// instances.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "tlas_build.h"
#include "wide_bvh.h"

namespace rt {

namespace {

constexpr int kThreads = 256;
constexpr int kCollapseThreads = 1024;
constexpr float kInf = __builtin_inff();

__device__ __forceinline__ uint32_t orderedBits(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fromOrderedBits(uint32_t u) {
    return __uint_as_float((u >> 31) ? (u & 0x7fffffffu) : ~u);
}

struct Box3 { float lx, ly, lz, hx, hy, hz; };

__device__ __forceinline__ Box3 instBox(const FInst* inst, uint32_t r) {
    const float4 bxy = inst[r].bxy;
    Box3 b;
    b.lx = bxy.x; b.hx = bxy.y; b.ly = bxy.z; b.hy = bxy.w; b.lz = inst[r].blz; b.hz = inst[r].bhz;
    return b;
}
// a centroid coordinate; boxes that are not finite sort to the origin (their own box still decides hits)
__device__ __forceinline__ float centre(float lo, float hi) {
    const float c = (lo + hi) * 0.5f;
    return (c - c == 0.0f) ? c : 0.0f;
}
__device__ __forceinline__ Box3 unite(const Box3& a, const Box3& b) {
    Box3 r;
    r.lx = fminf(a.lx, b.lx); r.ly = fminf(a.ly, b.ly); r.lz = fminf(a.lz, b.lz);
    r.hx = fmaxf(a.hx, b.hx); r.hy = fmaxf(a.hy, b.hy); r.hz = fmaxf(a.hz, b.hz);
    return r;
}
__device__ __forceinline__ float halfArea(const Box3& b) {
    const float dx = b.hx - b.lx, dy = b.hy - b.ly, dz = b.hz - b.lz;
    if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
    const float a = dx * dy + dy * dz + dz * dx;
    return a == a ? a : 0.f;
}

// Leaves of the top-level tree: one instance each.  prim(i) = box of input item i; after the sort leaf k is the item
// at sorted position k.
struct InstLeaves {
    const FInst* inst;
    const uint32_t* vals;
    static constexpr int kStride = 1;
    __device__ __forceinline__ Box3 prim(uint32_t i) const { return instBox(inst, i); }
    __device__ __forceinline__ Box3 leaf(int k) const { return instBox(inst, vals[k]); }
    __device__ __forceinline__ int32_t code(int k) const { return ~(int32_t)(vals[k] | kInstanceBit); }
};
// Leaves of a bottom-level tree: the radix tree is built over single triangles, and a binary node whose two children
// are both triangles becomes a leaf of two (they are neighbours in Morton order and nothing in the tree is closer to
// either: one 128-byte line of the leaf array); a triangle whose sibling is a subtree stays a leaf of one.  Codes
// are those of WideTree (~(first position << 3 | count)).
struct TriLeaves {
    const BuildBox* boxes;
    const uint32_t* vals;
    int nPrims;
    static constexpr int kStride = 1;
    __device__ __forceinline__ Box3 prim(uint32_t i) const {
        const BuildBox bb = boxes[i];
        Box3 b;
        b.lx = bb.lo[0]; b.ly = bb.lo[1]; b.lz = bb.lo[2]; b.hx = bb.hi[0]; b.hy = bb.hi[1]; b.hz = bb.hi[2];
        return b;
    }
    __device__ __forceinline__ Box3 leaf(int k) const { return prim(vals[k]); }
    __device__ __forceinline__ int32_t code(int k) const { return ~(int32_t)(((uint32_t)k << 3) | 1u); }
    __device__ __forceinline__ int32_t pairCode(int k) const { return ~(int32_t)(((uint32_t)k << 3) | 2u); }
};
template <typename Leaves> struct MergesPairs { static constexpr bool value = false; };
template <> struct MergesPairs<TriLeaves> { static constexpr bool value = true; };

// Finished 4-wide nodes: the production kernel's own record, or the host builder's (whose trees the upload code
// lays out together with those built on the host).
struct WNodeOut {
    WNode* out;
    static constexpr int32_t kEmpty = kEmptyChild;
    __device__ __forceinline__ void write(int wi, const Box3* bx, const int32_t* code) const {
        WNode w;
        w.lx = make_float4(bx[0].lx, bx[1].lx, bx[2].lx, bx[3].lx); w.hx = make_float4(bx[0].hx, bx[1].hx, bx[2].hx, bx[3].hx);
        w.ly = make_float4(bx[0].ly, bx[1].ly, bx[2].ly, bx[3].ly); w.hy = make_float4(bx[0].hy, bx[1].hy, bx[2].hy, bx[3].hy);
        w.lz = make_float4(bx[0].lz, bx[1].lz, bx[2].lz, bx[3].lz); w.hz = make_float4(bx[0].hz, bx[1].hz, bx[2].hz, bx[3].hz);
        w.child = make_int4(code[0], code[1], code[2], code[3]);
        w.pad = make_int4(0, 0, 0, 0);
        out[wi] = w;
    }
};
struct WideOut {
    WideTree::Node* out;
    static constexpr int32_t kEmpty = widebvh::kEmpty;
    __device__ __forceinline__ void write(int wi, const Box3* bx, const int32_t* code) const {
        WideTree::Node w;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            w.lo[s][0] = bx[s].lx; w.lo[s][1] = bx[s].ly; w.lo[s][2] = bx[s].lz;
            w.hi[s][0] = bx[s].hx; w.hi[s][1] = bx[s].hy; w.hi[s][2] = bx[s].hz;
            w.child[s] = code[s];
        }
        out[wi] = w;
    }
};

__global__ void kTlasInit(uint32_t* bounds, int32_t* parentInner, int32_t* result) {
    const int t = threadIdx.x;
    if (t < 3) bounds[t] = 0xffffffffu;
    else if (t < 6) bounds[t] = 0u;
    if (t == 0) { parentInner[0] = -1; result[0] = 0; result[1] = 0; }
}

template <typename Leaves>
__global__ __launch_bounds__(kThreads) void kTlasBounds(Leaves lv, int n, uint32_t* bounds) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
    if (i < n) {
        const Box3 b = lv.prim((uint32_t)i);
        lo[0] = hi[0] = centre(b.lx, b.hx);
        lo[1] = hi[1] = centre(b.ly, b.hy);
        lo[2] = hi[2] = centre(b.lz, b.hz);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            atomicMin(&bounds[k], orderedBits(lo[k]));
            atomicMax(&bounds[3 + k], orderedBits(hi[k]));
        }
    }
}

__device__ __forceinline__ uint64_t spread21(uint64_t x) {
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
// all three axes share one scale (the largest extent), so that a flat scene is not cut along its thin axis
__device__ __forceinline__ uint32_t quantise(float c, float lo, float scale) {
    const float q = (c - lo) * scale;
    if (!(q > 0.f)) return 0u;
    return q >= 2097151.0f ? 2097151u : (uint32_t)q;
}

template <typename Leaves>
__global__ __launch_bounds__(kThreads) void kTlasMorton(Leaves lv, int n, const uint32_t* bounds, uint64_t* keys,
                                                        uint32_t* vals) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const Box3 b = lv.prim((uint32_t)i);
    const float lx = fromOrderedBits(bounds[0]), ly = fromOrderedBits(bounds[1]), lz = fromOrderedBits(bounds[2]);
    const float extent = fmaxf(fmaxf(fromOrderedBits(bounds[3]) - lx, fromOrderedBits(bounds[4]) - ly), fromOrderedBits(bounds[5]) - lz);
    const float scale = extent > 0.f ? 2097152.0f / extent : 0.f;
    const uint32_t qx = quantise(centre(b.lx, b.hx), lx, scale);
    const uint32_t qy = quantise(centre(b.ly, b.hy), ly, scale);
    const uint32_t qz = quantise(centre(b.lz, b.hz), lz, scale);
    keys[i] = (spread21(qx) << 2) | (spread21(qy) << 1) | spread21(qz);
    vals[i] = (uint32_t)i;
}

// length of the common prefix of the codes of leaves i and j (-1 outside the array); leaf k has the code at sorted
// position k * stride (a leaf of several items: that of its first)
__device__ __forceinline__ int commonPrefix(const uint64_t* keys, int n, int stride, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[(size_t)i * stride], b = keys[(size_t)j * stride];
    if (a == b) return 64 + __clz((uint32_t)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// children: >= 0 binary inner node, < 0: ~(sorted position of a leaf)
__global__ __launch_bounds__(kThreads) void kTlasHierarchy(const uint64_t* keys, int n, int stride, int32_t* left, int32_t* right,
                                                           int32_t* parentInner, int32_t* parentLeaf) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n - 1) return;
    const int d = commonPrefix(keys, n, stride, i, i + 1) - commonPrefix(keys, n, stride, i, i - 1) >= 0 ? 1 : -1;
    const int dMin = commonPrefix(keys, n, stride, i, i - d);
    int lMax = 2;
    while (commonPrefix(keys, n, stride, i, i + lMax * d) > dMin) lMax <<= 1;
    int l = 0;
    for (int t = lMax >> 1; t >= 1; t >>= 1)
        if (commonPrefix(keys, n, stride, i, i + (l + t) * d) > dMin) l += t;
    const int j = i + l * d;
    const int dNode = commonPrefix(keys, n, stride, i, j);
    int s = 0;
    for (int t = (l + 1) >> 1;; t = (t + 1) >> 1) {
        if (commonPrefix(keys, n, stride, i, i + (s + t) * d) > dNode) s += t;
        if (t <= 1) break;
    }
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    if (lo == gamma) { left[i] = ~gamma; parentLeaf[gamma] = i; }
    else { left[i] = gamma; parentInner[gamma] = i; }
    if (hi == gamma + 1) { right[i] = ~(gamma + 1); parentLeaf[gamma + 1] = i; }
    else { right[i] = gamma + 1; parentInner[gamma + 1] = i; }
}

// a box written by another workgroup: read past this CU's vector cache
__device__ __forceinline__ Box3 loadBoxCoherent(const float4* blo, const float4* bhi, int p) {
    float* lo = const_cast<float*>(reinterpret_cast<const float*>(blo + p));
    float* hi = const_cast<float*>(reinterpret_cast<const float*>(bhi + p));
    Box3 b;
    b.lx = __hip_atomic_load(lo + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.ly = __hip_atomic_load(lo + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.lz = __hip_atomic_load(lo + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.hx = __hip_atomic_load(hi + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.hy = __hip_atomic_load(hi + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b.hz = __hip_atomic_load(hi + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return b;
}

template <typename Leaves>
__global__ __launch_bounds__(kThreads) void kTlasRefit(Leaves lv, int n, const int32_t* left,
                                                       const int32_t* right, const int32_t* parentInner,
                                                       const int32_t* parentLeaf, float4* blo, float4* bhi, int32_t* flags) {
    const int k = blockIdx.x * kThreads + threadIdx.x;
    if (k >= n) return;
    Box3 box = lv.leaf(k);
    int32_t me = ~k;
    int32_t p = parentLeaf[k];
    // every inner node is reached by exactly two threads; the walk ends at the root at the latest
    // (a radix tree over 63 + 32 key bits has fewer than 96 levels: the bound only guards the loop)
    for (int step = 0; p >= 0 && step < 128; step++) {
        __threadfence();
        if (atomicAdd(&flags[p], 1) == 0) return;      // the other child is not ready: its thread goes on
        __threadfence();
        const int32_t other = left[p] == me ? right[p] : left[p];
        const Box3 ob = other >= 0 ? loadBoxCoherent(blo, bhi, other) : lv.leaf(~other);
        box = unite(box, ob);
        blo[p] = make_float4(box.lx, box.ly, box.lz, 0.f);
        bhi[p] = make_float4(box.hx, box.hy, box.hz, 0.f);
        me = p;
        p = parentInner[p];
    }
}


template <typename Leaves>
__device__ __forceinline__ Box3 childBox(const Leaves& lv, const float4* blo, const float4* bhi, int32_t c) {
    if (c >= 0) {
        const float4 lo = blo[c], hi = bhi[c];
        Box3 b;
        b.lx = lo.x; b.ly = lo.y; b.lz = lo.z; b.hx = hi.x; b.hy = hi.y; b.hz = hi.z;
        return b;
    }
    return lv.leaf(~c);
}

// result[0] = number of 4-wide nodes, result[1] = number of levels
template <typename Leaves, typename Out>
__global__ __launch_bounds__(kCollapseThreads) void kTlasCollapse(Leaves lv, int n, const int32_t* left,
                                                                   const int32_t* right, const float4* blo, const float4* bhi,
                                                                   Out out, int reserve, int32_t* queueA, int32_t* queueB,
                                                                   int32_t* result) {
    __shared__ int32_t waveTotal[kCollapseThreads / 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // does child c get a 4-wide node of its own?  (a binary node, unless it is a pair of leaves that becomes one leaf)
    auto opens = [&](int32_t c) -> bool {
        if (c < 0) return false;
        if (MergesPairs<Leaves>::value) return left[c] >= 0 || right[c] >= 0;
        return true;
    };
    if (n == 1) {
        if (tid == 0) {
            Box3 bx[4];
            int32_t code[4];
            bx[0] = lv.leaf(0);
            code[0] = lv.code(0);
#pragma unroll
            for (int s = 1; s < 4; s++) {
                bx[s].lx = bx[s].ly = bx[s].lz = bx[s].hx = bx[s].hy = bx[s].hz = kInf;
                code[s] = Out::kEmpty;
            }
            out.write(0, bx, code);
            result[0] = 1; result[1] = 1;
        }
        return;
    }
    if (MergesPairs<Leaves>::value && n == 2) {
        if (tid == 0) {
            Box3 bx[4];
            int32_t code[4];
            bx[0] = unite(lv.leaf(0), lv.leaf(1));
            if constexpr (MergesPairs<Leaves>::value) code[0] = lv.pairCode(0); else code[0] = Out::kEmpty;
#pragma unroll
            for (int s = 1; s < 4; s++) {
                bx[s].lx = bx[s].ly = bx[s].lz = bx[s].hx = bx[s].hy = bx[s].hz = kInf;
                code[s] = Out::kEmpty;
            }
            out.write(0, bx, code);
            result[0] = 1; result[1] = 1;
        }
        return;
    }
    int32_t* cur = queueA;
    int32_t* nxt = queueB;
    if (tid == 0) cur[0] = 0;      // the root of the radix tree
    __syncthreads();
    int levelBase = 0, levelCount = 1, depth = 0;
    while (levelCount > 0) {
        // n - 1 binary nodes give at most n - 1 wide nodes; anything else is a broken tree
        if (levelBase + levelCount > reserve || depth >= 128) { levelBase = -1; break; }
        depth++;
        const int nextBase = levelBase + levelCount;
        int nextCount = 0;
        for (int chunk = 0; chunk < levelCount; chunk += kCollapseThreads) {
            const int idx = chunk + tid;
            const bool act = idx < levelCount;
            int32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            int ns = 0;
            if (act) {
                const int32_t b = cur[idx];
                s0 = left[b]; s1 = right[b]; ns = 2;
                for (int iter = 0; iter < 2; iter++) {
                    int pick = -1;
                    float bestA = -1.f;
                    if (opens(s0)) { const float a = halfArea(childBox(lv, blo, bhi, s0)); if (a > bestA) { bestA = a; pick = 0; } }
                    if (opens(s1)) { const float a = halfArea(childBox(lv, blo, bhi, s1)); if (a > bestA) { bestA = a; pick = 1; } }
                    if (ns > 2 && opens(s2)) { const float a = halfArea(childBox(lv, blo, bhi, s2)); if (a > bestA) { bestA = a; pick = 2; } }
                    if (pick < 0) break;
                    const int32_t c = pick == 0 ? s0 : (pick == 1 ? s1 : s2);
                    const int32_t l = left[c], r = right[c];
                    if (pick == 0) s0 = l; else if (pick == 1) s1 = l; else s2 = l;
                    if (ns == 2) s2 = r; else s3 = r;
                    ns++;
                }
            }
            const int k = act ? (opens(s0) + opens(s1) + (ns > 2 && opens(s2)) + (ns > 3 && opens(s3))) : 0;
            // exclusive scan of k over the workgroup
            int incl = k;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off, 64);
                if (lane >= off) incl += v;
            }
            if (lane == 63) waveTotal[wave] = incl;
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < kCollapseThreads / 64; w++) {
                const int t = waveTotal[w];
                if (w < wave) before += t;
                total += t;
            }
            int pos = nextCount + before + incl - k;
            if (act) {
                const int wi = levelBase + idx;
                int32_t code[4];
                Box3 bx[4];
                const int32_t sl[4] = {s0, s1, s2, s3};
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    if (s < ns) {
                        bx[s] = childBox(lv, blo, bhi, sl[s]);
                        if (opens(sl[s])) { if (pos < n) nxt[pos] = sl[s]; code[s] = nextBase + pos; pos++; }
                        else if (sl[s] >= 0) { if constexpr (MergesPairs<Leaves>::value) code[s] = lv.pairCode(~left[sl[s]]); else code[s] = Out::kEmpty; }
                        else code[s] = lv.code(~sl[s]);
                    } else {
                        bx[s].lx = bx[s].ly = bx[s].lz = bx[s].hx = bx[s].hy = bx[s].hz = kInf;    // never entered
                        code[s] = Out::kEmpty;
                    }
                }
                if (wi < reserve) out.write(wi, bx, code);
            }
            nextCount += total;
            __syncthreads();     // waveTotal is reused; nxt is read by the whole workgroup in the next level
        }
        int32_t* t = cur; cur = nxt; nxt = t;
        levelBase = nextBase;
        levelCount = nextCount;
    }
    if (tid == 0) { result[0] = levelBase; result[1] = depth; }
}

// Per-instance records of the production kernel from the caller's arrays as uploaded: instance r
// sits in leaf slot (slotOf[r] & 3) of TLAS node (slotOf[r] >> 2).
__global__ __launch_bounds__(kThreads) void kTlasPack(const rtamd_node* tlas, const rtamd_object_property* objects, const int32_t* slotOf,
                                                      const int32_t* blasRoot, const int32_t* blasDemoted, int n, FInst* inst, FShade* shade) {
    const int r = blockIdx.x * kThreads + threadIdx.x;
    if (r >= n) return;
    const int32_t code = slotOf[r];
    const rtamd_node& nd = tlas[code >> 2];
    const int i = code & 3;
    const int32_t d = nd.Data[i];
    const rtamd_object_property& o = objects[d];
    FInst fi;
    const float* m = o.WorldToLocal;
    fi.r0 = make_float4(m[0], m[4], m[8], m[12]);
    fi.r1 = make_float4(m[1], m[5], m[9], m[13]);
    fi.r2 = make_float4(m[2], m[6], m[10], m[14]);
    fi.bxy = make_float4(nd.Lx[i], nd.Hx[i], nd.Ly[i], nd.Hy[i]);
    fi.blz = nd.Lz[i]; fi.bhz = nd.Hz[i];
    fi.root = blasRoot[o.BLASIndex];
    fi.object = d | (blasDemoted && blasDemoted[o.BLASIndex] ? kDemotedBit : 0);      // (a mesh the culling guard marked: fast_layout.h)
    fi.pad = make_float4(__int_as_float(o.BLASIndex), 0.f, 0.f, 0.f);
    inst[r] = fi;
    FShade fs;
    const float* l = o.LocalToWorld;
    fs.r0 = make_float4(l[0], l[4], l[8], l[12]);
    fs.r1 = make_float4(l[1], l[5], l[9], l[13]);
    fs.r2 = make_float4(l[2], l[6], l[10], l[14]);
    fs.albedo = make_float4(o.albedo[0], o.albedo[1], o.albedo[2], o.albedo[3]);
    fs.specular = make_float4(o.specular[0], o.specular[1], o.specular[2], o.specular[3]);
    fs.emission = make_float4(o.emission[0], o.emission[1], o.emission[2], o.emission[3]);
    shade[r] = fs;
}

// Leaf records of the bottom-level trees: record j of the output is the record of the triangle with rank srcOf[j]
// (a blank that no ray enters when srcOf[j] < 0), and every pair of records (2p, 2p + 1) is stored component by
// component — lo.xyz, hi.xyz, v0.xyz, e1.xyz, e2.xyz, rank of record 2p beside those of record 2p + 1 — so that the
// traversal kernels test both with two-wide instructions.  One thread per output float.
__global__ __launch_bounds__(kThreads) void kGatherLeafRecords(const float* ranked, const int32_t* srcOf, size_t nFloats, float* out) {
    const size_t idx = (size_t)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= nFloats) return;
    const size_t pair = idx >> 5;
    const int within = (int)(idx & 31u), j = within >> 1, k = within & 1;
    const int component = j < 6 ? j : (j < 15 ? j + 1 : 6);      // (the rank, float 6 of a record, goes last)
    const int32_t src = srcOf[2 * pair + (size_t)k];
    float v;
    if (src >= 0) v = ranked[(size_t)src * 16 + (size_t)component];
    else v = component < 6 ? kInf : 0.f;
    out[idx] = v;
}

template <typename T>
struct Buf {
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
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

struct TlasDeviceBuilder {
    Buf<uint64_t> keysA, keysB;
    Buf<uint32_t> valsA, valsB, bounds;
    Buf<int32_t> left, right, parentInner, parentLeaf, flags, queueA, queueB, result;
    Buf<float4> blo, bhi;
    Buf<unsigned char> temp, prims, wide;      // sort scratch; a mesh's boxes and its finished nodes (bottom-level builds)
};

TlasDeviceBuilder* tlasDeviceCreate() { return new (std::nothrow) TlasDeviceBuilder(); }

void tlasDeviceDestroy(TlasDeviceBuilder* b) {
    if (!b) return;
    b->keysA.release(); b->keysB.release(); b->valsA.release(); b->valsB.release(); b->bounds.release();
    b->left.release(); b->right.release(); b->parentInner.release(); b->parentLeaf.release(); b->flags.release();
    b->queueA.release(); b->queueB.release(); b->result.release(); b->blo.release(); b->bhi.release(); b->temp.release(); b->prims.release(); b->wide.release();
    delete b;
}

hipError_t tlasPackInstances(const rtamd_node* tlas, const rtamd_object_property* objects, const int32_t* slotOf,
                             const int32_t* blasRoot, const int32_t* blasDemoted, int n, FInst* inst, FShade* shade, hipStream_t stream) {
    if (!tlas || !objects || !slotOf || !blasRoot || !inst || !shade || n < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kTlasPack, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, tlas, objects, slotOf,
                       blasRoot, blasDemoted, n, inst, shade);
    return hipGetLastError();
}

namespace {

// The shared sequence.  nPrims items, nLeaves = ceil(nPrims / Leaves::kStride) leaves; `pre` reads the items by input
// index (no sorted order yet), `post(vals)` makes the policy that knows the sorted order.  On return the stream has
// been synchronised.
template <typename Leaves, typename Out, typename MakeSorted>
hipError_t buildOnDevice(TlasDeviceBuilder* b, const Leaves& pre, MakeSorted post, int nPrims, Out out, int reserve,
                         hipStream_t stream, int* nNodes, int* depth) {
    const size_t N = (size_t)nPrims;
    const int nLeaves = (nPrims + Leaves::kStride - 1) / Leaves::kStride;
    hipError_t e;
    if ((e = b->keysA.ensure(N)) != hipSuccess || (e = b->keysB.ensure(N)) != hipSuccess || (e = b->valsA.ensure(N)) != hipSuccess ||
        (e = b->valsB.ensure(N)) != hipSuccess || (e = b->bounds.ensure(8)) != hipSuccess || (e = b->left.ensure(N)) != hipSuccess ||
        (e = b->right.ensure(N)) != hipSuccess || (e = b->parentInner.ensure(N)) != hipSuccess || (e = b->parentLeaf.ensure(N)) != hipSuccess ||
        (e = b->flags.ensure(N)) != hipSuccess || (e = b->queueA.ensure(N)) != hipSuccess || (e = b->queueB.ensure(N)) != hipSuccess ||
        (e = b->result.ensure(4)) != hipSuccess || (e = b->blo.ensure(N)) != hipSuccess || (e = b->bhi.ensure(N)) != hipSuccess)
        return e;
    const unsigned grid = (unsigned)((N + kThreads - 1) / kThreads);
    const unsigned leafGrid = (unsigned)(((size_t)nLeaves + kThreads - 1) / kThreads);
    hipLaunchKernelGGL(kTlasInit, dim3(1), dim3(64), 0, stream, b->bounds.p, b->parentInner.p, b->result.p);
    const uint32_t* vals = b->valsA.p;
    if (nPrims > 1) {
        if ((e = hipMemsetAsync(b->flags.p, 0, N * sizeof(int32_t), stream)) != hipSuccess) return e;
        hipLaunchKernelGGL((kTlasBounds<Leaves>), dim3(grid), dim3(kThreads), 0, stream, pre, nPrims, b->bounds.p);
        hipLaunchKernelGGL((kTlasMorton<Leaves>), dim3(grid), dim3(kThreads), 0, stream, pre, nPrims, b->bounds.p, b->keysA.p, b->valsA.p);
        size_t tempBytes = 0;
        if ((e = rocprim::radix_sort_pairs(nullptr, tempBytes, b->keysA.p, b->keysB.p, b->valsA.p, b->valsB.p, (unsigned)nPrims, 0u, 63u, stream)) != hipSuccess) return e;
        if ((e = b->temp.ensure(tempBytes)) != hipSuccess) return e;
        if ((e = rocprim::radix_sort_pairs(b->temp.p, tempBytes, b->keysA.p, b->keysB.p, b->valsA.p, b->valsB.p, (unsigned)nPrims, 0u, 63u, stream)) != hipSuccess) return e;
        vals = b->valsB.p;
    } else {
        const uint32_t zero = 0;
        if ((e = hipMemcpyAsync(b->valsA.p, &zero, sizeof zero, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    }
    const Leaves lv = post(vals);
    if (nLeaves > 1) {
        hipLaunchKernelGGL(kTlasHierarchy, dim3(leafGrid), dim3(kThreads), 0, stream, b->keysB.p, nLeaves, Leaves::kStride, b->left.p, b->right.p, b->parentInner.p, b->parentLeaf.p);
        hipLaunchKernelGGL((kTlasRefit<Leaves>), dim3(leafGrid), dim3(kThreads), 0, stream, lv, nLeaves, b->left.p, b->right.p, b->parentInner.p, b->parentLeaf.p,
                           b->blo.p, b->bhi.p, b->flags.p);
    }
    hipLaunchKernelGGL((kTlasCollapse<Leaves, Out>), dim3(1), dim3(kCollapseThreads), 0, stream, lv, nLeaves, b->left.p, b->right.p, b->blo.p, b->bhi.p,
                       out, reserve, b->queueA.p, b->queueB.p, b->result.p);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    int32_t res[2] = {0, 0};
    if ((e = hipMemcpyAsync(res, b->result.p, sizeof(res), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    if (res[0] < 1) return hipErrorUnknown;
    *nNodes = res[0];
    *depth = res[1];
    return hipSuccess;
}

}  // namespace

hipError_t tlasDeviceBuild(TlasDeviceBuilder* b, const FInst* inst, int n, WNode* nodesOut, int reserve, hipStream_t stream,
                           int* nNodes, int* depth) {
    if (!b || !inst || !nodesOut || n < 1 || reserve < (n > 2 ? n - 1 : 1)) return hipErrorInvalidValue;
    const InstLeaves pre{inst, nullptr};
    return buildOnDevice(b, pre, [inst](const uint32_t* vals) { return InstLeaves{inst, vals}; }, n, WNodeOut{nodesOut}, reserve, stream, nNodes, depth);
}

hipError_t blasDeviceBuild(TlasDeviceBuilder* b, const BuildBox* boxes, int n, WideTree& out, hipStream_t stream) {
    out.nodes.clear(); out.order.clear(); out.depth = 0;
    if (!b || !boxes || n < 1) return hipErrorInvalidValue;
    const int nLeaves = n;
    const int reserve = nLeaves > 2 ? nLeaves - 1 : 1;
    hipError_t e;
    if ((e = b->prims.ensure((size_t)n * sizeof(BuildBox))) != hipSuccess || (e = b->wide.ensure((size_t)reserve * sizeof(WideTree::Node))) != hipSuccess) return e;
    BuildBox* dBoxes = reinterpret_cast<BuildBox*>(b->prims.p);
    WideTree::Node* dNodes = reinterpret_cast<WideTree::Node*>(b->wide.p);
    if ((e = hipMemcpyAsync(dBoxes, boxes, (size_t)n * sizeof(BuildBox), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    int nNodes = 0, depth = 0;
    const TriLeaves pre{dBoxes, nullptr, n};
    if ((e = buildOnDevice(b, pre, [dBoxes, n](const uint32_t* vals) { return TriLeaves{dBoxes, vals, n}; }, n, WideOut{dNodes}, reserve, stream, &nNodes, &depth)) != hipSuccess)
        return e;
    try {
        out.nodes.resize((size_t)nNodes);
        out.order.resize((size_t)n);
    } catch (...) { out.nodes.clear(); out.order.clear(); return hipErrorOutOfMemory; }
    if ((e = hipMemcpyAsync(out.nodes.data(), dNodes, (size_t)nNodes * sizeof(WideTree::Node), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    const uint32_t* sorted = n > 1 ? b->valsB.p : b->valsA.p;
    if ((e = hipMemcpyAsync(out.order.data(), sorted, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    out.depth = depth;
    return hipStreamSynchronize(stream);
}

hipError_t gatherLeafRecords(TlasDeviceBuilder* b, const FTri* ranked, size_t nRanked, const int32_t* srcOf, size_t nRecords, FTri* out,
                             hipStream_t stream) {
    if (!b || !ranked || !srcOf || !out || (nRecords & 1u)) return hipErrorInvalidValue;
    if (nRecords == 0) return hipSuccess;
    hipError_t e;
    if ((e = b->prims.ensure(nRanked * sizeof(FTri))) != hipSuccess || (e = b->queueA.ensure(nRecords)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(b->prims.p, ranked, nRanked * sizeof(FTri), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(b->queueA.p, srcOf, nRecords * sizeof(int32_t), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    const size_t nFloats = nRecords * 16;
    hipLaunchKernelGGL(kGatherLeafRecords, dim3((unsigned)((nFloats + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream,
                       reinterpret_cast<const float*>(b->prims.p), b->queueA.p, nFloats, reinterpret_cast<float*>(out));
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return hipStreamSynchronize(stream);      // (the host arrays may go once this returns)
}

hipError_t sortFloatsOnDevice(TlasDeviceBuilder* b, float* values, size_t n, hipStream_t stream) {
    if (!b || !values || n >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    if (n < 2) return hipSuccess;
    hipError_t e;
    if ((e = b->keysA.ensure((n + 1) / 2)) != hipSuccess || (e = b->keysB.ensure((n + 1) / 2)) != hipSuccess) return e;
    float* in = reinterpret_cast<float*>(b->keysA.p);
    float* out = reinterpret_cast<float*>(b->keysB.p);
    if ((e = hipMemcpyAsync(in, values, n * sizeof(float), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    size_t tempBytes = 0;
    if ((e = rocprim::radix_sort_keys(nullptr, tempBytes, in, out, (unsigned)n, 0u, 32u, stream)) != hipSuccess) return e;
    if ((e = b->temp.ensure(tempBytes)) != hipSuccess) return e;
    if ((e = rocprim::radix_sort_keys(b->temp.p, tempBytes, in, out, (unsigned)n, 0u, 32u, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(values, out, n * sizeof(float), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    return hipStreamSynchronize(stream);
}

}  // namespace rt

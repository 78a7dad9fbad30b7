// Reference-order kernel: traverses the reference's own node arrays (116-byte AoS
// nodes, index -> int3 -> Vertex indirection) in exactly the order of
// Assets/Shaders/RayTracingShader.compute:277-384 — children pushed slot 0..3, popped
// last-in-first-out, leaf slots tested 0..3, a hit accepted only if strictly closer.
// One thread per pixel.  It is the in-library definition of "what the reference
// computes" on the GPU: it yields the work counters (nodes / triangles / instances /
// accepted hits per ray in the reference's order) and is the cross-check for the
// production kernel.  It is not the fast path.
#pragma once

#include "kernel_common.h"

namespace rt {

struct HitRec {
    float t, u, v;
    int32_t object, triangle;
    bool back;           // hit from behind (only rays inside a dielectric test both sides)
};

struct LocalCounters {
    uint32_t rays, nodes, tris, instances, hits, maxBlas, maxTlas, overflow;
};

constexpr int kRefBlasStack = 64 * 4;   // MAX_BLAS_STACK_INT4 = 64, .compute:93
constexpr int kRefTlasStack = 8 * 4;    // MAX_TLAS_STACK_INT4 = 8,  .compute:94

// Raycast, .compute:233-263, for child i of a node in the reference layout.
__device__ __forceinline__ bool refSlab(const rtamd_node& n, int i, const Ray& ray, float maxFraction) {
    float lx = n.Lx[i] - ray.origin.x, hx = n.Hx[i] - ray.origin.x;
    float nearXt = lx * ray.invDir.x, farXt = hx * ray.invDir.x;
    float ly = n.Ly[i] - ray.origin.y, hy = n.Hy[i] - ray.origin.y;
    float nearYt = ly * ray.invDir.y, farYt = hy * ray.invDir.y;
    float lz = n.Lz[i] - ray.origin.z, hz = n.Hz[i] - ray.origin.z;
    float nearZt = lz * ray.invDir.z, farZt = hz * ray.invDir.z;
    float nearX = fmin_(nearXt, farXt), farX = fmax_(nearXt, farXt);
    float nearY = fmin_(nearYt, farYt), farY = fmax_(nearYt, farYt);
    float nearZ = fmin_(nearZt, farZt), farZ = fmax_(nearZt, farZt);
    float nearMax = fmax_(fmax_(fmax_(nearX, nearY), nearZ), 0.0f);
    float farMin = fmin_(fmin_(fmin_(farX, farY), farZ), maxFraction);
    return nearMax <= farMin;
}

// Entry distance of the same test (its nearMax).
__device__ __forceinline__ float refSlabNear(const rtamd_node& n, int i, const Ray& ray) {
    float nearXt = (n.Lx[i] - ray.origin.x) * ray.invDir.x, farXt = (n.Hx[i] - ray.origin.x) * ray.invDir.x;
    float nearYt = (n.Ly[i] - ray.origin.y) * ray.invDir.y, farYt = (n.Hy[i] - ray.origin.y) * ray.invDir.y;
    float nearZt = (n.Lz[i] - ray.origin.z) * ray.invDir.z, farZt = (n.Hz[i] - ray.origin.z) * ray.invDir.z;
    return fmax_(fmax_(fmax_(fmin_(nearXt, farXt), fmin_(nearYt, farYt)), fmin_(nearZt, farZt)), 0.0f);
}

// IntersectBLAS, .compute:277-346 (position / normal / material are derived after the
// traversal from the surviving (t,u,v,object,triangle); they are pure functions of it).
// The four-box mask is evaluated once per node with the distance at that time (:298);
// hits accepted while walking a leaf's slots do not re-filter its later slots.
// instNear: entry distance of the instance's TLAS slot box (for the guard of the production kernels' culling only).
__device__ __forceinline__ void refIntersectBlas(const RefScene& sc, const Ray& ray, int index, HitRec& best,
                                                 int32_t* stack, LocalCounters& cnt, bool twoSided, float instNear) {
    const rtamd_object_property& prop = sc.objects[index];
    int blasIndex = prop.BLASIndex;
    int blasOffset = sc.blasOffsets[blasIndex];
    int triangleOffset = sc.triOffsets[blasIndex];
    int vertexOffset = sc.vtxOffsets[blasIndex];
    Ray localRay = makeRay(mulPoint(prop.WorldToLocal, ray.origin), mulDir(prop.WorldToLocal, ray.direction));
    cnt.instances++;
    int top = 0;
    stack[top++] = 1;
    do {
        top--;
        int topIndex = stack[top];
        const rtamd_node& node = sc.blas[topIndex + blasOffset];
        cnt.nodes++;
        bool hitMask[4];
#pragma unroll
        for (int i = 0; i < 4; i++) hitMask[i] = refSlab(node, i, localRay, best.t);
        if (node.Flags == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int childNodeIndex = node.Data[i];
                if (childNodeIndex != 0 && hitMask[i]) {
                    if (top < kRefBlasStack) stack[top++] = childNodeIndex;
                    else cnt.overflow++;
                    if ((uint32_t)top > cnt.maxBlas) cnt.maxBlas = top;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                int triangleIndex = node.Data[i];
                if (triangleIndex != -1 && hitMask[i]) {
                    rtamd_int3 tri = sc.tris[triangleIndex + triangleOffset];
                    const rtamd_vertex& a = sc.verts[tri.x + vertexOffset];
                    const rtamd_vertex& b = sc.verts[tri.y + vertexOffset];
                    const rtamd_vertex& c = sc.verts[tri.z + vertexOffset];
                    V3 v0 = v3(a.position[0], a.position[1], a.position[2]);
                    V3 v1 = v3(b.position[0], b.position[1], b.position[2]);
                    V3 v2 = v3(c.position[0], c.position[1], c.position[2]);
                    cnt.tris++;
                    float t, u, v;
                    bool back;
                    if (trisect(localRay.origin, localRay.direction, v0, v1 - v0, v2 - v0, t, u, v, twoSided, back)) {
                        // not part of the reference: the guard of the production kernels' distance culling (RT_GUARD_FACTOR)
                        if (fmax_(refSlabNear(node, i, localRay), instNear) > t * RT_GUARD_FACTOR) {
                            if (sc.guardCount) atomicAdd(sc.guardCount, 1);
                            if (sc.guardBlas) atomicAdd(sc.guardBlas + blasIndex, 1);
                        }
                        if (t > 0 && t < best.t) {
                            cnt.hits++;
                            best.t = t; best.u = u; best.v = v;
                            best.object = index; best.triangle = triangleIndex; best.back = back;
                        }
                    }
                }
            }
        }
    } while (top > 0);
}

// IntersectTLAS, .compute:348-384.  As in the BLAS loop the mask is computed once per
// node (:360), before any BLAS visit of that node can shrink the distance.
__device__ __forceinline__ HitRec refIntersectTlas(const RefScene& sc, const Ray& ray, int32_t* blasStack,
                                                   LocalCounters& cnt, bool twoSided = false) {
    HitRec best;
    best.t = RT_INF; best.u = 0; best.v = 0; best.object = -1; best.triangle = -1; best.back = false;
    cnt.rays++;
    int32_t stack[kRefTlasStack];
    int top = 0;
    stack[top++] = 1;
    do {
        top--;
        int topIndex = stack[top];
        const rtamd_node& node = sc.tlas[topIndex];
        cnt.nodes++;
        bool hitMask[4];
#pragma unroll
        for (int i = 0; i < 4; i++) hitMask[i] = refSlab(node, i, ray, best.t);
        bool internal = node.Flags == 0;
        for (int i = 0; i < 4; i++) {
            if (!hitMask[i]) continue;
            int data = node.Data[i];
            if (internal) {
                if (data != 0) {
                    if (top < kRefTlasStack) stack[top++] = data;
                    else cnt.overflow++;
                    if ((uint32_t)top > cnt.maxTlas) cnt.maxTlas = top;
                }
            } else if (data != -1) {
                refIntersectBlas(sc, ray, data, best, blasStack, cnt, twoSided, refSlabNear(node, i, ray));
            }
        }
    } while (top > 0);
    return best;
}


// What IntersectBLAS stores on an accepted hit (.compute:329-339), evaluated once for
// the hit that survived.
__device__ __forceinline__ Surf refResolve(const RefScene& sc, const Ray& ray, const HitRec& h) {
    const rtamd_object_property& prop = sc.objects[h.object];
    int blasIndex = prop.BLASIndex;
    int triangleOffset = sc.triOffsets[blasIndex];
    int vertexOffset = sc.vtxOffsets[blasIndex];
    Ray localRay = makeRay(mulPoint(prop.WorldToLocal, ray.origin), mulDir(prop.WorldToLocal, ray.direction));
    rtamd_int3 tri = sc.tris[h.triangle + triangleOffset];
    const rtamd_vertex& a = sc.verts[tri.x + vertexOffset];
    const rtamd_vertex& b = sc.verts[tri.y + vertexOffset];
    const rtamd_vertex& c = sc.verts[tri.z + vertexOffset];
    Surf s;
    s.position = mulPoint(prop.LocalToWorld, localRay.origin + h.t * localRay.direction);
    float w = 1 - h.u - h.v;
    V3 n0 = v3(a.normal[0], a.normal[1], a.normal[2]);
    V3 n1 = v3(b.normal[0], b.normal[1], b.normal[2]);
    V3 n2 = v3(c.normal[0], c.normal[1], c.normal[2]);
    V3 nl = ((h.u * n1) + (h.v * n2)) + (w * n0);
    s.normal = normalize(mulDir(prop.LocalToWorld, nl));
    s.albedo = v3(prop.albedo[0], prop.albedo[1], prop.albedo[2]);
    s.specular = v3(prop.specular[0], prop.specular[1], prop.specular[2]);
    s.emission = v3(prop.emission[0], prop.emission[1], prop.emission[2]);
    s.albedoW = prop.albedo[3]; s.specularW = prop.specular[3];
    s.back = h.back;
    if (h.back) s.normal = v3(-s.normal.x, -s.normal.y, -s.normal.z);
    s.object = h.object; s.triangle = h.triangle;
    return s;
}

// One path of PT mode in reference order: sample `sample` (counted from 0 over the whole render) of pixel (x, y).
__device__ __forceinline__ V3 refTracePath(const RefScene& sc, const Sky& sky, const RenderArgs& a, int x, int y, int sample,
                                           int32_t* blasStack, LocalCounters& cnt) {
    const size_t pix = (size_t)y * a.width + x;
    uint32_t key = rngKey(a.seedHi, a.seedLo, (uint32_t)pix, (uint32_t)sample);
    float jx = a.cb.PixelOffset[0], jy = a.cb.PixelOffset[1];
    if (!(a.flags & RTAMD_FLAG_NO_JITTER)) { jx = rngDraw(key, 0u); jy = rngDraw(key, 1u); }
    Ray ray = cameraRay(a.cb, pixelToNdc(x, jx, a.width), pixelToNdc(y, jy, a.height));
    V3 L = v3(0, 0, 0), energy = v3(1, 1, 1);
    const bool dielectrics = (a.flags & RTAMD_FLAG_DIELECTRICS) != 0;
    bool insideGlass = false;
    for (int b = 0;; b++) {
        HitRec h = refIntersectTlas(sc, ray, blasStack, cnt, insideGlass);
        if (!(h.t < RT_INF)) { L = L + energy * shadeMiss(sky, ray.direction); break; }
        Surf sf = refResolve(sc, ray, h);
        L = L + energy * sf.emission;
        if (b >= a.maxBounces) break;
        V3 org, dir;
        const bool goOn = dielectrics ? scatterSurface<true>(sf, ray.direction, key, (uint32_t)b, energy, org, dir, insideGlass)
                                      : scatterSurface<false>(sf, ray.direction, key, (uint32_t)b, energy, org, dir, insideGlass);
        if (!goOn) break;
        ray = makeRay(org, dir);
    }
    return L;
}

__device__ __forceinline__ unsigned long long waveSum(uint32_t v) {
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}
__device__ __forceinline__ uint32_t waveMax(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

// One thread per pixel, 8x8 pixels per wave ([numthreads(8,8,1)], .compute:431).  Unlike
// the reference dispatch (RayTracingPass.cs:101 launches width x height GROUPS) the grid
// covers the image once.
// With a pixel list (the pixels the production kernel handed over) the grid is
// one-dimensional: in REF mode thread i renders pixel list[i]; in PT mode thread i traces
// ONE sample, i % spp, of pixel list[i / spp] into sampleOut[i], and kSumListedSamples adds
// a pixel's samples in sample order afterwards (the same order a single thread would use).
// A short list is spread out: only every (1 << listSpread)-th lane takes an item, so that the few
// paths there are do not wait for each other's branches inside a wave (the nested TLAS / BLAS loops
// of the reference order serialise lanes that are in different instances).
__global__ __launch_bounds__(64) void kFaithful(RefScene sc, Sky sky, RenderArgs a, const int32_t* list, int listCount,
                                                float4* sampleOut, int listSpread) {
    int x = blockIdx.x * 8 + (threadIdx.x & 7);
    int y = blockIdx.y * 8 + (threadIdx.x >> 3);
    bool inside;
    int sFirst = 0, sLast = a.spp;       // samples this thread traces
    bool doPrimary = true;
    int gi = 0;
    if (list) {
        const unsigned slot = blockIdx.x * 64u + threadIdx.x;
        gi = (int)(slot >> listSpread);
        const int per = a.mode == RTAMD_MODE_PT ? a.spp : 1;
        inside = (slot & ((1u << listSpread) - 1u)) == 0u && (long long)gi < (long long)listCount * per;
        int p = inside ? list[gi / per] : 0;
        x = p % a.width;
        y = p / a.width;
        if (a.mode == RTAMD_MODE_PT) { sFirst = gi % per; sLast = sFirst + 1; doPrimary = sFirst == 0; }
    } else {
        inside = x < a.width && y < a.height && pixelSelected(a, x, y);
    }
    LocalCounters cnt = {0, 0, 0, 0, 0, 0, 0, 0};
    int32_t blasStack[kRefBlasStack];
    if (inside) {
        size_t pix = (size_t)y * a.width + x;
        if (doPrimary && (a.mode == RTAMD_MODE_REF || a.outIds || a.outT)) {
            Ray cam = cameraRay(a.cb, pixelToNdc(x, a.cb.PixelOffset[0], a.width), pixelToNdc(y, a.cb.PixelOffset[1], a.height));
            LocalCounters side = {0, 0, 0, 0, 0, 0, 0, 0};
            LocalCounters& use = a.mode == RTAMD_MODE_REF ? cnt : side;
            HitRec h = refIntersectTlas(sc, cam, blasStack, use);
            cnt.overflow += side.overflow;
            if (a.outIds) { a.outIds[2 * pix] = h.object; a.outIds[2 * pix + 1] = h.triangle; }
            if (a.outT) a.outT[pix] = h.t;
            if (a.mode == RTAMD_MODE_REF) {
                // Trace, .compute:413-429; Shade takes the ray by value (:386), so a hit
                // contributes its emission and a miss the sky along the mirrored direction.
                float4 px = make_float4(0, 0, 0, 0);
                if (h.t < RT_INF) {
                    Surf s = refResolve(sc, cam, h);
                    Ray ray = makeRay(s.position, reflect(cam.direction, s.normal));
                    HitRec h2 = refIntersectTlas(sc, ray, blasStack, cnt);
                    V3 shade;
                    if (h2.t < RT_INF) shade = refResolve(sc, ray, h2).emission;
                    else shade = shadeMiss(sky, ray.direction);
                    V3 result = v3(0, 0, 0) + v3(1, 1, 1) * shade;
                    px = make_float4(result.x, result.y, result.z, 1.0f);
                }
                if (a.outRgba) a.outRgba[pix] = px;
            }
        }
        if (a.mode == RTAMD_MODE_PT) {
            V3 sum = v3(0, 0, 0);
            for (int s = sFirst; s < sLast; s++) sum = sum + refTracePath(sc, sky, a, x, y, a.sampleOffset + s, blasStack, cnt);
            if (sampleOut) sampleOut[gi] = make_float4(sum.x, sum.y, sum.z, 0.f);
            else writePathPixel(a, pix, sum);
        }
    }
    // per-wave reduction of the counters, one atomic per wave and counter
    unsigned long long rays = waveSum(cnt.rays);
    unsigned long long nodes = waveSum(cnt.nodes), tris = waveSum(cnt.tris), inst = waveSum(cnt.instances);
    unsigned long long hits = waveSum(cnt.hits), ovf = waveSum(cnt.overflow);
    uint32_t mb = waveMax(cnt.maxBlas), mt = waveMax(cnt.maxTlas);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(a.rayCount, rays);
        if (a.counters) {
            atomicAdd(&a.counters->rays, rays);
            atomicAdd(&a.counters->nodes, nodes);
            atomicAdd(&a.counters->tris, tris);
            atomicAdd(&a.counters->instances, inst);
            atomicAdd(&a.counters->hits, hits);
            atomicMax(&a.counters->maxBlasStack, (unsigned long long)mb);
            atomicMax(&a.counters->maxTlasStack, (unsigned long long)mt);
            atomicAdd(&a.counters->overflow, ovf);
        }
        if (ovf) atomicOr(a.errorFlag, 1);
    }
}

// A path the path-pool kernel could not settle (an order-ambiguous hit, a ray with a zero direction component, a
// stack beyond its overflow area): traced here in reference order, its radiance and segment count put where the
// pool kernel would have put them, before the batch's samples are added up.
struct PathRedo { uint32_t slot; uint32_t pixel; int32_t sample; };      // slot = sample in the batch * work items of the batch + work item
// One path per wave (lane 0; the nested loops of the reference order serialise lanes anyway); a fixed grid strides
// over however many paths the batch listed.  `count` may exceed `cap`: the pool kernel falls back to handing the
// whole pixel over once the list is full.
__global__ __launch_bounds__(64) void kFaithfulPaths(RefScene sc, Sky sky, RenderArgs a, const PathRedo* list, const int32_t* count, int cap,
                                                     float4* sampleOut, uint32_t nWork, int32_t firstSample) {
    int n = *count;
    if (n > cap) n = cap;
    int32_t blasStack[kRefBlasStack];
    uint32_t overflow = 0;
    if (threadIdx.x == 0) {
        for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {
            const PathRedo r = list[i];
            LocalCounters cnt = {0, 0, 0, 0, 0, 0, 0, 0};
            const V3 L = refTracePath(sc, sky, a, (int)(r.pixel % (uint32_t)a.width), (int)(r.pixel / (uint32_t)a.width),
                                      firstSample + (int32_t)(r.slot / nWork), blasStack, cnt);
            sampleOut[r.slot] = make_float4(L.x, L.y, L.z, __uint_as_float(cnt.rays));
            overflow += cnt.overflow;
        }
        if (overflow) atomicOr(a.errorFlag, 1);
    }
}

// Adds the spp samples of every listed pixel in sample order and writes the pixel.
__global__ void kSumListedSamples(RenderArgs a, const int32_t* list, int listCount, const float4* samples) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= listCount) return;
    V3 sum = v3(0, 0, 0);
    for (int s = 0; s < a.spp; s++) {
        float4 L = samples[(size_t)i * a.spp + s];
        sum = sum + v3(L.x, L.y, L.z);
    }
    writePathPixel(a, (size_t)list[i], sum);
}

}  // namespace rt

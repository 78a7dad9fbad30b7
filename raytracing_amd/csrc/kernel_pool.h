// Path-tracing kernel with more paths than lanes: the production path of PT mode
// (RTAMD_PIPELINE=persistent selects kFast<PT> instead, for comparison).
//
// kFast keeps one path per lane, so hit shading, sky shading, sample bookkeeping and camera-ray
// generation all run inside the traversal kernel with whatever lanes happen to need them (about a
// third of the wave each), and that code is 40 % of the kernel's instructions.  Here a wave owns
// kPoolSlots paths, each in a 128-byte record in HBM (L2 / Infinity-Cache resident: 16 KiB per
// wave).  A lane only traverses: when its ray is done it writes the hit into the path's record,
// queues the path for shading (a per-wave queue in LDS) and takes the next ready ray from a second
// per-wave queue.  When 64 hits have collected, the whole wave shades them — one path per lane,
// all 64 lanes busy, traversal state of the lanes untouched — and queues the follow-up rays;
// finished paths are replaced on the spot by new (pixel, sample) pairs.  The queues are private
// to a wave, so there is no synchronisation, no waiting and nothing to deadlock on.
//
// A finished sample's radiance goes to a sample buffer; kPoolSum adds each pixel's samples in
// sample order afterwards, so the image is bit-identical to kFast's and the oracle's.
#pragma once

#include "kernel_fast.h"

namespace rt {

// A path's record: a 64-byte hot part (what traversal and refilling touch) and a 32-byte cold part.
// Per wave the hot parts of its kPoolSlots slots come first, then the cold parts, so two hot records share a
// 128-byte line and the record stream through L2 and the fabric is 96 instead of 128 bytes per path.
//   hot 0   origin.xyz, direction.x
//   hot 1   direction.yz, invDir.xy
//   hot 2   invDir.z, bounce (bit 16: inside glass), RNG key, -
//   hot 3   -
//   cold 0  throughput.xyz, index of the path in the batch's sample buffer
//   cold 1  radiance gathered so far .xyz (read and written only where a surface emits or the path ends), pixel
// What a finished ray hands to the shading pass does not go through memory: t, u, v, triangle rank and the
// record's state (instance rank of the hit; kPoolMiss; kPoolUnsettled: hand the pixel over; kPoolFresh: no path)
// stay in LDS next to the queues (20 bytes per slot).  Two write requests per ray segment less at L2, each
// measured to cost 1.5 % on the 1M-triangle scene.
struct alignas(32) PoolRec { float4 piece[6]; };
static_assert(sizeof(PoolRec) == 96, "64 hot + 32 cold bytes per path");

constexpr int32_t kPoolMiss = -1, kPoolUnsettled = -3, kPoolFresh = -4;
#ifndef RT_POOL_SLOTS
#define RT_POOL_SLOTS 128
#endif
constexpr int kPoolSlots = RT_POOL_SLOTS;      // paths per wave (record slots); 144 .. 192 gain 1-2 % on a scene that fits the caches and lose 2-4 % on one that does not
constexpr int kPoolQueue = RT_POOL_SLOTS > 128 ? 256 : 128;      // entries per queue (power of two >= kPoolSlots; a slot index fits a byte)
constexpr int kPoolChunk = 256;      // path indices a wave takes from the batch at a time (fewer when the batch is small)
#ifndef RT_POOL_HOT_NODES
#define RT_POOL_HOT_NODES 456
#endif
#ifndef RT_POOL_HOT_NODES_BIG
#define RT_POOL_HOT_NODES_BIG 164
#endif
constexpr int kPoolHotNodesBig = RT_POOL_HOT_NODES_BIG;   // the variant for large scenes: 18 KiB of nodes, and 32 KiB for the paths' throughput words instead
constexpr int kPoolBigSceneNodes = 16384;                 // scenes with more nodes than this run the large-scene variant (RTAMD_POOL_BIG=0/1 overrides; the 50 M-instanced scene, 31 760 nodes: +1 %)
constexpr int kPoolHotNodes = RT_POOL_HOT_NODES;   // LDS: 64 KiB stacks + 6 KiB queues + 40 KiB hit results and states + 50 KiB nodes

struct PoolBatch {
    PoolRec* recs;                // [waves of the grid][kPoolSlots]
    float4* sampleL;              // [sample in batch][work item]: radiance, segment count (bits)
    float4* runSum;               // [work item]: running sum, segment count (bits)
    unsigned long long* next;     // next path of the batch to hand out
    uint32_t nWork;               // work items = 1024 * super-tiles of this chunk
    int32_t superBase;            // first super-tile of the chunk in FastScene::superTiles
    int32_t sampleBase;           // first sample of the batch (relative to sample_offset)
    int32_t nSamples;             // samples in the batch
    int32_t firstBatch, lastBatch;
    int32_t hotTlas, hotBlas;     // hot-node ranges for kPoolHotNodes
    int32_t chunk;                // path indices a wave takes at a time
    // paths this kernel cannot settle go to the reference-order kernel one by one (kFaithfulPaths, between this
    // launch and the summation); when the list is full the whole pixel is handed over instead
    PathRedo* pathList;
    int32_t* pathCount;           // entries of this batch (reset by kPoolSum)
    int32_t* pathTotal;           // entries of the whole render
    int32_t pathCap;
};

// work item -> pixel; false when the item lies outside the image or the requested rectangle
__device__ __forceinline__ bool poolPixel(const FastScene& sc, const RenderArgs& a, const PoolBatch& b, uint32_t w, int& x, int& y) {
    const int st = sc.superTiles[b.superBase + (int)(w >> 10)];     // tile column | tile row << 16
    const int within = (int)(w & 1023u);
    const int sx = st & 0xffff, sy = st >> 16;
    const int t8 = within >> 6, p8 = within & 63;
    x = sx * 32 + (t8 & 3) * 8 + (p8 & 7);
    y = sy * 32 + (t8 >> 2) * 8 + (p8 >> 3);
    return x < a.width && y < a.height && x >= a.tileX && y >= a.tileY && x < a.tileX + a.tileW && y < a.tileY + a.tileH;
}

// Path records keep the default cache policy: they are re-read tens of microseconds after they were written,
// and moving them with the non-temporal policy (to leave L1 / L2 to the BVH) was measured 22-35 % slower.
__device__ __forceinline__ float4 recLoad(const float4* p) { return *p; }
__device__ __forceinline__ void recStore(float4* p, float4 v) { *p = v; }

// A finished sample: written once, read once by kPoolSum after the launch; stored past the caches'
// usual policy so that 16 bytes per path do not push BVH nodes and path records out of them.
__device__ __forceinline__ void poolStoreSample(float4* dst, V3 L, int segments) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = {L.x, L.y, L.z, __int_as_float(segments)};
    __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst));
}

// path `slot` of the batch = sample `slot / nWork` of pixel `pix`
__device__ __forceinline__ void poolHandOver(const RenderArgs& a, uint32_t p);
__device__ __forceinline__ void poolHandOverPath(const RenderArgs& a, const PoolBatch& b, uint32_t slot, uint32_t pix) {
    const int i = atomicAdd(b.pathCount, 1);
    if (i < b.pathCap) {
        PathRedo r;
        r.slot = slot; r.pixel = pix;
        r.sample = 0;        // (kFaithfulPaths works the sample number out of the slot: no division by the batch's width here)
        b.pathList[i] = r;
        atomicAdd(b.pathTotal, 1);
    } else poolHandOver(a, pix);
}

__device__ __forceinline__ void poolHandOver(const RenderArgs& a, uint32_t p) {
    const uint32_t bit = 1u << (p & 31u);
    const uint32_t old = atomicOr(a.redoMask + (p >> 5), bit);
    if (!(old & bit)) {
        const int i = atomicAdd(a.redoCount, 1);
        if (i < a.redoCap) a.redoList[i] = (int32_t)p;
        else atomicOr(a.errorFlag, 4);
    }
}

// STATS = true builds a diagnostic variant that also counts, per phase, how often the wave executed
// it and with how many active lanes (written to `stats`, 25 counters; see launchPool for the names).
// DIEL = true (RTAMD_FLAG_DIELECTRICS): glass objects refract; bit 16 of a record's bounce word says that
// the path is inside glass, and such a ray tests triangles from both sides.
constexpr int32_t kPoolInsideBit = 1 << 16;
// BIG = true: the variant for scenes whose nodes are far too many for LDS anyway (the LDS copy of the top of the
// trees measured worth 0.3 % there between 160 and 456 nodes): most of the node copy's room holds piece cold 0
// of every record instead, which takes one more read and one more write request per ray segment from L2.
template <bool STATS, bool DIEL = false, bool BIG = false>
__global__ __launch_bounds__(kBlockThreads, RT_MIN_WAVES) void kPool(rtamd_frame_cbuffer camera /* first: read through the argument block's address */,
                                                                       FastScene sc, Sky sky, RenderArgs a, PoolBatch b, int refillBelow,
                                                                       int nodeMin, int leafGroupMin, int stackLimit, int poolSlots,
                                                                       unsigned long long* stats) {
    unsigned long long st_[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // 12 phases x (executions, lanes)
#define RT_STAT(slot, cond)                                                         \
    do {                                                                            \
        if (STATS) {                                                                \
            unsigned long long m_ = __ballot(cond);                                 \
            if (m_) { st_[2 * (slot)] += 1; st_[2 * (slot) + 1] += __popcll(m_); }  \
        }                                                                           \
    } while (0)
#define RT_STAT2(slot, cond) RT_STAT(slot, cond)
    __shared__ int32_t ldsStack[kLdsStack][kBlockThreads];
    __shared__ float4 ldsNodes[(BIG ? kPoolHotNodesBig : kPoolHotNodes) * kHotStride];
    __shared__ float4 ldsCold0[BIG ? (kBlockThreads / 64) * kPoolSlots : 1];
    __shared__ uint8_t ldsQueues[kBlockThreads / 64][3][kPoolQueue];
    // what a finished ray leaves for the shading pass stays in LDS (written by one lane of the wave, read by another):
    // t, u, v, triangle of the hit, and the record's state
    __shared__ float4 ldsHit[kBlockThreads / 64][kPoolSlots];
    __shared__ int32_t ldsState[kBlockThreads / 64][kPoolSlots];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (the same in every lane: addresses formed from it stay in scalar registers)
    // three queues of record slots, private to the wave:
    uint8_t* const rayQ = ldsQueues[wave][0];     // paths with a ray ready to traverse
    uint8_t* const hitQ = ldsQueues[wave][1];     // paths whose ray hit a surface: to be shaded and scattered
    uint8_t* const endQ = ldsQueues[wave][2];     // paths whose ray left the scene, that ended at a surface, that cannot be settled here,
                                                  // and records without a path: each gets (its sky light and) the next path of the batch
    // the wave's records (see PoolRec): hot parts [slot][4 pieces], then cold parts [slot][2 pieces]
    float4* const recHot = reinterpret_cast<float4*>(b.recs + ((size_t)blockIdx.x * (kBlockThreads / 64) + wave) * kPoolSlots);
    float4* const recCold = recHot + kPoolSlots * 4;
#define RT_HOT(piece, sl) (recHot + (sl) * 4 + (piece))
#define RT_COLD(piece, sl) (recCold + (sl) * 2 + (piece))
#define RT_REC_STATE(sl) (ldsState[wave][sl])
#define RT_COLD0_LD(sl) (BIG ? ldsCold0[wave * kPoolSlots + (sl)] : recLoad(RT_COLD(0, sl)))
#define RT_COLD0_ST(sl, v) do { if (BIG) ldsCold0[wave * kPoolSlots + (sl)] = (v); else recStore(RT_COLD(0, sl), (v)); } while (0)
#define RT_REC_BOUNCE(sl) (reinterpret_cast<int32_t*>(RT_HOT(2, sl))[1])
#define RT_REC_PIX(sl) (reinterpret_cast<uint32_t*>(RT_COLD(1, sl))[3])
    const unsigned long long totalPaths = (unsigned long long)b.nWork * (unsigned long long)b.nSamples;
    // number of set bits of a wave mask below this lane (the lane's place among the lanes the mask names)
#define RT_BELOW(m_) ((int)__builtin_amdgcn_mbcnt_hi((uint32_t)((m_) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(m_), 0u)))

    const int hotTlas = b.hotTlas, hotBlas = b.hotBlas;
    copyHotNodes(ldsNodes, sc, hotTlas, hotBlas, tid);
    __syncthreads();
    const int32_t rootWord = hotWord(sc.tlasRoot, hotTlas, sc.blasBase, hotBlas);

    // the same in every lane of the wave
    int rayHead = 0, rayTail = 0, hitHead = 0, hitTail = 0, endHead = 0, endTail = 0;
    int nAlloc = 0;               // record slots handed out so far
    unsigned long long pathNext = 0, pathEnd = 0;   // the wave's current chunk of path indices
    uint32_t nextItem = 0;        // pathNext as (sample, work item): pathNext = nextSmp * nWork + nextItem
    int nextSmp = 0;
    bool noMoreWork = false;

    // lane state
    bool trav = false, unsettled = false;
    bool twoSided = false;        // DIEL: the ray this lane traverses is inside glass
    int32_t slot = -1;            // the path this lane traverses
    Ray world = makeRay(v3(0, 0, 0), v3(0, 0, 1));
    Ray cur = world;
    FastHit best;
    best.t = RT_INF; best.u = best.v = 0; best.inst = -1; best.tri = -1;
    float secondT = RT_INF, bestNear = 0.0f, instNear = 0.0f;
    int32_t node = 0;
    int sp = 0;
    int instSp = kNoInstance;
    int spilled = 0;
    int32_t curInst = -1;
    int deepest = 0;          // STATS: largest stack height of this lane
#define RT_WORLD_RAY() world
#include "traverse_macros.inc"
    // a ray that cannot be settled here: its pixel goes to the reference-order kernel (when the path is shaded)
#define RT_REDO() do { unsettled = true; trav = false; } while (0)

    for (;;) {
        // ---------------------------------------- finished lanes publish their results
        {
            const bool fin = !trav && slot >= 0;
            const bool amb = best.t < RT_INF && secondT < bestNear;   // order-dependent in the reference: see kernel_fast.h
            const bool toHit = fin && best.inst >= 0 && !unsettled && !amb;
            const unsigned long long mh = __ballot(toHit), me = __ballot(fin && !toHit);
            RT_STAT(5, fin);
            if (fin) {
                const int rs = slot;
                if (toHit) {
                    RT_REC_STATE(rs) = best.inst;
                    hitQ[(hitTail + RT_BELOW(mh)) & (kPoolQueue - 1)] = (uint8_t)slot;
                } else {
                    RT_REC_STATE(rs) = (unsettled || amb) ? kPoolUnsettled : kPoolMiss;
                    endQ[(endTail + RT_BELOW(me)) & (kPoolQueue - 1)] = (uint8_t)slot;
                }
                slot = -1;
            }
            hitTail += __popcll(mh);
            endTail += __popcll(me);
            __threadfence_block();      // records and queue entries are read by other lanes of the wave below
        }
        // ---------------------------------------- shading passes, 64 queued paths at a time, one kind of work per pass
        for (;;) {
            const int idle = __popcll(__ballot(!trav));
            const int hitCnt = hitTail - hitHead, endCnt = endTail - endHead, rayCnt = rayTail - rayHead;
            const bool canStart = !noMoreWork && nAlloc < poolSlots;       // record slots never used so far
            int kind = 0;             // 1: surface hits; 2: path ends and starts
            if (hitCnt >= 64) kind = 1;
            else if (endCnt >= 64) kind = 2;
            else if (rayCnt < idle) {
                // lanes would go idle: whichever queue holds more, or brand-new paths
                if (hitCnt > 0 && hitCnt >= endCnt) kind = 1;
                else if (endCnt > 0 || canStart) kind = 2;
            }
            if (kind == 0) break;
            __builtin_amdgcn_wave_barrier();
            bool goOn = false, ended = false;
            int32_t s = -1;
            if (kind == 1) {
                // ---- surface hits: emission, scattering, the next ray
                const int n = hitCnt < 64 ? hitCnt : 64;
                if (lane < n) s = hitQ[(hitHead + lane) & (kPoolQueue - 1)];
                hitHead += n;
                RT_STAT(0, s >= 0);
                if (s >= 0) {
                    const int rs = s;
                    const float4 meta = recLoad(RT_HOT(2, rs));
                    const float4 r0 = recLoad(RT_HOT(0, rs)), r1 = recLoad(RT_HOT(1, rs)), h4 = ldsHit[wave][rs], e4 = RT_COLD0_LD(rs);
                    const int32_t inst = RT_REC_STATE(rs);
                    const uint32_t key = __float_as_uint(meta.z);
                    const int bounce = DIEL ? (__float_as_int(meta.y) & (kPoolInsideBit - 1)) : __float_as_int(meta.y);
                    bool insideGlass = DIEL && (__float_as_int(meta.y) & kPoolInsideBit) != 0;
                    Ray w;
                    w.origin = v3(r0.x, r0.y, r0.z);
                    w.direction = v3(r0.w, r1.x, r1.y);
                    w.invDir = v3(r1.z, r1.w, meta.x);
                    V3 energy = v3(e4.x, e4.y, e4.z);
                    FastHit h;
                    h.t = h4.x; h.u = h4.y; h.v = h4.z; h.inst = inst; h.tri = __float_as_int(h4.w);
                    const Surf sf = resolveHit(sc, w, h);
                    // light picked up here: most surfaces emit nothing, and then the radiance word is neither read nor
                    // written (adding a zero leaves its bits; a NaN contribution compares unequal to 0 and is added)
                    const V3 gain = energy * sf.emission;
                    const bool lit = gain.x != 0.0f || gain.y != 0.0f || gain.z != 0.0f;
                    V3 org, dir;
                    bool write = true;
                    if (bounce < a.maxBounces && scatterSurface<DIEL>(sf, w.direction, key, (uint32_t)bounce, energy, org, dir, insideGlass)) {
                        const Ray next = makeRay(org, dir);
                        write = false;
                        if (rayDegenerate(next)) { poolHandOverPath(a, b, __float_as_uint(e4.w), RT_REC_PIX(rs)); ended = true; }
                        else {
                            recStore(RT_HOT(0, rs), make_float4(next.origin.x, next.origin.y, next.origin.z, next.direction.x));
                            recStore(RT_HOT(1, rs), make_float4(next.direction.y, next.direction.z, next.invDir.x, next.invDir.y));
                            recStore(RT_HOT(2, rs), make_float4(next.invDir.z, __int_as_float((bounce + 1) | (insideGlass ? kPoolInsideBit : 0)), meta.z, meta.w));
                            RT_COLD0_ST(rs, make_float4(energy.x, energy.y, energy.z, e4.w));
                            if (lit) {
                                const float4 l4 = recLoad(RT_COLD(1, rs));
                                const V3 L = v3(l4.x, l4.y, l4.z) + gain;
                                recStore(RT_COLD(1, rs), make_float4(L.x, L.y, L.z, l4.w));
                            }
                            goOn = true;
                        }
                    }
                    if (write) {
                        const float4 l4 = recLoad(RT_COLD(1, rs));
                        const V3 L = v3(l4.x, l4.y, l4.z) + gain;
                        poolStoreSample(b.sampleL + __float_as_uint(e4.w), L, bounce + 1);
                        ended = true;
                    }
                    if (ended) RT_REC_STATE(rs) = kPoolFresh;      // the record waits in the other queue for its next path
                }
            } else {
                // ---- path ends (sky light, hand-over) and starts: queued records first, then record slots never used so far
                const int n = endCnt < 64 ? endCnt : 64;
                int nNew = 0;
                if (canStart) { nNew = 64 - n; if (nNew > poolSlots - nAlloc) nNew = poolSlots - nAlloc; }
                if (lane < n) s = endQ[(endHead + lane) & (kPoolQueue - 1)];
                else if (lane < n + nNew) s = nAlloc + (lane - n);
                endHead += n;
                nAlloc += nNew;
                RT_STAT(8, s >= 0);
                if (s >= 0) {
                    ended = true;
                    if (lane < n) {
                        const int rs = s;
                        const int32_t inst = RT_REC_STATE(rs);
                        if (inst == kPoolMiss) {
                            const float4 meta = recLoad(RT_HOT(2, rs));
                            const float4 r0 = recLoad(RT_HOT(0, rs)), r1 = recLoad(RT_HOT(1, rs)), e4 = RT_COLD0_LD(rs), l4 = recLoad(RT_COLD(1, rs));
                            const int bounce = DIEL ? (__float_as_int(meta.y) & (kPoolInsideBit - 1)) : __float_as_int(meta.y);
                            V3 L = v3(l4.x, l4.y, l4.z) + v3(e4.x, e4.y, e4.z) * shadeMiss(sky, v3(r0.w, r1.x, r1.y));
                            poolStoreSample(b.sampleL + __float_as_uint(e4.w), L, bounce + 1);
                        } else if (inst == kPoolUnsettled) {
                            const float4 e4 = RT_COLD0_LD(rs);
                            poolHandOverPath(a, b, __float_as_uint(e4.w), RT_REC_PIX(rs));
                        }
                    }
                }
            }
            // ended (and never used) records take the next (pixel, sample) of the batch
            bool fresh = false;      // the record holds a placeholder that has to come round again
            {
                const unsigned long long me = __ballot(ended);
                if (kind == 2 && me != 0ull && !noMoreWork) {
                    // path indices come in chunks of b.chunk per wave (one atomic per chunk); a lane
                    // that finds the chunk used up keeps a placeholder and asks again in the next pass
                    if (pathNext == pathEnd) {
                        unsigned long long got = 0;
                        if (lane == 0) got = atomicAdd(b.next, (unsigned long long)b.chunk);
                        got = __shfl(got, 0, 64);
                        if (got >= totalPaths) noMoreWork = true;
                        else {
                            pathNext = got; pathEnd = totalPaths - got < (unsigned long long)b.chunk ? totalPaths : got + b.chunk;
                            nextSmp = (int)(got / b.nWork);                 // (one division per chunk, not per path)
                            nextItem = (uint32_t)(got - (unsigned long long)nextSmp * b.nWork);
                        }
                    }
                    const unsigned long long base = pathNext;
                    const unsigned long long avail = pathEnd - pathNext;
                    const unsigned long long wantN = (unsigned long long)__popcll(me);
                    const uint32_t take = (uint32_t)(wantN < avail ? wantN : avail);
                    const uint32_t baseItem = nextItem;
                    const int baseSmp = nextSmp;
                    pathNext += take;
                    nextItem += take;
                    if (nextItem >= b.nWork) { nextItem -= b.nWork; nextSmp++; }      // (a pass takes at most 64 <= nWork paths)
                    if (ended) {
                        const uint32_t mine = (uint32_t)RT_BELOW(me);
                        const unsigned long long p = base + mine;
                        if (noMoreWork) { }
                        else if (p >= pathEnd) { RT_REC_STATE(s) = kPoolFresh; fresh = true; }
                        else {
                            const int rs = s;
                            uint32_t wItem = baseItem + mine;
                            int smp = baseSmp;
                            if (wItem >= b.nWork) { wItem -= b.nWork; smp++; }
                            bool live = false;
                            int x, y;
                            if (poolPixel(sc, a, b, wItem, x, y)) {
                                const uint32_t pix = (uint32_t)y * (uint32_t)a.width + (uint32_t)x;
                                if (!((a.redoMask[pix >> 5] >> (pix & 31u)) & 1u)) {
                                    const uint32_t key = rngKey(a.seedHi, a.seedLo, pix, (uint32_t)(a.sampleOffset + b.sampleBase + smp));
                                    // (the camera block is read here, from the argument block: see cameraArg)
                                    (void)camera;
                                    const CameraArgPtr camPtr = cameraArg();
                                    float jx = camPtr->PixelOffset[0], jy = camPtr->PixelOffset[1];
                                    if (!(a.flags & RTAMD_FLAG_NO_JITTER)) { jx = rngDraw(key, 0u); jy = rngDraw(key, 1u); }
                                    // (The frame's size as floats, the zero of the shader's float4(uv, 0, 1) and the 1 of a new path's
                                    // throughput are the same for every path; formed from values the compiler cannot see
                                    // through, they are made here, where they are needed, instead of once for the whole kernel
                                    // in registers that then sit in scratch memory.)
                                    int wOpaque = a.width, hOpaque = a.height;
                                    float zeroOpaque = 0.0f;
                                    asm volatile("" : "+s"(wOpaque), "+s"(hOpaque));
                                    asm volatile("" : "+v"(zeroOpaque));
                                    const Ray r = cameraRayArg(camPtr, pixelToNdc(x, jx, wOpaque), pixelToNdc(y, jy, hOpaque), zeroOpaque);
                                    if (rayDegenerate(r)) poolHandOverPath(a, b, (uint32_t)p, pix);
                                    else {
                                        recStore(RT_HOT(0, rs), make_float4(r.origin.x, r.origin.y, r.origin.z, r.direction.x));
                                        recStore(RT_HOT(1, rs), make_float4(r.direction.y, r.direction.z, r.invDir.x, r.invDir.y));
                                        recStore(RT_HOT(2, rs), make_float4(r.invDir.z, __int_as_float(0), __uint_as_float(key), __int_as_float(0)));
                                        float one = 1.0f;
                                        asm volatile("" : "+v"(one));
                                        RT_COLD0_ST(rs, make_float4(one, one, one, __uint_as_float((uint32_t)p)));
                                        recStore(RT_COLD(1, rs), make_float4(0.f, 0.f, 0.f, __uint_as_float(pix)));
                                        live = true;
                                    }
                                }
                            }
                            if (live) goOn = true;
                            else { RT_REC_STATE(rs) = kPoolFresh; fresh = true; }     // no such path: take another one next time
                        }
                    }
                } else if (kind == 1 && ended) fresh = true;          // ended at a surface: queued for a new path
            }
            // queue the paths that have a ray to traverse; placeholders go (back) to the end queue
            {
                const unsigned long long mg = __ballot(goOn);
                if (goOn) rayQ[(rayTail + RT_BELOW(mg)) & (kPoolQueue - 1)] = (uint8_t)s;
                rayTail += __popcll(mg);
                const unsigned long long mf = __ballot(fresh);
                if (fresh) endQ[(endTail + RT_BELOW(mf)) & (kPoolQueue - 1)] = (uint8_t)s;
                endTail += __popcll(mf);
            }
            __threadfence_block();
        }
        // ---------------------------------------- idle lanes take the next ready rays
        {
            const bool idleLane = !trav;
            const unsigned long long m = __ballot(idleLane);
            const int rayCnt = rayTail - rayHead;
            const int want = __popcll(m);
            const int k = want < rayCnt ? want : rayCnt;
            const int mine = RT_BELOW(m);
            if (BIG) {
                if (idleLane && mine < k) {
                    slot = rayQ[(rayHead + mine) & (kPoolQueue - 1)];
                    {
                        const float4 r0 = recLoad(RT_HOT(0, slot)), r1 = recLoad(RT_HOT(1, slot));
                        const float iz = RT_HOT(2, slot)->x;
                        world.origin = v3(r0.x, r0.y, r0.z);
                        world.direction = v3(r0.w, r1.x, r1.y);
                        world.invDir = v3(r1.z, r1.w, iz);
                    }
                    if (DIEL) twoSided = (RT_REC_BOUNCE(slot) & kPoolInsideBit) != 0;
                    cur = world;
                    best.t = RT_INF; best.u = best.v = 0; best.inst = -1; best.tri = -1;
                    secondT = RT_INF; bestNear = 0.0f;
                    node = rootWord; sp = 0; instSp = kNoInstance; curInst = -1; spilled = 0;
                    unsettled = false;
                    trav = true;
                }
            } else {
                // Small-scene variant: every lane reads its path's world-space ray from the record here, the lanes that
                // go on traversing too (the ray is needed again when the lane leaves an instance): nine registers that
                // need not survive the shading passes, where the register file is full and the compiler keeps traversal
                // state in scratch memory (+1.7 % on the spheres scene; the large-scene variant, bound by L2 requests,
                // loses 1 % to the extra record reads and reads the ray only when a lane takes a new path).
                const bool taken = idleLane && mine < k;
                if (taken) slot = rayQ[(rayHead + mine) & (kPoolQueue - 1)];
                {
                    const int ws = slot < 0 ? 0 : slot;
                    const float4 r0 = recLoad(RT_HOT(0, ws)), r1 = recLoad(RT_HOT(1, ws));
                    const float iz = RT_HOT(2, ws)->x;
                    world.origin = v3(r0.x, r0.y, r0.z);
                    world.direction = v3(r0.w, r1.x, r1.y);
                    world.invDir = v3(r1.z, r1.w, iz);
                }
                if (taken) {
                    if (DIEL) twoSided = (RT_REC_BOUNCE(slot) & kPoolInsideBit) != 0;
                    cur = world;
                    best.t = RT_INF; best.u = best.v = 0; best.inst = -1; best.tri = -1;
                    secondT = RT_INF; bestNear = 0.0f;
                    node = rootWord; sp = 0; instSp = kNoInstance; curInst = -1; spilled = 0;
                    unsettled = false;
                    trav = true;
                }
            }
            rayHead += k;
        }
        if (__ballot(trav) == 0ull) {
            // nothing to traverse: more shading / new paths, or (all queues empty, no work left) the end
            if (hitTail - hitHead == 0 && endTail - endHead == 0 && rayTail - rayHead == 0) break;
            continue;
        }
        // a lane is served when it has a result to publish
#define RT_CAN_REFILL (!trav && slot >= 0)
#define RT_TWO_SIDED (DIEL && twoSided)
        // the closest hit's t, u, v, triangle go straight to the slot's LDS entry (two registers less to carry)
#define RT_KEEP_HIT(t_, u_, v_, tri_) do { ldsHit[wave][slot] = make_float4((t_), (u_), (v_), __int_as_float(tri_)); } while (0)
#define RT_LOCAL_RAY_UNSETTLED(rank_) rayDegenerate(cur)
#define RT_BAD_INDEX() do { } while (0)
        constexpr bool CHECKED = false;      // (the bounds-checked diagnostic variant is kFast's)
#include "traverse_body.inc"
#undef RT_BAD_INDEX
#undef RT_LOCAL_RAY_UNSETTLED
#undef RT_KEEP_HIT
#undef RT_TWO_SIDED
#undef RT_CAN_REFILL
    }
#undef RT_PUSH
#undef RT_PUSH_IF
#undef RT_POP
#undef RT_MAKE_ROOM
#undef RT_SPILL_AT
#undef RT_WORLD_RAY
#undef RT_REDO
#undef RT_STAT
#undef RT_STAT2
#undef RT_HOT
#undef RT_COLD
#undef RT_REC_STATE
#undef RT_BELOW
#undef RT_COLD0_LD
#undef RT_COLD0_ST
#undef RT_REC_PIX
#undef RT_REC_BOUNCE
    if (STATS) {
        for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(deepest, off, 64); deepest = o > deepest ? o : deepest; }
        if (lane == 0) {
            for (int i = 0; i < 24; i++) if (st_[i]) atomicAdd(stats + i, st_[i]);
            atomicMax(stats + 24, (unsigned long long)deepest);
        }
    }
}

// Adds the batch's samples to each pixel's running sum, in sample order; the last batch writes
// the pixel and counts its ray segments.
__global__ __launch_bounds__(256) void kPoolSum(FastScene sc, RenderArgs a, PoolBatch b) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w == 0) b.pathCount[0] = 0;       // (the next batch starts an empty list)
    unsigned long long rays = 0;
    if (w < b.nWork) {
        int x, y;
        if (poolPixel(sc, a, b, w, x, y)) {
            float4 run = b.firstBatch ? make_float4(0.f, 0.f, 0.f, 0.f) : b.runSum[w];
            V3 sum = v3(run.x, run.y, run.z);
            uint32_t segs = __float_as_uint(run.w);
            for (int s = 0; s < b.nSamples; s++) {
                const float4 v = b.sampleL[(size_t)s * b.nWork + w];
                sum = sum + v3(v.x, v.y, v.z);
                segs += __float_as_uint(v.w);
            }
            if (b.lastBatch) {
                const uint32_t pix = (uint32_t)y * (uint32_t)a.width + (uint32_t)x;
                if (!((a.redoMask[pix >> 5] >> (pix & 31u)) & 1u)) {
                    writePathPixel(a, pix, sum);
                    rays = segs;
                }
            } else {
                b.runSum[w] = make_float4(sum.x, sum.y, sum.z, __uint_as_float(segs));
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rays += __shfl_xor(rays, off, 64);
    if ((threadIdx.x & 63) == 0 && rays) atomicAdd(a.rayCount, rays);
}

struct PoolHost {
    DBuf<PoolRec> recs;
    DBuf<float4> sampleL, runSum;
    DBuf<PathRedo> pathList;
    static constexpr int kPathCap = 65536;
    // A frame of several batches alternates between two halves of the sample buffer and of the path list: while
    // the kernel traces batch i + 1 on the caller's stream, the paths it passed on in batch i are traced and
    // batch i is added up on a second stream.
    hipStream_t side = nullptr;
    hipEvent_t evTraced[2] = {nullptr, nullptr}, evSummed[2] = {nullptr, nullptr};
    hipError_t sideStream() {
        if (side) return hipSuccess;
        hipError_t err = hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
        for (int k = 0; k < 2 && err == hipSuccess; k++) {
            err = hipEventCreateWithFlags(&evTraced[k], hipEventDisableTiming);
            if (err == hipSuccess) err = hipEventCreateWithFlags(&evSummed[k], hipEventDisableTiming);
        }
        return err;
    }
    // device time of the kPool launches of the last render: one pair of events per launch
    std::vector<hipEvent_t> evBegin, evEnd;
    int nTimed = 0;
    // frames that return before they are complete (rtamd_render_begin): frame k uses half k & 1 of the sample buffer
    // and of the path list, and a pair of timing events of its own
    hipEvent_t asyncBegin[2] = {nullptr, nullptr}, asyncEnd[2] = {nullptr, nullptr};
    bool halfSummedOnce[2] = {false, false};
    hipError_t asyncEvents() {
        hipError_t err = hipSuccess;
        for (int k = 0; k < 2 && err == hipSuccess; k++) {
            if (!asyncBegin[k]) err = hipEventCreate(&asyncBegin[k]);
            if (err == hipSuccess && !asyncEnd[k]) err = hipEventCreate(&asyncEnd[k]);
        }
        return err;
    }
    void release() {
        recs.release(); sampleL.release(); runSum.release(); pathList.release();
        for (hipEvent_t ev : evBegin) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : evEnd) (void)hipEventDestroy(ev);
        evBegin.clear(); evEnd.clear();
        nTimed = 0;
        for (int k = 0; k < 2; k++) {
            if (asyncBegin[k]) (void)hipEventDestroy(asyncBegin[k]);
            if (asyncEnd[k]) (void)hipEventDestroy(asyncEnd[k]);
            asyncBegin[k] = asyncEnd[k] = nullptr;
            halfSummedOnce[k] = false;
            if (evTraced[k]) (void)hipEventDestroy(evTraced[k]);
            if (evSummed[k]) (void)hipEventDestroy(evSummed[k]);
            evTraced[k] = evSummed[k] = nullptr;
        }
        if (side) (void)hipStreamDestroy(side);
        side = nullptr;
    }
    // events of launch i (created on first use)
    hipError_t eventsFor(int i) {
        while ((int)evBegin.size() <= i) {
            hipEvent_t b = nullptr, e = nullptr;
            hipError_t err = hipEventCreate(&b);
            if (err == hipSuccess && (err = hipEventCreate(&e)) != hipSuccess) (void)hipEventDestroy(b);
            if (err != hipSuccess) return err;
            evBegin.push_back(b); evEnd.push_back(e);
        }
        return hipSuccess;
    }
    // after the stream has been synchronised
    float tracedMs() const {
        float total = 0.f;
        for (int i = 0; i < nTimed; i++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, evBegin[i], evEnd[i]) == hipSuccess) total += ms;
        }
        return total;
    }
};

// Paths per launch.  Every launch has a fixed cost (the pool fills, and drains while the slowest paths end: 1.7 ms
// on the 1M-triangle scene), so a batch is as large as memory reasonably allows: 2^29 paths (2 x 8 GB of samples
// for a frame of several batches) when that is at most an eighth of the device's memory, 2^27 at least.  Worked
// out per context, with its device current, when the context is created.
inline size_t poolMaxPathsForCurrentDevice() {
    size_t paths = (size_t)1 << 27;
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) == hipSuccess)
        while (paths < ((size_t)1 << 29) && 2 * (2 * paths) * sizeof(float4) <= totalB / 8) paths *= 2;
    else
        (void)hipGetLastError();
    return paths;
}
inline size_t poolMaxPaths(const FastSceneHost& fs) {
    size_t cap = fs.poolPathCap;
    if (const char* s = getenv("RTAMD_POOL_MAX_PATHS")) { long long v = atoll(s); if (v >= 1024) cap = (size_t)v; }
    if (cap > ((size_t)1 << 31)) cap = (size_t)1 << 31;       // (a path's place in its batch is a 32-bit number)
    return cap;
}

// PT-mode render of the pixels selected by `a` (radiance only: the identity of the primary
// hit comes from kFast<kModePrimary>, launched here first when asked for).
// asyncSlot >= 0 (rtamd_render_begin): the frame must fit one batch; it uses half `asyncSlot` of the sample buffer and
// of the path list, its hand-over paths and its summation always run on the second stream behind the trace, and
// *lastStream receives the stream the frame's last kernel was queued on.  A frame of the other kind that does not fit
// one batch returns hipErrorNotSupported here before anything is launched.
inline hipError_t launchPool(FastSceneHost& fs, PoolHost& ph, const Sky& sky, const RenderArgs& a, hipStream_t stream,
                             const RefScene& rsc, int32_t* pathCountA, int32_t* pathCountB, int32_t* pathTotal,
                             int asyncSlot = -1, hipStream_t* lastStream = nullptr) {
    if (asyncSlot < 0) ph.nTimed = 0;
    if (lastStream) *lastStream = stream;
    int tilesX = (a.width + 31) / 32;
    fs.hSuper.clear();
    int sx0 = a.tileX / 32, sx1 = (a.tileX + a.tileW - 1) / 32, sy0 = a.tileY / 32, sy1 = (a.tileY + a.tileH - 1) / 32;
    for (int sy = sy0; sy <= sy1; sy++)
        for (int sx = sx0; sx <= sx1; sx++) {
            int id = sy * tilesX + sx;
            if (a.shardCount > 1 && id % a.shardCount != a.shardRank) continue;
            fs.hSuper.push_back(sx | (sy << 16));
        }
    if (fs.hSuper.empty()) return hipSuccess;
    hipError_t e = fs.putSuperTiles(stream);
    if (e != hipSuccess) return e;

    const int numCU = fs.numCU;
    if (numCU <= 0) return hipErrorInvalidDevice;
    FastLaunchConfig cfg = fastLaunchConfig();
    FastLaunchConfig primaryCfg = cfg;
    chooseThresholds(primaryCfg, fs, kModePrimary);
    bool big = fs.nBlasNodes + (size_t)fs.nTlasNodes > (size_t)kPoolBigSceneNodes && !(a.flags & RTAMD_FLAG_DIELECTRICS);
    if (const char* bv = getenv("RTAMD_POOL_BIG")) big = atoi(bv) != 0 && !(a.flags & RTAMD_FLAG_DIELECTRICS);
    {
        // measured optima of the three benchmark scenes: shading is cheap here (full waves), so
        // lanes are refilled earlier than in kFast; small scenes (same criterion as kFast's) walk on with few lanes
        // (re-swept on the round's final kernels, tools/sweep_thresholds.sh: every node in LDS 12 / 36; nodes beyond LDS
        // but few enough for L2, the small-scene variant, 28 / 48; the large-scene variant 24 / 44.  Since the node
        // loads of the LDS lanes and of the others are separate instructions, a scene with at most twice the nodes
        // the copy holds — the random-spheres scene: about 480 — runs best on the first pair too: +2.4 %.)
        const bool allHot = fs.nBlasNodes + (size_t)fs.nTlasNodes <= 2 * (size_t)kPoolHotNodes;
        // (a large scene of many instances — the 50 M-instanced one: 1 251 — mixes instance entries and triangle leaves
        // in every leaf phase; there a small group of either kind is better served at once than kept waiting: group
        // threshold 12 instead of 20 and refill below 48: +1.4 %; the 1M-triangle scene, 7 instances, loses 2 % with 12)
        const bool manyInstances = big && fs.nInst >= 256;
        if (manyInstances && !getenv("RTAMD_LEAF_GROUP_MIN")) cfg.leafGroupMin = 12;
        if (cfg.refillBelow == 0) cfg.refillBelow = allHot ? 36 : (big ? (manyInstances ? 48 : 44) : 48);
        if (cfg.nodeMin == 0) {
            cfg.nodeMin = allHot ? 12 : (big ? 24 : 28);
            // a frame of one or two samples per pixel is over before the waves fall out of step: there the shorter
            // chain of steps counts for more than full node steps (1M-triangle scene, 1 spp: 2.47 -> 2.34 ms)
            const unsigned long long paths = (unsigned long long)fs.hSuper.size() * 1024ull * (unsigned long long)a.spp;
            if (paths <= (4ull << 20)) cfg.nodeMin = 8;
        }
    }

    FastScene sc;
    sc.nodes = fs.nodes.p; sc.inst = fs.inst.p; sc.shade = fs.shade.p;
    sc.tris = fs.tris.p; sc.nrms = fs.nrms.p; sc.triOrig = fs.triOrig.p; sc.superTiles = fs.superTiles.p; sc.nSuperTiles = (int)fs.hSuper.size();
    sc.tlasRoot = fs.tlasRoot;
    sc.planes = fs.planes.p; sc.planeRange = fs.planeRange.p; sc.planeTlas = fs.nBlasOfScene;
    sc.nNodes = (int32_t)fs.hNodes.size(); sc.nInst = fs.nInst; sc.nTriRecords = fs.nTriRecords; sc.nTriangles = fs.nTriangles;
    if ((e = fs.stackSpill.ensure((size_t)kSpillMax * (size_t)(numCU * cfg.blocksPerCU) * kBlockThreads)) != hipSuccess) return e;
    sc.stackSpill = fs.stackSpill.p;
    sc.spillMax = kSpillMax;
    sc.blasBase = fs.tlasReserve;
    auto hotRanges = [&](int budget, int32_t& hotTlas, int32_t& hotBlas) { hotNodeRanges(fs, budget, hotTlas, hotBlas); };
    hotRanges(kHotNodes, sc.hotTlas, sc.hotBlas);

    if (a.outIds || a.outT) {
        RenderArgs ids = a;
        ids.outRgba = nullptr; ids.accum = nullptr; ids.spp = 1;
        long long blocksNeeded = ((long long)fs.hSuper.size() * 1024 + kBlockThreads - 1) / kBlockThreads;
        int grid = numCU * cfg.blocksPerCU;
        if (blocksNeeded < grid) grid = (int)blocksNeeded;
        hipLaunchKernelGGL((kFast<kModePrimary, false>), dim3(grid), dim3(kBlockThreads), 0, stream, ids.cb, sc, sky, ids, primaryCfg.refillBelow, primaryCfg.nodeMin,
                           cfg.leafGroupMin, cfg.stackLimit, (unsigned long long*)nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = hipMemsetAsync(a.rayCount, 0, 2 * sizeof(unsigned long long), stream)) != hipSuccess) return e;
    }

    // chunks of super-tiles x batches of samples, each at most `cap` paths in the sample buffer
    // (16 bytes per path; when the allocation fails the cap is halved down to a million paths)
    const int gridMax = numCU * cfg.blocksPerCU;
    int poolSlots = kPoolSlots;
    if (const char* sv = getenv("RTAMD_POOL_SLOTS")) { int v = atoi(sv); if (v >= 64 && v <= kPoolSlots) poolSlots = v; }
    if ((e = ph.recs.ensure((size_t)gridMax * (kBlockThreads / 64) * kPoolSlots)) != hipSuccess) return e;
    if ((e = ph.pathList.ensure((size_t)2 * PoolHost::kPathCap)) != hipSuccess) return e;
    size_t tilesPerChunk = 0, workPerChunk = 0, samplesPerBatch = 0;
    for (size_t cap = poolMaxPaths(fs);; cap /= 2) {
        tilesPerChunk = cap / 1024;
        if (tilesPerChunk < 1) tilesPerChunk = 1;
        if (tilesPerChunk > fs.hSuper.size()) tilesPerChunk = fs.hSuper.size();
        workPerChunk = tilesPerChunk * 1024;
        samplesPerBatch = cap / workPerChunk;
        if (samplesPerBatch < 1) samplesPerBatch = 1;
        if (samplesPerBatch > (size_t)a.spp) samplesPerBatch = (size_t)a.spp;
        // (two halves when the frame takes more than one batch, or is one of two frames in flight)
        const bool several = workPerChunk < fs.hSuper.size() * 1024 || samplesPerBatch < (size_t)a.spp;
        if (asyncSlot >= 0 && several) return hipErrorNotSupported;
        e = ph.sampleL.ensure(workPerChunk * samplesPerBatch * (several || asyncSlot >= 0 ? 2 : 1));
        if (e == hipSuccess) e = ph.runSum.ensure(workPerChunk);
        if (e == hipSuccess) break;
        if (e != hipErrorOutOfMemory || cap <= ((size_t)1 << 20)) return e;
        (void)hipGetLastError();
    }

    if (cfg.stats) {
        if ((e = fs.stats.ensure(25)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(fs.stats.p, 0, 25 * sizeof(unsigned long long), stream)) != hipSuccess) return e;
    }
    const bool overlap = asyncSlot >= 0 || ((workPerChunk < fs.hSuper.size() * 1024 || samplesPerBatch < (size_t)a.spp) && !getenv("RTAMD_POOL_NO_OVERLAP"));
    if (overlap && (e = ph.sideStream()) != hipSuccess) return e;
    if (asyncSlot >= 0 && (e = ph.asyncEvents()) != hipSuccess) return e;
    int batchNo = 0;
    for (size_t t0 = 0; t0 < fs.hSuper.size(); t0 += tilesPerChunk) {
        const size_t nt = fs.hSuper.size() - t0 < tilesPerChunk ? fs.hSuper.size() - t0 : tilesPerChunk;
        for (int s0 = 0; s0 < a.spp; s0 += (int)samplesPerBatch, batchNo++) {
            const int half = asyncSlot >= 0 ? asyncSlot : (overlap ? (batchNo & 1) : 0);
            PoolBatch b;
            b.recs = ph.recs.p; b.sampleL = ph.sampleL.p + (size_t)half * workPerChunk * samplesPerBatch; b.runSum = ph.runSum.p;
            b.next = a.rayCount + 1;
            b.nWork = (uint32_t)(nt * 1024); b.superBase = (int32_t)t0;
            b.sampleBase = s0;
            b.nSamples = a.spp - s0 < (int)samplesPerBatch ? a.spp - s0 : (int)samplesPerBatch;
            b.firstBatch = s0 == 0; b.lastBatch = s0 + b.nSamples >= a.spp;
            b.pathList = ph.pathList.p + (size_t)half * PoolHost::kPathCap; b.pathCount = half ? pathCountB : pathCountA; b.pathTotal = pathTotal;
            b.pathCap = PoolHost::kPathCap;
            if (const char* pc = getenv("RTAMD_POOL_PATH_CAP")) { int v = atoi(pc); if (v >= 0 && v < b.pathCap) b.pathCap = v; }      // (tests: a short list)
            hotRanges(big ? kPoolHotNodesBig : kPoolHotNodes, b.hotTlas, b.hotBlas);
            const unsigned long long total = (unsigned long long)b.nWork * (unsigned long long)b.nSamples;
            // this half of the buffers is free once the batch before the last (the frame before the last) has been added up
            if (overlap && (asyncSlot >= 0 ? ph.halfSummedOnce[half] : batchNo >= 2) && (e = hipStreamWaitEvent(stream, ph.evSummed[half], 0)) != hipSuccess) return e;
            if ((e = hipMemsetAsync(a.rayCount + 1, 0, sizeof(unsigned long long), stream)) != hipSuccess) return e;
            long long blocksNeeded = (long long)((total + kBlockThreads - 1) / kBlockThreads);
            int grid = blocksNeeded < gridMax ? (int)blocksNeeded : gridMax;
            {
                // about eight chunks per wave even in a small batch, so that no wave ends up with twice the work of another
                const unsigned long long perWave = total / ((unsigned long long)grid * (kBlockThreads / 64) * 8ull);
                b.chunk = perWave >= (unsigned long long)kPoolChunk ? kPoolChunk : (perWave < 64ull ? 64 : (int)(perWave / 64ull) * 64);
                if (const char* cv = getenv("RTAMD_POOL_CHUNK")) { int v = atoi(cv); if (v >= 1 && v <= 65536) b.chunk = v; }
                if ((uint32_t)b.chunk > b.nWork) b.chunk = (int)b.nWork;
            }
            if (asyncSlot >= 0) e = hipEventRecord(ph.asyncBegin[asyncSlot], stream);
            else if ((e = ph.eventsFor(ph.nTimed)) == hipSuccess) e = hipEventRecord(ph.evBegin[ph.nTimed], stream);
            if (e != hipSuccess) return e;
#define RT_LAUNCH_POOL(S_, D_, B_, stats_) hipLaunchKernelGGL((kPool<S_, D_, B_>), dim3(grid), dim3(kBlockThreads), 0, stream, a.cb, sc, sky, a, b, cfg.refillBelow, cfg.nodeMin, cfg.leafGroupMin, cfg.stackLimit, poolSlots, stats_)
            if (a.flags & RTAMD_FLAG_DIELECTRICS) RT_LAUNCH_POOL(false, true, false, (unsigned long long*)nullptr);
            else if (cfg.stats && big) RT_LAUNCH_POOL(true, false, true, fs.stats.p);
            else if (cfg.stats) RT_LAUNCH_POOL(true, false, false, fs.stats.p);
            else if (big) RT_LAUNCH_POOL(false, false, true, (unsigned long long*)nullptr);
            else RT_LAUNCH_POOL(false, false, false, (unsigned long long*)nullptr);
#undef RT_LAUNCH_POOL
            if ((e = hipEventRecord(asyncSlot >= 0 ? ph.asyncEnd[asyncSlot] : ph.evEnd[ph.nTimed++], stream)) != hipSuccess) return e;
            // the paths the launch could not settle (a few hundred per frame; blocks without a path end at once), then
            // the summation: behind the launch on the second stream when more batches follow on this one
            hipStream_t after = stream;
            if (overlap) {
                if ((e = hipEventRecord(ph.evTraced[half], stream)) != hipSuccess) return e;
                if ((e = hipStreamWaitEvent(ph.side, ph.evTraced[half], 0)) != hipSuccess) return e;
                after = ph.side;
            }
            hipLaunchKernelGGL(kFaithfulPaths, dim3(2048), dim3(64), 0, after, rsc, sky, a, (const PathRedo*)b.pathList, (const int32_t*)b.pathCount, b.pathCap, b.sampleL,
                               b.nWork, a.sampleOffset + b.sampleBase);
            hipLaunchKernelGGL(kPoolSum, dim3((b.nWork + 255) / 256), dim3(256), 0, after, sc, a, b);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            if (overlap && (e = hipEventRecord(ph.evSummed[half], ph.side)) != hipSuccess) return e;
            if (asyncSlot >= 0) { ph.halfSummedOnce[half] = true; if (lastStream) *lastStream = ph.side; }
        }
    }
    if (asyncSlot >= 0) return hipSuccess;       // (the caller's stream goes straight on to the next frame; rtamd_render_wait waits for the second stream)
    // the caller's stream goes on when the last batch has been added up (the batches before it were added up earlier
    // on the same second stream)
    if (overlap && batchNo > 0 && (e = hipStreamWaitEvent(stream, ph.evSummed[(batchNo - 1) & 1], 0)) != hipSuccess) return e;
    if (cfg.stats) {
        // diagnostic build: per-phase execution counts and active lanes of the whole render, to stderr
        unsigned long long h[25];
        if ((e = hipMemcpyAsync(h, fs.stats.p, sizeof(h), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        const char* names[12] = {"hitPass", "nodeStep", "instEnter", "triLeaf(slot boxes)", "triTest", "publish", "leafPhase(trav lanes)", "nodeLoop(trav lanes)", "endStartPass",
                                 "nodeStep, stale", "instEnter, stale", "nodeStep, node in LDS"};
        for (int i = 0; i < 12; i++)
            if (h[2 * i]) fprintf(stderr, "[rtamd stats] kPool %-22s executions %12llu  mean active lanes %6.2f\n", names[i], h[2 * i], (double)h[2 * i + 1] / (double)h[2 * i]);
        fprintf(stderr, "[rtamd stats] kPool deepest traversal stack %llu of %d entries in LDS\n", h[24], kLdsStack);
    }
    return hipSuccess;
}

}  // namespace rt

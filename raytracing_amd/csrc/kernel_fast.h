// Production path: GPU-side scene layout, its construction at upload time, and the persistent
// one-path-per-lane kernel kFast (REF mode, the primary-id pass, and PT mode when asked for; PT
// mode normally runs through kPool, kernel_pool.h, which shares the traversal code below).
//
// What the kernel must reproduce is fixed by Assets/Shaders/RayTracingShader.compute:
// the closest front-facing triangle along each ray with the arithmetic of Trisection
// (:198-231) in the instance's local space (:287), and — when two triangles give the
// same t bit for bit — the one the reference's fixed visiting order reaches first
// (strict '<' at :327).  It does NOT have to visit nodes in the reference's order:
//   * triangles and instances are stored in the order the reference's depth-first walk
//     would reach them, so "reached first" is "smaller index" and ties are broken by
//     index, independent of the traversal order used here;
//   * each triangle (instance) keeps the box of its leaf slot in the caller's tree and is
//     tested (entered) only if the reference's slab test (:233-263) of that box passes;
//   * the hierarchy above the primitives is rebuilt at upload (wide_bvh.h): every inner
//     box is a union of those leaf-slot boxes, so inner boxes only prune by distance and
//     nodes can be visited nearest first;
//   * the one case where the caller's own inner boxes decide a hit — a ray with a zero
//     direction component lying exactly in a box face (0 * inf = NaN is ignored by min/max,
//     the box is missed and with it everything below) — is detected per ray segment, and the
//     pixel is handed to the reference-order kernel;
//   * the reference's result can depend on its visiting order when two triangles are hit at
//     distances a rounding error apart: it tests a slot only if the slot's box is entered before
//     the closest hit so far, and a box's computed entry distance may exceed the computed hit
//     distance of its own triangle by an ulp or two — whether that triangle is then tested
//     depends on what was found before it.  Here boxes are culled against the closest hit times
//     (1 + 2^-8), so such triangles are all tested, the closest one wins, and the segment is
//     declared ambiguous when another triangle was hit closer than the entry distance of the
//     winner's own boxes (slot box and instance box): only then could the reference have culled
//     the winner.  Ambiguous pixels go to the reference-order kernel as well.
//
// Layout (arrays 128-byte aligned in HBM; a lane's node is exactly one 128-byte line):
//   WNode  128 B  lx,hx,ly,hy,lz,hz (float4 = the four children), child[4], pad;
//                 the TLAS in front (breadth first, one slot per object reserved), then the BLAS nodes level by level
//   FTri    64 B  per triangle: leaf-slot box lo.xyz, hi.xyz, v0, e1, e2, rank (= position in the reference's visiting
//                 order); stored in leaf order of the new tree, two records to a 128-byte line, the pair interleaved
//                 component by component (kGatherLeafRecords, tlas_build.hip)
//   FNrm    48 B  n0,n1,n2, indexed by rank                  — read once per closest hit
//   FInst   96 B  three rows of WorldToLocal, the instance's TLAS slot box, BLAS root, original object index
//   FShade  96 B  three rows of LocalToWorld, albedo, specular, emission
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "kernel_common.h"
#include "fast_layout.h"
#include "tlas_build.h"
#include "wide_bvh.h"

namespace rt {

// Host-side owner of the GPU layout.
#ifndef RT_LEAF_TRIS
#define RT_LEAF_TRIS 2
#endif
// Triangles per leaf of the rebuilt BLAS trees (2 or 4; the kernels test them in pairs).  Two: a leaf is one
// 128-byte line, and fewer slot boxes and triangles are tested per visit than with the reference's four.
constexpr int kLeafTris = RT_LEAF_TRIS;
static_assert(kLeafTris == 2 || kLeafTris == 4, "leaves hold whole pairs of triangles");

struct FastSceneHost {
    DBuf<WNode> nodes;
    DBuf<FInst> inst;
    DBuf<FShade> shade;
    DBuf<FTri> tris;
    DBuf<FNrm> nrms;
    DBuf<int32_t> triOrig;
    DBuf<int32_t> superTiles;
    DBuf<int32_t> stackSpill;
    DBuf<unsigned long long> stats;
    std::vector<int32_t> blasRoot;     // wide-node index of each BLAS root
    int32_t nTriRecords = 0, nTriangles = 0;   // entries of tris / of nrms and triOrig in use (the bounds-checked kernel variant compares with them)
    // node array: [0, tlasReserve) the TLAS of the last upload (breadth first), then every BLAS node,
    // the top levels of all BLAS trees first (breadth first, round robin over the trees)
    std::vector<WNode> hNodes;
    size_t nBlasNodes = 0;
    int tlasReserve = 0, nTlasNodes = 0;
    std::vector<int32_t> hSlotOf;      // per instance (visiting order): TLAS node << 2 | leaf slot
    DBuf<int32_t> slotOf, dBlasRoot;
    // box bounds of the caller's trees, sorted and distinct, per tree and axis (FastScene::planes); the top-level
    // tree's part is rewritten by every object refresh behind the bottom-level part
    std::vector<float> hPlanes;
    std::vector<int32_t> hPlaneRange;
    size_t blasPlaneCount = 0;
    DBuf<float> planes;
    DBuf<int32_t> planeRange;
    // The culling guard (RT_GUARD_FACTOR): guardBlas counts, per BLAS, the hits the reference-order kernels saw come
    // near the slack since the host last looked; blasDemoted holds the host's verdict per BLAS (1: every ray entering
    // an instance of it goes to the reference-order kernel), cleared by a scene upload, kept over object refreshes.
    DBuf<int32_t> guardBlas, blasDemoted;
    std::vector<int32_t> hDemoted;
    int nBlasOfScene = 0;
    std::vector<int32_t> hSuper, hSuperOnDevice;      // the tile list of this launch / the one superTiles holds
    // The tile list goes up only when it differs from the one on the device: consecutive frames of one size share it,
    // and a frame still in flight (rtamd_render_begin) may be reading it.
    hipError_t putSuperTiles(hipStream_t stream) {
        if (superTiles.p && hSuper == hSuperOnDevice) return hipSuccess;
        if (asyncInFlight > 0) {
            // (a frame in flight reads the old list: wait for it before the list changes)
            hipError_t se = hipDeviceSynchronize();
            if (se != hipSuccess) return se;
        }
        hipError_t e = superTiles.put(hSuper, stream);
        if (e == hipSuccess) hSuperOnDevice = hSuper;
        return e;
    }
    int asyncInFlight = 0;             // frames begun with rtamd_render_begin and not yet waited for
    int nInst = 0;
    int tlasRoot = 0;
    int maxDepth = 0;
    // of the device this scene lives on (set when its context is created): compute units, and the largest
    // batch of paths whose sample buffers take a reasonable share of the device's memory
    int numCU = 0;
    size_t poolPathCap = (size_t)1 << 27;

    // who builds the tree over the instances: 0 = by instance count, 1 = host (binned SAH), 2 = device (Morton order)
    int tlasBuildMode = 0;
    bool tlasBuiltOnDevice = false;
    // who builds the trees of the meshes at upload: 0 / 1 = host (binned SAH), 2 = device (Morton order); what the last upload did
    int blasBuildMode = 0;
    int blasBuiltOnDevice = 0;
    float blasBuildMs = 0.f;
    TlasDeviceBuilder* tlasBuilder = nullptr;

    void release() {
        nodes.release(); inst.release(); shade.release(); tris.release(); nrms.release(); triOrig.release();
        superTiles.release(); stackSpill.release(); stats.release(); slotOf.release(); dBlasRoot.release(); guardBlas.release(); blasDemoted.release(); planes.release(); planeRange.release();
        tlasDeviceDestroy(tlasBuilder);
        tlasBuilder = nullptr;
    }

    // Convert the nodes of a built tree; node i of the tree gets index remap[i].
    // leafCode(start, count) encodes a leaf.
    template <typename LeafCode>
    void placeTree(const WideTree& t, const std::vector<int32_t>& remap, LeafCode leafCode) {
        for (size_t i = 0; i < t.nodes.size(); i++) {
            const WideTree::Node& n = t.nodes[i];
            WNode w;
            std::memset(&w, 0, sizeof(w));
            int32_t ch[4];
            for (int s = 0; s < 4; s++) {
                (&w.lx.x)[s] = n.lo[s][0]; (&w.hx.x)[s] = n.hi[s][0];
                (&w.ly.x)[s] = n.lo[s][1]; (&w.hy.x)[s] = n.hi[s][1];
                (&w.lz.x)[s] = n.lo[s][2]; (&w.hz.x)[s] = n.hi[s][2];
                if (n.child[s] == widebvh::kEmpty) ch[s] = kEmptyChild;     // box is all +inf: never entered
                else if (n.child[s] >= 0) ch[s] = remap[n.child[s]];
                else {
                    uint32_t code = (uint32_t)~n.child[s];
                    ch[s] = leafCode((int)(code >> 3), (int)(code & 7u));
                }
            }
            w.child = make_int4(ch[0], ch[1], ch[2], ch[3]);
            hNodes[remap[i]] = w;
        }
    }

    // breadth-first order of a tree's nodes, and where each level starts in it
    static void levelOrder(const WideTree& t, std::vector<int32_t>& order, std::vector<int32_t>& levelStart) {
        order.clear(); levelStart.clear();
        order.push_back(0);
        size_t at = 0;
        while (at < order.size()) {
            levelStart.push_back((int32_t)at);
            const size_t end = order.size();
            for (; at < end; at++)
                for (int s = 0; s < 4; s++) {
                    int32_t c = t.nodes[order[at]].child[s];
                    if (c >= 0) order.push_back(c);
                }
        }
        levelStart.push_back((int32_t)order.size());
    }

    // sort every set and drop repeated values (NaN bounds cannot equal a ray's origin and are dropped too).  On up to
    // 6 host threads, one set each; with a device builder the sets of 65 536 values and more are sorted by the GPU's
    // radix sort in between (a mesh of a million triangles has three sets of 2-3 million bounds).
    static hipError_t sortPlaneSets(std::vector<std::vector<float>>& sets, TlasDeviceBuilder* dev = nullptr, hipStream_t stream = nullptr) {
        constexpr size_t kDeviceFrom = (size_t)1 << 16;
        auto onThreads = [&sets](void (*one)(std::vector<float>&, bool), bool flag) {
            const size_t nThreads = sets.size() < 6 ? sets.size() : 6;
            std::vector<std::thread> th;
            std::atomic<size_t> next{0};
            auto work = [&] { for (size_t k; (k = next.fetch_add(1)) < sets.size();) one(sets[k], flag); };
            try {
                for (size_t t = 1; t < nThreads; t++) th.emplace_back(work);
            } catch (...) { }      // (fewer threads: the caller's thread does the rest)
            work();
            for (auto& t : th) t.join();
        };
        onThreads([](std::vector<float>& v, bool haveDevice) {
            v.erase(std::remove_if(v.begin(), v.end(), [](float x) { return x != x; }), v.end());
            if (!haveDevice || v.size() < kDeviceFrom) std::sort(v.begin(), v.end());
        }, dev != nullptr);
        if (dev)
            for (std::vector<float>& v : sets)
                if (v.size() >= kDeviceFrom) {
                    const hipError_t e = sortFloatsOnDevice(dev, v.data(), v.size(), stream);
                    if (e != hipSuccess) return e;
                }
        onThreads([](std::vector<float>& v, bool) { v.erase(std::unique(v.begin(), v.end()), v.end()); }, false);
        return hipSuccess;
    }

    // Re-build every BLAS.  Inputs were validated by the caller (indices in range, every
    // node reachable at most once).  From the caller's tree come, per triangle, its leaf-slot
    // box and its rank in the reference's visiting order (children pushed 0..3, popped last
    // first, leaf slots 0..3: RayTracingShader.compute:293-345); the hierarchy itself is new.
    hipError_t buildBottom(const int32_t* blasOffsets, const rtamd_node* blas, const int32_t* nodeCount,
                           const int32_t* vtxOffsets, const rtamd_vertex* verts, const int32_t* triOffsets,
                           const rtamd_int3* tris_, const int32_t* triCount, int nBlas, int tlasReserve_,
                           hipStream_t stream, std::string& err) {
        tlasReserve = tlasReserve_ < 1 ? 1 : tlasReserve_;
        hNodes.clear();
        std::vector<WideTree> trees((size_t)nBlas);
        std::vector<uint32_t> triBases((size_t)nBlas);
        blasRoot.assign(nBlas, 0);
        maxDepth = 0;
        blasBuiltOnDevice = 0;
        blasBuildMs = 0.f;
        std::vector<std::vector<float>> planeSets((size_t)nBlas * 3);
        // (RTAMD_VERBOSE: where the time of an upload goes)
        float phaseMs[6] = {0, 0, 0, 0, 0, 0};
        auto clockNow = [] { return std::chrono::steady_clock::now(); };
        auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        auto tPhase = clockNow();
        // Pass 1, one thread: the visiting order.  Per mesh the leaf slots that hold a triangle, in the order the
        // reference reaches them (node << 2 | slot), and the nodes visited with the number of used slots before each
        // (where the node's box bounds go in the tables of bounds).
        struct Walk {
            std::vector<uint32_t> slots;
            std::vector<int32_t> visited;
            std::vector<uint32_t> usedBefore;
            size_t rankBase = 0, usedSlots = 0;
        };
        std::vector<Walk> walks((size_t)nBlas);
        size_t nRanked = 0;
        {
            std::vector<int32_t> st;
            for (int b = 0; b < nBlas; b++) {
                if (nodeCount[b] >= (1 << 29)) { err = "too many nodes in one BLAS"; return hipSuccess; }
                const rtamd_node* nodes_ = blas + blasOffsets[b];
                Walk& wk = walks[(size_t)b];
                wk.rankBase = nRanked;
                wk.slots.reserve((size_t)triCount[b]);
                wk.visited.reserve((size_t)nodeCount[b]);
                wk.usedBefore.reserve((size_t)nodeCount[b]);
                st.clear();
                st.push_back(1);
                uint32_t used = 0;
                while (!st.empty()) {
                    const int ni = st.back();
                    st.pop_back();
                    const rtamd_node& n = nodes_[ni];
                    wk.visited.push_back(ni);
                    wk.usedBefore.push_back(used);
                    if (n.Flags == 0) {
                        for (int i = 0; i < 4; i++) if (n.Data[i] != 0) { st.push_back(n.Data[i]); used++; }
                        continue;
                    }
                    for (int i = 0; i < 4; i++)
                        if (n.Data[i] != -1) { wk.slots.push_back(((uint32_t)ni << 2) | (uint32_t)i); used++; }
                }
                wk.usedSlots = used;
                nRanked += wk.slots.size();
                if (nRanked >= (1u << 28)) { err = "too many triangles for the leaf encoding"; return hipSuccess; }
            }
        }
        // Pass 2, on up to 8 threads: the records.  Per triangle (by rank) its leaf-slot box, vertex, edges and rank
        // (`ranked`), its box for the tree builder, its normals and the caller's index; per used slot its six bounds.
        std::vector<FTri> ranked(nRanked);     // by global rank
        std::vector<FNrm> fn(nRanked);         // by global rank
        std::vector<int32_t> orig(nRanked);    // by global rank: the caller's triangle index
        std::vector<std::vector<BuildBox>> boxesOf((size_t)nBlas);
        {
            struct Chunk { int blas, kind; size_t lo, hi; };
            std::vector<Chunk> chunks;
            constexpr size_t kChunk = 8192;
            for (int b = 0; b < nBlas; b++) {
                const Walk& wk = walks[(size_t)b];
                boxesOf[(size_t)b].resize(wk.slots.size());
                for (int k = 0; k < 3; k++) planeSets[(size_t)b * 3 + k].resize(2 * wk.usedSlots);
                for (size_t lo = 0; lo < wk.slots.size(); lo += kChunk) chunks.push_back({b, 0, lo, std::min(lo + kChunk, wk.slots.size())});
                for (size_t lo = 0; lo < wk.visited.size(); lo += kChunk) chunks.push_back({b, 1, lo, std::min(lo + kChunk, wk.visited.size())});
            }
            auto fill = [&](const Chunk& c) {
                const rtamd_node* nodes_ = blas + blasOffsets[c.blas];
                const Walk& wk = walks[(size_t)c.blas];
                if (c.kind == 1) {
                    float* px = planeSets[(size_t)c.blas * 3 + 0].data();
                    float* py = planeSets[(size_t)c.blas * 3 + 1].data();
                    float* pz = planeSets[(size_t)c.blas * 3 + 2].data();
                    for (size_t v = c.lo; v < c.hi; v++) {
                        const rtamd_node& n = nodes_[wk.visited[v]];
                        size_t at = 2 * (size_t)wk.usedBefore[v];
                        for (int i = 0; i < 4; i++)
                            if (n.Flags == 0 ? n.Data[i] != 0 : n.Data[i] != -1) {
                                px[at] = n.Lx[i]; px[at + 1] = n.Hx[i];
                                py[at] = n.Ly[i]; py[at + 1] = n.Hy[i];
                                pz[at] = n.Lz[i]; pz[at + 1] = n.Hz[i];
                                at += 2;
                            }
                    }
                    return;
                }
                const rtamd_int3* tri = tris_ + triOffsets[c.blas];
                const rtamd_vertex* vtx = verts + vtxOffsets[c.blas];
                BuildBox* boxes = boxesOf[(size_t)c.blas].data();
                for (size_t r = c.lo; r < c.hi; r++) {
                    const rtamd_node& n = nodes_[wk.slots[r] >> 2];
                    const int i = (int)(wk.slots[r] & 3u);
                    const int t = n.Data[i];
                    const float* p0 = vtx[tri[t].x].position;
                    const float* p1 = vtx[tri[t].y].position;
                    const float* p2 = vtx[tri[t].z].position;
                    // e1 = v1 - v0, e2 = v2 - v0: the same binary32 subtractions Trisection does (:201-202)
                    const float e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
                    const float e2[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
                    const int32_t rank = (int32_t)(wk.rankBase + r);
                    float rk;
                    std::memcpy(&rk, &rank, 4);
                    FTri f;
                    f.a = make_float4(n.Lx[i], n.Ly[i], n.Lz[i], n.Hx[i]);
                    f.b = make_float4(n.Hy[i], n.Hz[i], rk, p0[0]);
                    f.c = make_float4(p0[1], p0[2], e1[0], e1[1]);
                    f.d = make_float4(e1[2], e2[0], e2[1], e2[2]);
                    ranked[(size_t)rank] = f;
                    boxes[r] = {{n.Lx[i], n.Ly[i], n.Lz[i]}, {n.Hx[i], n.Hy[i], n.Hz[i]}};
                    const float* n0 = vtx[tri[t].x].normal;
                    const float* n1 = vtx[tri[t].y].normal;
                    const float* n2 = vtx[tri[t].z].normal;
                    FNrm g;
                    g.a = make_float4(n0[0], n0[1], n0[2], n1[0]);
                    g.b = make_float4(n1[1], n1[2], n2[0], n2[1]);
                    g.c = make_float4(n2[2], 0.f, 0.f, 0.f);
                    fn[(size_t)rank] = g;
                    orig[(size_t)rank] = t;
                }
            };
            std::atomic<size_t> next{0};
            auto work = [&] { for (size_t k; (k = next.fetch_add(1)) < chunks.size();) fill(chunks[k]); };
            const size_t nThreads = chunks.size() < 8 ? chunks.size() : 8;
            std::vector<std::thread> th;
            try {
                for (size_t t = 1; t < nThreads; t++) th.emplace_back(work);
            } catch (...) { }      // (fewer threads: the caller's thread does the rest)
            work();
            for (auto& t : th) t.join();
        }
        phaseMs[0] = since(tPhase);
        // Per mesh: the tree, then where its leaves' records go.  srcOf[j] = rank of the triangle whose record becomes
        // record j of the leaf array (-1: the blank that fills up an odd leaf); the records themselves are gathered and
        // interleaved on the GPU (gatherLeafRecords).
        std::vector<int32_t> srcOf;
        srcOf.reserve(nRanked + nRanked / 8 + 16);
        {
            // Who builds a mesh's tree: the host's binned-SAH builder (the better tree) or, when the caller chose speed
            // of upload (blasBuildMode / RTAMD_BLAS_BUILD=device; two-triangle leaves, 256 triangles or more), the GPU.
            // The host's builds run side by side, largest mesh first, on up to widebvh::kMaxThreads threads in all
            // (a large mesh's build spreads over threads by itself; a scene of many small meshes needs this).
            int mode = blasBuildMode;
            if (const char* sv = getenv("RTAMD_BLAS_BUILD")) mode = !strcmp(sv, "device") ? 2 : (!strcmp(sv, "host") ? 1 : mode);
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<int> onHost;
            for (int b = 0; b < nBlas; b++) {
                std::vector<BuildBox>& boxes = boxesOf[(size_t)b];
                if (mode == 2 && kLeafTris == 2 && boxes.size() >= 256) {
                    if (!tlasBuilder && !(tlasBuilder = tlasDeviceCreate())) return hipErrorOutOfMemory;
                    hipError_t be = blasDeviceBuild(tlasBuilder, boxes.data(), (int)boxes.size(), trees[(size_t)b], stream);
                    if (be != hipSuccess) return be;
                    blasBuiltOnDevice++;
                } else onHost.push_back(b);
            }
            std::sort(onHost.begin(), onHost.end(), [&](int x, int y) {
                return boxesOf[(size_t)x].size() != boxesOf[(size_t)y].size() ? boxesOf[(size_t)x].size() > boxesOf[(size_t)y].size() : x < y;
            });
            std::atomic<size_t> next{0};
            std::atomic<int> atWork{0};
            std::atomic<bool> failed{false};
            auto work = [&] {
                for (size_t k; (k = next.fetch_add(1)) < onHost.size();) {
                    const int b = onHost[k];
                    try { buildWideTree(boxesOf[(size_t)b].data(), (int)boxesOf[(size_t)b].size(), kLeafTris, trees[(size_t)b], &atWork); }
                    catch (...) { failed.store(true); }
                }
            };
            std::vector<std::thread> th;
            try {
                // (a helper only where a second mesh waits and a thread is free: atWork counts this thread too)
                atWork.fetch_add(1);
                for (size_t t = 1; t < onHost.size() && t < (size_t)widebvh::kMaxThreads; t++) {
                    if (atWork.fetch_add(1) >= widebvh::kMaxThreads) { atWork.fetch_sub(1); break; }
                    th.emplace_back([&] { work(); atWork.fetch_sub(1); });
                }
            } catch (...) { atWork.fetch_sub(1); }
            work();
            for (auto& t : th) t.join();
            if (failed.load()) throw std::bad_alloc();
            blasBuildMs += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        for (int b = 0; b < nBlas; b++) {
            std::vector<BuildBox>& boxes = boxesOf[(size_t)b];
            const size_t rankBase = walks[(size_t)b].rankBase;
            WideTree& tree = trees[b];
            if (tree.depth > maxDepth) maxDepth = tree.depth;
            tPhase = clockNow();
            triBases[b] = (uint32_t)srcOf.size();
            {
                // every leaf starts a 128-byte line and is a whole number of lines (two 64-byte records each, the pair
                // interleaved): an odd leaf ends with a blank record
                const size_t nItems = boxes.size();
                std::vector<int32_t> leafAt(nItems + 1, -1);      // first item of a leaf -> its number of items
                for (const WideTree::Node& nd : tree.nodes)
                    for (int s4 = 0; s4 < 4; s4++)
                        if (nd.child[s4] < 0 && nd.child[s4] != widebvh::kEmpty) { const uint32_t code = (uint32_t)~nd.child[s4]; leafAt[code >> 3] = (int32_t)(code & 7u); }
                std::vector<int32_t> newStart(nItems + 1, 0);
                for (size_t i = 0; i < nItems;) {
                    const int cnt = leafAt[i] > 0 ? leafAt[i] : 1;
                    newStart[i] = (int32_t)(srcOf.size() - triBases[b]);
                    for (int k = 0; k < cnt; k++) srcOf.push_back((int32_t)(rankBase + (size_t)tree.order[i + (size_t)k]));
                    if (srcOf.size() & 1u) srcOf.push_back(-1);
                    i += (size_t)cnt;
                }
                for (WideTree::Node& nd : tree.nodes)
                    for (int s4 = 0; s4 < 4; s4++)
                        if (nd.child[s4] < 0 && nd.child[s4] != widebvh::kEmpty) {
                            const uint32_t code = (uint32_t)~nd.child[s4];
                            nd.child[s4] = ~(int32_t)(((uint32_t)newStart[code >> 3] << 3) | (code & 7u));
                        }
            }
            std::vector<BuildBox>().swap(boxes);
            std::vector<int32_t>().swap(tree.order);
            if (srcOf.size() >= (1u << 28)) { err = "too many triangles for the leaf encoding"; return hipSuccess; }
            phaseMs[1] += since(tPhase);
        }
        walks.clear();
        tPhase = clockNow();
        // node indices: level by level over all trees (so that a prefix of the BLAS region holds the
        // top of every tree), behind the TLAS region
        {
            std::vector<std::vector<int32_t>> order((size_t)nBlas), levelStart((size_t)nBlas), remap((size_t)nBlas);
            size_t total = 0;
            for (int b = 0; b < nBlas; b++) {
                levelOrder(trees[b], order[b], levelStart[b]);
                remap[b].assign(trees[b].nodes.size(), -1);
                total += trees[b].nodes.size();
            }
            int32_t next = tlasReserve;
            for (size_t level = 0;; level++) {
                bool any = false;
                for (int b = 0; b < nBlas; b++) {
                    if (level + 1 >= levelStart[b].size()) continue;
                    any = true;
                    for (int32_t i = levelStart[b][level]; i < levelStart[b][level + 1]; i++) remap[b][order[b][i]] = next++;
                }
                if (!any) break;
            }
            hNodes.resize((size_t)tlasReserve + total);
            for (int b = 0; b < nBlas; b++) {
                const uint32_t triBase = triBases[b];
                placeTree(trees[b], remap[b], [&](int start, int count) -> int32_t {
                    return ~(int32_t)(((triBase + (uint32_t)start) << 2) | (uint32_t)(count - 1));
                });
                blasRoot[b] = remap[b][0];
                // a mesh that fits one leaf (a quad: floors, walls) is entered at that leaf: its root node would only
                // repeat, on the union of the leaf's slot boxes, the test the slot boxes get anyway
                {
                    const WNode& rootNode = hNodes[(size_t)remap[b][0]];
                    const int32_t ch[4] = {rootNode.child.x, rootNode.child.y, rootNode.child.z, rootNode.child.w};
                    int used = 0, leaf = 0;
                    for (int s4 = 0; s4 < 4; s4++)
                        if (ch[s4] != kEmptyChild) { used++; leaf = ch[s4]; }
                    if (used == 1 && leaf < 0) blasRoot[b] = leaf;
                }
            }
        }
        nBlasNodes = hNodes.size() - (size_t)tlasReserve;
        blasUploaded = false;
        hipError_t e;
        nBlasOfScene = nBlas;
        phaseMs[2] = since(tPhase);
        tPhase = clockNow();
        {
            // the tables of box bounds: sorted and distinct per BLAS and axis
            if (!tlasBuilder && !(tlasBuilder = tlasDeviceCreate())) return hipErrorOutOfMemory;
            if ((e = sortPlaneSets(planeSets, tlasBuilder, stream)) != hipSuccess) return e;
            hPlanes.clear();
            hPlaneRange.assign(((size_t)nBlas + 1) * 6, 0);
            for (size_t k = 0; k < planeSets.size(); k++) {
                hPlaneRange[2 * k] = (int32_t)hPlanes.size();
                hPlaneRange[2 * k + 1] = (int32_t)planeSets[k].size();
                hPlanes.insert(hPlanes.end(), planeSets[k].begin(), planeSets[k].end());
            }
            blasPlaneCount = hPlanes.size();
            planesUploaded = false;
            if (blasPlaneCount >= ((size_t)1 << 31)) { err = "too many box bounds for the table of planes"; return hipSuccess; }
        }
        phaseMs[3] = since(tPhase);
        tPhase = clockNow();
        hDemoted.assign((size_t)nBlas, 0);
        if ((e = guardBlas.ensure((size_t)nBlas)) != hipSuccess || (e = blasDemoted.ensure((size_t)nBlas)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(guardBlas.p, 0, (size_t)nBlas * sizeof(int32_t), stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(blasDemoted.p, 0, (size_t)nBlas * sizeof(int32_t), stream)) != hipSuccess) return e;
        nTriRecords = (int32_t)srcOf.size(); nTriangles = (int32_t)fn.size();
        if ((e = tris.ensure(srcOf.size())) != hipSuccess) return e;
        if (!tlasBuilder && !(tlasBuilder = tlasDeviceCreate())) return hipErrorOutOfMemory;
        if ((e = gatherLeafRecords(tlasBuilder, ranked.data(), ranked.size(), srcOf.data(), srcOf.size(), tris.p, stream)) != hipSuccess) return e;
        if ((e = nrms.put(fn, stream)) != hipSuccess) return e;
        if ((e = triOrig.put(orig, stream)) != hipSuccess) return e;
        e = hipStreamSynchronize(stream);   // ranked / fn / orig die with this scope; the nodes go up with the TLAS
        phaseMs[4] = since(tPhase);
        if (getenv("RTAMD_VERBOSE"))
            fprintf(stderr, "[rtamd] upload, meshes: %d triangles in %d leaf records (%.1f %% blanks), %zu nodes\n", nTriangles, nTriRecords,
                    nTriRecords ? 100.0 * (nTriRecords - nTriangles) / nTriRecords : 0.0, nBlasNodes);
        if (getenv("RTAMD_VERBOSE"))
            fprintf(stderr, "[rtamd] upload, meshes: walk of the caller's trees %.1f ms, tree builds %.1f ms (%d on the GPU), leaf records %.1f ms, node layout %.1f ms, box-bound tables %.1f ms, copies %.1f ms\n",
                    phaseMs[0], blasBuildMs, blasBuiltOnDevice, phaseMs[1], phaseMs[2], phaseMs[3], phaseMs[4]);
        return e;
    }

    // Re-build the TLAS behind the BLAS nodes and the per-object data.  Instances are kept
    // in the reference's visiting order (IntersectTLAS, .compute:355-381: inner children
    // popped last first, leaf slots 0..3) and keep the box of their TLAS leaf slot.
    hipError_t buildTop(const rtamd_node* tlas, int nTlas, const rtamd_object_property* objects, int nObjects,
                        const rtamd_node* dTlas, const rtamd_object_property* dObjects, hipStream_t stream, std::string& err) {
        (void)nObjects; (void)nTlas; (void)objects;
        int mode = tlasBuildMode;
        if (const char* s = getenv("RTAMD_TLAS_BUILD")) mode = !strcmp(s, "device") ? 2 : (!strcmp(s, "host") ? 1 : mode);
        // where each instance sits in the caller's tree, in visiting order; the per-instance records
        // are then filled on the GPU from the caller's arrays as uploaded
        hSlotOf.clear();
        std::vector<int32_t> st;
        st.push_back(1);
        while (!st.empty()) {
            int ni = st.back();
            st.pop_back();
            const rtamd_node& n = tlas[ni];
            if (n.Flags == 0) {
                for (int i = 0; i < 4; i++) if (n.Data[i] != 0) st.push_back(n.Data[i]);
                continue;
            }
            for (int i = 0; i < 4; i++)
                if (n.Data[i] != -1) hSlotOf.push_back((int32_t)((uint32_t)ni << 2 | (uint32_t)i));
        }
        if (hSlotOf.empty()) { err = "TLAS references no object"; return hipSuccess; }
        {
            // box bounds of the caller's top-level tree (a large tree, which the device rebuilds in well under a
            // millisecond, gets no table: its zero-component rays go to the reference-order kernel as before)
            const size_t tl = (size_t)nBlasOfScene * 6;
            hPlanes.resize(blasPlaneCount);
            if (hSlotOf.size() > (size_t)kTlasDeviceMin) {
                for (int ax = 0; ax < 3; ax++) { hPlaneRange[tl + 2 * ax] = 0; hPlaneRange[tl + 2 * ax + 1] = -1; }
            } else {
                std::vector<std::vector<float>> sets(3);
                std::vector<int32_t> walk;
                walk.push_back(1);
                while (!walk.empty()) {
                    const rtamd_node& n = tlas[walk.back()];
                    walk.pop_back();
                    for (int i = 0; i < 4; i++)
                        if (n.Flags == 0 ? n.Data[i] != 0 : n.Data[i] != -1) {
                            sets[0].push_back(n.Lx[i]); sets[0].push_back(n.Hx[i]);
                            sets[1].push_back(n.Ly[i]); sets[1].push_back(n.Hy[i]);
                            sets[2].push_back(n.Lz[i]); sets[2].push_back(n.Hz[i]);
                            if (n.Flags == 0) walk.push_back(n.Data[i]);
                        }
                }
                (void)sortPlaneSets(sets);      // (on the host: cannot fail)
                for (int ax = 0; ax < 3; ax++) {
                    hPlaneRange[tl + 2 * ax] = (int32_t)hPlanes.size();
                    hPlaneRange[tl + 2 * ax + 1] = (int32_t)sets[ax].size();
                    hPlanes.insert(hPlanes.end(), sets[ax].begin(), sets[ax].end());
                }
            }
        }
        if (hSlotOf.size() >= kInstanceBit || (unsigned)nTlas >= (1u << 29)) { err = "too many instances"; return hipSuccess; }
        nInst = (int)hSlotOf.size();
        tlasRoot = 0;
        hipError_t e;
        // the device build needs room for nInst - 1 nodes whatever the tree turns out to be
        const bool onDevice = (mode == 2 || (mode == 0 && nInst >= kTlasDeviceMin)) && tlasReserve >= (nInst > 2 ? nInst - 1 : 1);
        tlasBuiltOnDevice = onDevice;
        WideTree tree;
        if (!onDevice) {
            std::vector<BuildBox> boxes((size_t)nInst);
            for (int r = 0; r < nInst; r++) {
                const rtamd_node& n = tlas[hSlotOf[r] >> 2];
                const int i = hSlotOf[r] & 3;
                boxes[r] = {{n.Lx[i], n.Ly[i], n.Lz[i]}, {n.Hx[i], n.Hy[i], n.Hz[i]}};
            }
            buildWideTree(boxes.data(), nInst, 1, tree);
            if ((int)tree.nodes.size() > tlasReserve) { err = "TLAS grew beyond the node range reserved at scene upload"; return hipSuccess; }
            std::vector<int32_t> order, levelStart, remap(tree.nodes.size());
            levelOrder(tree, order, levelStart);
            for (size_t i = 0; i < order.size(); i++) remap[order[i]] = (int32_t)i;
            placeTree(tree, remap, [&](int start, int count) -> int32_t {
                (void)count;   // one instance per leaf: the leaf's box is the instance's own TLAS slot box
                return ~(int32_t)((uint32_t)tree.order[start] | kInstanceBit);
            });
            nTlasNodes = (int)tree.nodes.size();
            tlasDepth = tree.depth;
        }
        if (!blasUploaded) {
            if ((e = nodes.put(hNodes, stream)) != hipSuccess) return e;
            if ((e = dBlasRoot.put(blasRoot, stream)) != hipSuccess) return e;
            blasUploaded = true;
        } else if (!onDevice && (e = hipMemcpyAsync(nodes.p, hNodes.data(), (size_t)nTlasNodes * sizeof(WNode), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
        {
            // the bottom-level part of the table goes up with the scene, the top-level part with every refresh
            const bool whole = planes.n < hPlanes.size() || !planesUploaded;
            if ((e = planes.ensure(hPlanes.size() + (whole ? hPlanes.size() / 8 + 1024 : 0))) != hipSuccess) return e;
            const size_t from = whole ? 0 : blasPlaneCount;
            if (hPlanes.size() > from &&
                (e = hipMemcpyAsync(planes.p + from, hPlanes.data() + from, (hPlanes.size() - from) * sizeof(float), hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
            if ((e = planeRange.put(hPlaneRange, stream)) != hipSuccess) return e;
            planesUploaded = true;
        }
        if ((e = slotOf.put(hSlotOf, stream)) != hipSuccess) return e;
        if ((e = inst.ensure((size_t)nInst)) != hipSuccess || (e = shade.ensure((size_t)nInst)) != hipSuccess) return e;
        if ((e = tlasPackInstances(dTlas, dObjects, slotOf.p, dBlasRoot.p, blasDemoted.p, nInst, inst.p, shade.p, stream)) != hipSuccess) return e;
        if (onDevice) {
            if (!tlasBuilder && !(tlasBuilder = tlasDeviceCreate())) return hipErrorOutOfMemory;
            return tlasDeviceBuild(tlasBuilder, inst.p, nInst, nodes.p, tlasReserve, stream, &nTlasNodes, &tlasDepth);
        }
        return hipSuccess;
    }
    // The host marked another mesh (hDemoted) after a render: the verdicts go up and the per-instance records are
    // written again, now with kDemotedBit on every instance of a marked mesh (the tree over the instances stays).
    hipError_t applyDemotions(const rtamd_node* dTlas, const rtamd_object_property* dObjects, hipStream_t stream) {
        hipError_t e = hipMemcpyAsync(blasDemoted.p, hDemoted.data(), hDemoted.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
        return tlasPackInstances(dTlas, dObjects, slotOf.p, dBlasRoot.p, blasDemoted.p, nInst, inst.p, shade.p, stream);
    }
    static constexpr int kTlasDeviceMin = 4096;   // instances from which the device build is used when the caller does not choose
    int tlasDepth = 0;
    bool blasUploaded = false;
    bool planesUploaded = false;
};

// ------------------------------------------------------------------ kernel ---

#ifndef RT_BLOCK_THREADS
#define RT_BLOCK_THREADS 1024
#endif
constexpr int kBlockThreads = RT_BLOCK_THREADS;  // one workgroup per CU: its 16 waves share one LDS copy of the hot nodes
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4      // waves per SIMD the register allocator must leave room for (<= 128 VGPRs)
#endif
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 16
#endif
constexpr int kLdsStack = RT_LDS_STACK;        // stack entries per lane in LDS: [entry][thread], conflict-free (64 KiB); deeper stacks overflow into HBM
                                               // (the benchmark scenes reach 16 / 22 / 27 entries; a limit of 16 measured no slower than 24)
#ifndef RT_HOT_NODES
#define RT_HOT_NODES 864
#endif
constexpr int kHotNodes = RT_HOT_NODES;       // nodes every workgroup keeps in LDS (94.5 KiB): the TLAS and the top of the BLAS trees
#ifndef RT_HOT_STRIDE
#define RT_HOT_STRIDE 7
#endif
constexpr int kHotStride = RT_HOT_STRIDE;        // float4 per node in LDS: 112 bytes = 28 banks, so piece j of 16 consecutive nodes falls on 16 different bank groups
constexpr int kModeRef = 0, kModePt = 1, kModePrimary = 2;
constexpr int kNoInstance = -(1 << 24);
constexpr int kSpillChunk = kLdsStack / 2;      // entries moved between the LDS stack and its overflow area in HBM at a time
constexpr int kSpillMax = 96;        // overflow entries per lane

typedef float f2 __attribute__((ext_vector_type(2)));

struct FastHit {
    float t, u, v;
    int32_t inst, tri;   // instance rank, triangle rank (positions in the reference's visiting order); tri | kBackFaceBit when hit from behind
};
constexpr int32_t kBackFaceBit = 1 << 30;   // (triangle ranks stay below 2^28)

// mul(M, float4(v,1)).xyz and mul(M, float4(v,0)).xyz for a matrix stored as three rows.
__device__ __forceinline__ V3 rowsPoint(float4 r0, float4 r1, float4 r2, V3 v) {
    return v3(((r0.x * v.x + r0.y * v.y) + r0.z * v.z) + r0.w,
              ((r1.x * v.x + r1.y * v.y) + r1.z * v.z) + r1.w,
              ((r2.x * v.x + r2.y * v.y) + r2.z * v.z) + r2.w);
}
__device__ __forceinline__ V3 rowsDir(float4 r0, float4 r1, float4 r2, V3 v) {
    return v3((r0.x * v.x + r0.y * v.y) + r0.z * v.z,
              (r1.x * v.x + r1.y * v.y) + r1.z * v.z,
              (r2.x * v.x + r2.y * v.y) + r2.z * v.z);
}

// The word a kernel walks for node `n` of the node array: its slot in the LDS copy when the workgroup holds it
// (the copy is [0, hotTlas) of the array followed by [blasBase, blasBase + hotBlas)), the index itself otherwise;
// leaf words (negative) pass through.
__device__ __forceinline__ int32_t hotWord(int32_t n, int hotTlas, int blasBase, int hotBlas) {
    if ((uint32_t)n < (uint32_t)hotTlas) return n | kHotNodeBit;
    if ((uint32_t)(n - blasBase) < (uint32_t)hotBlas) return (hotTlas + (n - blasBase)) | kHotNodeBit;
    return n;
}
// Copy the hot nodes into LDS (seven 16-byte pieces per node), the child words rewritten with hotWord.
__device__ __forceinline__ void copyHotNodes(float4* ldsNodes, const FastScene& sc, int hotTlas, int hotBlas, int tid) {
    for (int i = tid; i < (hotTlas + hotBlas) * 8; i += kBlockThreads) {
        const int n = i >> 3, c = i & 7;
        const int g = n < hotTlas ? n : sc.blasBase + (n - hotTlas);
        if (c < 6) ldsNodes[n * kHotStride + c] = reinterpret_cast<const float4*>(sc.nodes + g)[c];
        else if (c == 6) {
            const int4 ch = sc.nodes[g].child;
            ldsNodes[n * kHotStride + 6] = make_float4(__int_as_float(hotWord(ch.x, hotTlas, sc.blasBase, hotBlas)), __int_as_float(hotWord(ch.y, hotTlas, sc.blasBase, hotBlas)),
                                                       __int_as_float(hotWord(ch.z, hotTlas, sc.blasBase, hotBlas)), __int_as_float(hotWord(ch.w, hotTlas, sc.blasBase, hotBlas)));
        }
    }
}

// The four-box slab test (.compute:233-263), two children per packed instruction.  The
// products and differences are the same binary32 operations as the scalar form.
__device__ __forceinline__ void slabPair(f2 lx, f2 hx, f2 ly, f2 hy, f2 lz, f2 hz, f2 ox, f2 oy, f2 oz, f2 ix, f2 iy,
                                         f2 iz, float maxFraction, float& dA, float& dB) {
    f2 nxt = (lx - ox) * ix, fxt = (hx - ox) * ix;
    f2 nyt = (ly - oy) * iy, fyt = (hy - oy) * iy;
    f2 nzt = (lz - oz) * iz, fzt = (hz - oz) * iz;
    float nearA = fmax_(fmax_(fmax_(fmin_(nxt.x, fxt.x), fmin_(nyt.x, fyt.x)), fmin_(nzt.x, fzt.x)), 0.0f);
    float farA = fmin_(fmin_(fmin_(fmax_(nxt.x, fxt.x), fmax_(nyt.x, fyt.x)), fmax_(nzt.x, fzt.x)), maxFraction);
    float nearB = fmax_(fmax_(fmax_(fmin_(nxt.y, fxt.y), fmin_(nyt.y, fyt.y)), fmin_(nzt.y, fzt.y)), 0.0f);
    float farB = fmin_(fmin_(fmin_(fmax_(nxt.y, fxt.y), fmax_(nyt.y, fyt.y)), fmax_(nzt.y, fzt.y)), maxFraction);
    dA = nearA <= farA ? nearA : RT_INF;
    dB = nearB <= farB ? nearB : RT_INF;
}

__device__ __forceinline__ bool slabHit(float lo_x, float hi_x, float lo_y, float hi_y, float lo_z, float hi_z,
                                        const Ray& r, float maxFraction) {
    float nearXt = (lo_x - r.origin.x) * r.invDir.x, farXt = (hi_x - r.origin.x) * r.invDir.x;
    float nearYt = (lo_y - r.origin.y) * r.invDir.y, farYt = (hi_y - r.origin.y) * r.invDir.y;
    float nearZt = (lo_z - r.origin.z) * r.invDir.z, farZt = (hi_z - r.origin.z) * r.invDir.z;
    float nearX = fmin_(nearXt, farXt), farX = fmax_(nearXt, farXt);
    float nearY = fmin_(nearYt, farYt), farY = fmax_(nearYt, farYt);
    float nearZ = fmin_(nearZt, farZt), farZ = fmax_(nearZt, farZt);
    float nearMax = fmax_(fmax_(fmax_(nearX, nearY), nearZ), 0.0f);
    float farMin = fmin_(fmin_(fmin_(farX, farY), farZ), maxFraction);
    return nearMax <= farMin;
}

// Entry distance of the same test (its nearMax).
__device__ __forceinline__ float slabNear(float lo_x, float hi_x, float lo_y, float hi_y, float lo_z, float hi_z, const Ray& r) {
    float nearXt = (lo_x - r.origin.x) * r.invDir.x, farXt = (hi_x - r.origin.x) * r.invDir.x;
    float nearYt = (lo_y - r.origin.y) * r.invDir.y, farYt = (hi_y - r.origin.y) * r.invDir.y;
    float nearZt = (lo_z - r.origin.z) * r.invDir.z, farZt = (hi_z - r.origin.z) * r.invDir.z;
    return fmax_(fmax_(fmax_(fmin_(nearXt, farXt), fmin_(nearYt, farYt)), fmin_(nearZt, farZt)), 0.0f);
}

// Boxes are culled against the closest hit so far times this factor (see the file header).
#ifndef RT_CULL_RELAX
#define RT_CULL_RELAX 1.00390625f   /* 1 + 2^-8 */
#endif
// The guard of that slack.  Culling against best.t * RT_CULL_RELAX can only hide a triangle whose own boxes (leaf
// slot box, instance box) are entered, as computed, more than 2^-8 (relative) behind the triangle's own computed hit
// distance: in exact arithmetic a box is never entered behind what it holds, so this is rounding error of the two
// computations, a few ulps on ordinary meshes and growing with the aspect of slivers.  Every hit the production
// kernels compute is checked: entered more than 2^-10 behind its distance (a quarter of the slack), the ray is
// handed to the reference-order kernel (four instructions per hit, no branch).  The reference-order kernels make the
// same check on the hits they compute, count such hits and mark the hit's mesh; the host then lets every ray that
// enters an instance of a marked mesh go to the reference-order kernel (kDemotedBit) and renders the frame again
// (rtamd.hip, rtamd_last_guard_stats).
// (RT_GUARD_FACTOR itself is defined in kernel_common.h: both families of kernels use it)

__device__ __forceinline__ void cswap(float& da, int32_t& ca, float& db, int32_t& cb) {
    bool sw = db < da;
    float td = sw ? db : da; float tb = sw ? da : db;
    int32_t tc = sw ? cb : ca; int32_t te = sw ? ca : cb;
    da = td; db = tb; ca = tc; cb = te;
}

// A reciprocal direction component that is not finite (the direction component is 0, or
// so small that 1/d overflows) can turn a slab product into 0 * inf = NaN.
__device__ __forceinline__ bool rayDegenerate(const Ray& r) {
    return !(__builtin_fabsf(r.invDir.x) < RT_INF) || !(__builtin_fabsf(r.invDir.y) < RT_INF) || !(__builtin_fabsf(r.invDir.z) < RT_INF);
}

// A ray with a zero direction component (a non-finite reciprocal) is traversed like any other as long as its origin
// does not lie exactly on a box bound of that axis: every slab product of the axis is then -inf or +inf, which min
// and max treat like any number, in the caller's trees and in the rebuilt ones alike.  Only a bound equal to the
// origin's coordinate gives 0 * inf = NaN, which min / max drop, and then the caller's own inner boxes decide
// (RayTracingShader.compute:233-263).  The tables hold every bound of the caller's tree `tree` (a BLAS index, or
// sc.planeTlas), sorted: one binary search per degenerate axis.  False: hand the ray to the reference-order kernel.
__device__ __forceinline__ bool planesClear(const FastScene& sc, int tree, const Ray& r) {
    const float o[3] = {r.origin.x, r.origin.y, r.origin.z}, inv[3] = {r.invDir.x, r.invDir.y, r.invDir.z};
    bool clear = true;
#pragma unroll 1
    for (int ax = 0; ax < 3; ax++) {
        if (__builtin_fabsf(inv[ax]) < RT_INF) continue;
        const int first = sc.planeRange[(tree * 3 + ax) * 2], count = sc.planeRange[(tree * 3 + ax) * 2 + 1];
        if (count < 0 || !(o[ax] == o[ax])) { clear = false; continue; }
        int lo = 0, hi = count;        // first value >= o[ax]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sc.planes[first + mid] < o[ax]) lo = mid + 1; else hi = mid;
        }
        if (lo < count && sc.planes[first + lo] == o[ax]) clear = false;
    }
    return clear;
}

// Position, shading normal and material of the closest hit (.compute:329-339).
__device__ __forceinline__ Surf resolveHit(const FastScene& sc, const Ray& world, const FastHit& h) {
    const FInst fi = sc.inst[h.inst];
    const FShade fs = sc.shade[h.inst];
    Ray local = makeRay(rowsPoint(fi.r0, fi.r1, fi.r2, world.origin), rowsDir(fi.r0, fi.r1, fi.r2, world.direction));
    Surf s;
    s.position = rowsPoint(fs.r0, fs.r1, fs.r2, local.origin + h.t * local.direction);
    const int32_t tri = h.tri & ~kBackFaceBit;
    const FNrm g = sc.nrms[tri];
    V3 n0 = v3(g.a.x, g.a.y, g.a.z), n1 = v3(g.a.w, g.b.x, g.b.y), n2 = v3(g.b.z, g.b.w, g.c.x);
    float w = 1 - h.u - h.v;
    V3 nl = ((h.u * n1) + (h.v * n2)) + (w * n0);
    s.normal = normalize(rowsDir(fs.r0, fs.r1, fs.r2, nl));
    s.albedo = v3(fs.albedo.x, fs.albedo.y, fs.albedo.z);
    s.specular = v3(fs.specular.x, fs.specular.y, fs.specular.z);
    s.emission = v3(fs.emission.x, fs.emission.y, fs.emission.z);
    s.albedoW = fs.albedo.w; s.specularW = fs.specular.w;
    s.back = (h.tri & kBackFaceBit) != 0;
    if (s.back) s.normal = v3(-s.normal.x, -s.normal.y, -s.normal.z);
    s.object = fi.object & ~kDemotedBit;
    s.triangle = sc.triOrig[tri];
    return s;
}

// Persistent kernel.  Each lane owns one pixel at a time and walks all of its samples
// and bounces, so a pixel's sum is formed in a fixed order.  Three levels of regrouping,
// all driven by wave ballots:
//   * lanes whose ray has finished wait until fewer than `refillBelow` lanes are still
//     traversing, then shade / start new rays / fetch new pixels together;
//   * inside traversal, lanes that reached a leaf (triangles or an instance) wait while at
//     least `nodeMin` lanes still have inner nodes to test, then all leaves are handled
//     together (a per-round vote between node / triangle / instance steps measured 8 % slower),
//     except that a small group of instance entries next to a large group of triangle leaves (or
//     the reverse) waits for a later leaf phase;
//   * the wave draws pixels from one global counter, 8x8 pixel tiles first.
// The traversal stack lives in LDS ([entry][thread], one bank per lane) and overflows in chunks
// into HBM.  A pixel with a ray that could meet 0 * inf in a slab test, with an order-ambiguous
// hit, or with a stack beyond LDS plus the overflow area is dropped unfinished and listed for the
// reference-order kernel, which the host runs over that list afterwards.
// STATS = true builds a diagnostic variant that also counts, per phase, how often the wave
// executed it and with how many active lanes (written to `stats`, 16 counters).
// DIEL = true (PT mode with RTAMD_FLAG_DIELECTRICS): glass objects refract; a path inside glass tests
// triangles from both sides.
// CHECKED = true (RTAMD_FLAG_DEBUG_CHECKS): the diagnostic variant compares every node, instance, triangle and stack
// index with the scene's sizes before following it; an index outside ends the pixel and sets bit 8 of the error flag.
template <int MODE, bool STATS, bool DIEL = false, bool CHECKED = false>
__global__ __launch_bounds__(kBlockThreads, RT_MIN_WAVES) void kFast(rtamd_frame_cbuffer camera /* first: see cameraArg */, FastScene sc, Sky sky, RenderArgs a, int refillBelow, int nodeMin, int leafGroupMin,
                                                       int stackLimit, unsigned long long* stats) {
    unsigned long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define RT_STAT2(slot, cond) do { } while (0)
#define RT_STAT(slot, cond)                                                         \
    do {                                                                            \
        if (STATS) {                                                                \
            unsigned long long m_ = __ballot(cond);                                 \
            if (m_) { st_[2 * (slot)] += 1; st_[2 * (slot) + 1] += __popcll(m_); }  \
        }                                                                           \
    } while (0)
    __shared__ int32_t ldsStack[kLdsStack][kBlockThreads];
    __shared__ float4 ldsNodes[kHotNodes * kHotStride];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const unsigned long long totalWork = (unsigned long long)sc.nSuperTiles * 1024ull;
    // An inner-node test costs seven 16-byte loads per lane, each lane in its own 128-byte line, a
    // gather the vector cache serves one lane per cycle.  The nodes every ray passes through — the
    // TLAS and the first levels of the BLAS trees — are therefore read from an LDS copy.
    const int hotTlas = sc.hotTlas, hotBlas = sc.hotBlas;
    copyHotNodes(ldsNodes, sc, hotTlas, hotBlas, tid);
    __syncthreads();
    const int32_t rootWord = hotWord(sc.tlasRoot, hotTlas, sc.blasBase, hotBlas);

    // pixel state
    bool pixelActive = false, noMoreWork = false;
    int px = 0, py = 0;
    int sample = 0;
    V3 sum = v3(0, 0, 0);
    // path state
    bool pathActive = false;
    uint32_t key = 0;
    int bounce = 0;
    V3 energy = v3(1, 1, 1), L = v3(0, 0, 0);
    bool insideGlass = false;     // DIEL: the path is travelling inside a dielectric
    // traversal state
    bool trav = false;
    Ray world = makeRay(v3(0, 0, 0), v3(0, 0, 1));
    Ray cur = world;
    FastHit best;
    best.t = RT_INF; best.u = best.v = 0; best.inst = -1; best.tri = -1;
    float secondT = RT_INF;   // closest hit among the triangles that are not `best`
    float bestNear = 0.0f;    // entry distance of best's slot box and instance box, whichever is larger
    float instNear = 0.0f;    // entry distance of the current instance's TLAS slot box
    int32_t node = 0;
    int sp = 0;
    int instSp = kNoInstance; // stack height on entering the current instance
    int spilled = 0;          // entries of this lane's stack that were moved out to HBM (a multiple of kSpillChunk)
    int32_t curInst = -1;
    uint32_t rays = 0, pixRays = 0;   // ray segments of finished pixels / of the current pixel

    int deepest = 0;          // STATS: largest stack height of this lane
#define RT_WORLD_RAY() world
#include "traverse_macros.inc"

    // Hand the current pixel to the reference-order kernel: mark it, list it once, drop it.
#define RT_REDO()                                                                   \
    do {                                                                            \
        const uint32_t p_ = (uint32_t)py * (uint32_t)a.width + (uint32_t)px;        \
        const uint32_t bit_ = 1u << (p_ & 31u);                                     \
        const uint32_t old_ = atomicOr(a.redoMask + (p_ >> 5), bit_);               \
        if (!(old_ & bit_)) {                                                       \
            const int i_ = atomicAdd(a.redoCount, 1);                               \
            if (i_ < a.redoCap) a.redoList[i_] = (int32_t)p_;                       \
            else atomicOr(a.errorFlag, 4);                                          \
        }                                                                           \
        pixelActive = false; pathActive = false; trav = false; pixRays = 0;         \
    } while (0)

    for (;;) {
        // ------------------------------------------------ phase A: lanes without a ray
        RT_STAT(0, !trav);
        if (!trav) {
            bool haveRay = false;
            V3 nextO = world.origin, nextD = world.direction;    // the reciprocal direction is formed once, below
            // another triangle was hit before the winner's own boxes are entered: the reference's answer
            // depends on its visiting order
            if (pathActive && best.t < RT_INF && secondT < bestNear) RT_REDO();
            if (CHECKED && pathActive && best.t < RT_INF && !((uint32_t)best.inst < (uint32_t)sc.nInst && (uint32_t)(best.tri & ~kBackFaceBit) < (uint32_t)sc.nTriangles)) {
                atomicOr(a.errorFlag, 8);
                pixelActive = false; pathActive = false; pixRays = 0;
            }
            if (pathActive) {
                // the segment just traversed
                if (best.t < RT_INF) {
                    Surf s = resolveHit(sc, world, best);
                    if (MODE == kModePt) {
                        L = L + energy * s.emission;
                        V3 org, dir;
                        if (bounce < a.maxBounces && scatterSurface<DIEL>(s, world.direction, key, (uint32_t)bounce, energy, org, dir, insideGlass)) {
                            nextO = org; nextD = dir;
                            bounce++;
                            haveRay = true;
                        }
                    } else {
                        size_t pix = (size_t)py * a.width + px;
                        if (bounce == 0) {
                            if (a.outIds) { a.outIds[2 * pix] = s.object; a.outIds[2 * pix + 1] = s.triangle; }
                            if (a.outT) a.outT[pix] = best.t;
                            if (MODE == kModeRef) {
                                // Trace: mirror ray from the hit position, no offset (.compute:419)
                                nextO = s.position; nextD = reflect(world.direction, s.normal);
                                bounce = 1;
                                haveRay = true;
                            }
                        } else {
                            L = v3(0, 0, 0) + v3(1, 1, 1) * s.emission;      // Shade, hit branch (:400)
                            if (a.outRgba) a.outRgba[pix] = make_float4(L.x, L.y, L.z, 1.0f);
                        }
                    }
                } else {
                    if (MODE == kModePt) {
                        L = L + energy * shadeMiss(sky, world.direction);
                    } else {
                        size_t pix = (size_t)py * a.width + px;
                        if (bounce == 0) {
                            if (a.outIds) { a.outIds[2 * pix] = -1; a.outIds[2 * pix + 1] = -1; }
                            if (a.outT) a.outT[pix] = RT_INF;
                            if (MODE == kModeRef && a.outRgba) a.outRgba[pix] = make_float4(0, 0, 0, 0);   // .compute:427
                        } else {
                            L = v3(0, 0, 0) + v3(1, 1, 1) * shadeMiss(sky, world.direction);
                            if (a.outRgba) a.outRgba[pix] = make_float4(L.x, L.y, L.z, 1.0f);
                        }
                    }
                }
                if (!haveRay) {
                    pathActive = false;
                    if (MODE == kModePt) { sum = sum + L; sample++; }
                    else sample = 1;
                }
            }
            if (!haveRay && pixelActive && sample >= (MODE == kModePt ? a.spp : 1)) {
                if (MODE == kModePt) writePathPixel(a, (size_t)py * a.width + px, sum);
                pixelActive = false;
                rays += pixRays;
                pixRays = 0;
            }
            // fetch pixels for lanes that have none: one atomic per wave and round, shared by ballot
            for (;;) {
                bool need = !haveRay && !pixelActive && !noMoreWork;
                unsigned long long m = __ballot(need);
                if (m == 0ull) break;
                int leader = __ffsll((long long)m) - 1;
                unsigned long long base = 0;
                if (lane == leader) base = atomicAdd(a.rayCount + 1, (unsigned long long)__popcll(m));
                base = __shfl(base, leader, 64);
                if (need) {
                    unsigned long long w = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
                    if (w >= totalWork) noMoreWork = true;
                    else {
                        int st = sc.superTiles[(int)(w >> 10)];       // tile column | tile row << 16
                        int within = (int)(w & 1023ull);
                        int sx = st & 0xffff, sy = st >> 16;
                        int t8 = within >> 6, p8 = within & 63;
                        int x = sx * 32 + (t8 & 3) * 8 + (p8 & 7);
                        int y = sy * 32 + (t8 >> 2) * 8 + (p8 >> 3);
                        if (x < a.width && y < a.height && x >= a.tileX && y >= a.tileY && x < a.tileX + a.tileW && y < a.tileY + a.tileH) {
                            const uint32_t p = (uint32_t)y * (uint32_t)a.width + (uint32_t)x;
                            // (a pixel an earlier pass of this render already handed over is skipped)
                            if (!((a.redoMask[p >> 5] >> (p & 31u)) & 1u)) { px = x; py = y; pixelActive = true; sample = 0; sum = v3(0, 0, 0); pixRays = 0; }
                        }
                    }
                }
            }
            if (!haveRay && pixelActive) {
                // start the next sample of this pixel
                (void)camera;
                const CameraArgPtr camPtr = cameraArg();
                float jx = camPtr->PixelOffset[0], jy = camPtr->PixelOffset[1];
                if (MODE == kModePt) {
                    key = rngKey(a.seedHi, a.seedLo, (uint32_t)((size_t)py * a.width + px), (uint32_t)(a.sampleOffset + sample));
                    if (!(a.flags & RTAMD_FLAG_NO_JITTER)) { jx = rngDraw(key, 0u); jy = rngDraw(key, 1u); }
                }
                const Ray cam = cameraRayArg(camPtr, pixelToNdc(px, jx, a.width), pixelToNdc(py, jy, a.height), 0.0f);
                nextO = cam.origin; nextD = cam.direction;
                energy = v3(1, 1, 1); L = v3(0, 0, 0); bounce = 0;
                insideGlass = false;
                pathActive = true;
                haveRay = true;
            }
            if (haveRay) {
                const Ray next = makeRay(nextO, nextD);
                world = next; cur = next;
                best.t = RT_INF; best.inst = -1; best.tri = -1;
                secondT = RT_INF; bestNear = 0.0f;
                node = rootWord; sp = 0; instSp = kNoInstance; curInst = -1; spilled = 0;
                trav = true;
                pixRays++;
                // (coherent rays -- REF mode, the primary-id pass -- meet exact zeros as a matter of course: the centre
                // column and row of every un-jittered frame of an axis-aligned camera; they are settled against the tables
                // of box bounds.  Path-traced rays practically never do, and go to the reference-order kernel.)
                if (rayDegenerate(next) && (MODE == kModePt || !planesClear(sc, sc.planeTlas, next))) RT_REDO();
            }
        }
        if (__ballot(trav) == 0ull) break;   // nobody got a ray: every lane is out of work

        // ------------------------------------------------ phase B: traversal
#define RT_CAN_REFILL (!trav && !(noMoreWork && !pixelActive && !pathActive))
#define RT_TWO_SIDED (DIEL && insideGlass)
#define RT_KEEP_HIT(t_, u_, v_, tri_) do { best.u = (u_); best.v = (v_); } while (0)
#define RT_LOCAL_RAY_UNSETTLED(rank_) (rayDegenerate(cur) && (MODE == kModePt || !planesClear(sc, __float_as_int(sc.inst[rank_].pad.x), cur)))
#define RT_BAD_INDEX() do { atomicOr(a.errorFlag, 8); pixelActive = false; pathActive = false; trav = false; pixRays = 0; } while (0)
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
    if (STATS) {
        for (int off = 32; off > 0; off >>= 1) { int o = __shfl_xor(deepest, off, 64); deepest = o > deepest ? o : deepest; }
        if (lane == 0) {
            for (int i = 0; i < 15; i++) if (st_[i]) atomicAdd(stats + i, st_[i]);
            atomicMax(stats + 15, (unsigned long long)deepest);
        }
    }
    unsigned long long r = rays;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) r += __shfl_xor(r, off, 64);
    if (lane == 0 && r) atomicAdd(a.rayCount, r);
}

// How a budget of LDS node slots is shared: the top-level tree gets what the bottom-level trees leave, and at least
// its share of the budget (RTAMD_HOT_TLAS_SHARE, percent) as far as it has nodes.
inline void hotNodeRanges(const FastSceneHost& fs, int budget, int32_t& hotTlas, int32_t& hotBlas) {
    static const int share = [] { const char* s = getenv("RTAMD_HOT_TLAS_SHARE"); const int v = s ? atoi(s) : 50; return v < 0 ? 0 : (v > 100 ? 100 : v); }();
    const int nb = fs.nBlasNodes < (size_t)budget ? (int)fs.nBlasNodes : budget;
    int tl = budget - nb > budget * share / 100 ? budget - nb : budget * share / 100;
    if (tl > fs.nTlasNodes) tl = fs.nTlasNodes;
    hotTlas = tl;
    hotBlas = budget - tl < nb ? budget - tl : nb;
}

struct FastLaunchConfig {
    // refillBelow / nodeMin = 0: chosen per scene in launchFast
    int blocksPerCU = 1, refillBelow = 0, nodeMin = 0, leafGroupMin = 20;
    bool pool = true;                           // PT mode through kPool (more paths than lanes); false: kFast<PT>
    int stackLimit = kLdsStack;                 // LDS stack entries a lane may use before it moves a chunk to HBM (tests lower it)
    bool stats = false;
};

inline FastLaunchConfig fastLaunchConfig() {
    FastLaunchConfig c;
    if (const char* s = getenv("RTAMD_BLOCKS_PER_CU")) { int v = atoi(s); if (v > 0 && v <= 2) c.blocksPerCU = v; }
    if (const char* s = getenv("RTAMD_REFILL_BELOW")) { int v = atoi(s); if (v >= 1 && v <= 64) c.refillBelow = v; }
    if (const char* s = getenv("RTAMD_NODE_MIN")) { int v = atoi(s); if (v >= 1 && v <= 64) c.nodeMin = v; }
    if (const char* s = getenv("RTAMD_LEAF_GROUP_MIN")) { int v = atoi(s); if (v >= 0 && v <= 64) c.leafGroupMin = v; }
    if (const char* s = getenv("RTAMD_STACK_LIMIT")) { int v = atoi(s); if (v >= kSpillChunk + 4 && v <= kLdsStack) c.stackLimit = v; }
    if (const char* s = getenv("RTAMD_PIPELINE")) c.pool = strcmp(s, "persistent") != 0;
    if (const char* s = getenv("RTAMD_STATS")) c.stats = atoi(s) != 0;
    return c;
}

// Regrouping thresholds.  With every inner node in LDS a node step waits for nothing, and small
// batches cost little: shade early (refillBelow 32), keep stepping nodes with few lanes (nodeMin 8).
// A scene whose lower levels live in L2 / HBM pays a memory round trip per node step, which larger
// batches amortise (48 / 16).  Measured optima of the spheres, mesh and instanced benchmark scenes.
// Camera rays and their mirror rays (REF mode, the primary-id pass) are coherent: the lanes of a wave finish
// together, and waiting for more of them to finish before shading and refilling only lengthens the chain of
// dependent steps a frame of one or two segments per pixel consists of.  There a lane is served as soon as it is
// done and node steps go on down to 4 lanes: 1080p REF frames 1.76 -> 1.05 ms (spheres), 1.56 -> 0.82 ms (1M
// triangles), 4.1 -> 2.0 ms at 3840x2160 against the path-tracing thresholds.
inline void chooseThresholds(FastLaunchConfig& cfg, const FastSceneHost& fs, int mode) {
    const bool allHot = fs.nBlasNodes + (size_t)fs.nTlasNodes <= (size_t)kHotNodes;
    if (mode != kModePt) {
        if (cfg.refillBelow == 0) cfg.refillBelow = 1;
        if (cfg.nodeMin == 0) cfg.nodeMin = 4;
        return;
    }
    if (cfg.refillBelow == 0) cfg.refillBelow = allHot ? 32 : 48;
    if (cfg.nodeMin == 0) cfg.nodeMin = allHot ? 8 : 16;
}

// Device memory the caller lends to one launch in place of the context's growable buffers (the
// allocation-free entry point): room for the tile list and a fixed stack overflow area, shared out
// among the lanes of the grid (16 entries per lane on a 256-CU part: stacks of up to 32 entries, the
// deepest of the benchmark scenes is 27; a deeper one hands its pixel to the reference-order kernel).
struct LaunchScratch {
    int32_t* superTiles;
    int32_t superCap;
    int32_t* stackSpill;
    size_t spillWords;
};

inline hipError_t launchFast(FastSceneHost& fs, const Sky& sky, const RenderArgs& a, hipStream_t stream, const LaunchScratch* lent = nullptr) {
    // 32x32 tiles intersecting the rectangle and owned by this shard
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
    hipError_t e;
    if (lent) {
        if (fs.hSuper.size() > (size_t)lent->superCap) return hipErrorInvalidValue;
        e = hipMemcpyAsync(lent->superTiles, fs.hSuper.data(), fs.hSuper.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream);
    } else e = fs.putSuperTiles(stream);
    if (e != hipSuccess) return e;

    const int numCU = fs.numCU;
    if (numCU <= 0) return hipErrorInvalidDevice;
    FastLaunchConfig cfg = fastLaunchConfig();
    chooseThresholds(cfg, fs, a.mode == RTAMD_MODE_PT ? kModePt : kModeRef);
    FastLaunchConfig primaryCfg = fastLaunchConfig();
    chooseThresholds(primaryCfg, fs, kModePrimary);
    long long wavesNeeded = ((long long)fs.hSuper.size() * 1024 + 63) / 64;
    long long blocksNeeded = (wavesNeeded + kBlockThreads / 64 - 1) / (kBlockThreads / 64);
    int grid = numCU * cfg.blocksPerCU;
    if (blocksNeeded < grid) grid = (int)blocksNeeded;
    if (grid < 1) grid = 1;

    FastScene sc;
    sc.nodes = fs.nodes.p; sc.inst = fs.inst.p; sc.shade = fs.shade.p;
    sc.tris = fs.tris.p; sc.nrms = fs.nrms.p; sc.triOrig = fs.triOrig.p; sc.nSuperTiles = (int)fs.hSuper.size();
    sc.tlasRoot = fs.tlasRoot;
    sc.planes = fs.planes.p; sc.planeRange = fs.planeRange.p; sc.planeTlas = fs.nBlasOfScene;
    sc.nNodes = (int32_t)fs.hNodes.size(); sc.nInst = fs.nInst; sc.nTriRecords = fs.nTriRecords; sc.nTriangles = fs.nTriangles;
    if (lent) {
        sc.superTiles = lent->superTiles;
        // (the kernel indexes the area by gridDim.x: whole chunks per lane of the grid actually launched)
        const size_t perLane = lent->spillWords / ((size_t)grid * kBlockThreads);
        sc.spillMax = (int32_t)(perLane > (size_t)kSpillMax ? (size_t)kSpillMax : perLane) / kSpillChunk * kSpillChunk;
        sc.stackSpill = sc.spillMax > 0 ? lent->stackSpill : nullptr;
    }
    else {
        sc.superTiles = fs.superTiles.p;
        if ((e = fs.stackSpill.ensure((size_t)kSpillMax * (size_t)(numCU * cfg.blocksPerCU) * kBlockThreads)) != hipSuccess) return e;
        sc.stackSpill = fs.stackSpill.p;
        sc.spillMax = kSpillMax;
    }
    // LDS budget: the whole TLAS if it leaves half of the room (or more, when the BLAS trees are small) to the BLAS tops
    {
        hotNodeRanges(fs, kHotNodes, sc.hotTlas, sc.hotBlas);
        sc.blasBase = fs.tlasReserve;
    }

    if (a.mode == RTAMD_MODE_PT && (a.outIds || a.outT)) {
        // identity of the un-jittered primary hit: a separate, uncounted pass
        RenderArgs ids = a;
        ids.outRgba = nullptr; ids.accum = nullptr; ids.spp = 1;
        hipLaunchKernelGGL((kFast<kModePrimary, false>), dim3(grid), dim3(kBlockThreads), 0, stream, ids.cb, sc, sky, ids, primaryCfg.refillBelow, primaryCfg.nodeMin, cfg.leafGroupMin, cfg.stackLimit, (unsigned long long*)nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        // the work-queue head and ray counter are shared: reset them between the passes
        if ((e = hipMemsetAsync(a.rayCount, 0, 2 * sizeof(unsigned long long), stream)) != hipSuccess) return e;
    }
    if (cfg.stats && a.mode == RTAMD_MODE_PT && !lent && !(a.flags & RTAMD_FLAG_DIELECTRICS)) {
        // diagnostic build: per-phase execution counts and active lanes, printed to stderr
        if ((e = fs.stats.ensure(16)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(fs.stats.p, 0, 16 * sizeof(unsigned long long), stream)) != hipSuccess) return e;
        hipLaunchKernelGGL((kFast<kModePt, true>), dim3(grid), dim3(kBlockThreads), 0, stream, a.cb, sc, sky, a, cfg.refillBelow, cfg.nodeMin, cfg.leafGroupMin, cfg.stackLimit, fs.stats.p);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        unsigned long long h[16];
        if ((e = hipMemcpyAsync(h, fs.stats.p, sizeof(h), hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        const char* names[8] = {"phaseA", "nodeStep", "instEnter", "triLeaf", "triTest", "-", "leafPhase(trav lanes)", "nodeLoop(trav lanes)"};
        for (int i = 0; i < 8; i++)
            if (h[2 * i]) fprintf(stderr, "[rtamd stats] %-22s executions %12llu  mean active lanes %6.2f\n", names[i], h[2 * i], (double)h[2 * i + 1] / (double)h[2 * i]);
        fprintf(stderr, "[rtamd stats] deepest traversal stack %llu of %d entries\n", h[15], kLdsStack);
        return hipSuccess;
    }
    if (a.flags & RTAMD_FLAG_DEBUG_CHECKS) {
        // the bounds-checked diagnostic variant (glass objects are not part of it: RTAMD_FLAG_DIELECTRICS is refused with it)
        // Test hook RTAMD_TEST_DAMAGE = node | inst | tri: for this launch only (the word is put back in stream order
        // behind the kernel) one child word of the scene's tree names a node, an instance or a triangle range that
        // does not exist.  Only here, where every index is compared before it is followed.
        int32_t* damaged = nullptr;
        int32_t saved = 0;
        if (const char* what = getenv("RTAMD_TEST_DAMAGE")) {
            int32_t word = 0;
            if (!strcmp(what, "node") && !fs.blasRoot.empty() && fs.blasRoot[0] >= 0) { damaged = &fs.nodes.p[fs.blasRoot[0]].child.x; word = 0x0FFFFFFF; }
            else if (!strcmp(what, "tri") && !fs.blasRoot.empty() && fs.blasRoot[0] >= 0) { damaged = &fs.nodes.p[fs.blasRoot[0]].child.x; word = ~(int32_t)((0x00FFFFFFu << 2) | 1u); }
            else if (!strcmp(what, "inst")) { damaged = &fs.nodes.p[sc.tlasRoot].child.x; word = ~(int32_t)(0x00FFFFFFu | kInstanceBit); }
            if (damaged) {
                hipError_t e = hipMemcpy(&saved, damaged, sizeof saved, hipMemcpyDeviceToHost);
                if (e == hipSuccess) e = hipMemcpyAsync(damaged, &word, sizeof word, hipMemcpyHostToDevice, stream);
                if (e == hipSuccess) e = hipStreamSynchronize(stream);
                if (e != hipSuccess) return e;
            }
        }
        if (a.mode == RTAMD_MODE_PT)
            hipLaunchKernelGGL((kFast<kModePt, false, false, true>), dim3(grid), dim3(kBlockThreads), 0, stream, a.cb, sc, sky, a, cfg.refillBelow, cfg.nodeMin, cfg.leafGroupMin, cfg.stackLimit, (unsigned long long*)nullptr);
        else
            hipLaunchKernelGGL((kFast<kModeRef, false, false, true>), dim3(grid), dim3(kBlockThreads), 0, stream, a.cb, sc, sky, a, cfg.refillBelow, cfg.nodeMin, cfg.leafGroupMin, cfg.stackLimit, (unsigned long long*)nullptr);
        hipError_t le = hipGetLastError();
        if (damaged) {
            hipError_t e = hipStreamSynchronize(stream);
            if (e == hipSuccess) e = hipMemcpy(damaged, &saved, sizeof saved, hipMemcpyHostToDevice);
            if (le == hipSuccess) le = e;
        }
        return le;
    }
    if (a.mode == RTAMD_MODE_PT && (a.flags & RTAMD_FLAG_DIELECTRICS))
        hipLaunchKernelGGL((kFast<kModePt, false, true>), dim3(grid), dim3(kBlockThreads), 0, stream, a.cb, sc, sky, a, cfg.refillBelow, cfg.nodeMin, cfg.leafGroupMin, cfg.stackLimit, (unsigned long long*)nullptr);
    else if (a.mode == RTAMD_MODE_PT)
        hipLaunchKernelGGL((kFast<kModePt, false>), dim3(grid), dim3(kBlockThreads), 0, stream, a.cb, sc, sky, a, cfg.refillBelow, cfg.nodeMin, cfg.leafGroupMin, cfg.stackLimit, (unsigned long long*)nullptr);
    else
        hipLaunchKernelGGL((kFast<kModeRef, false>), dim3(grid), dim3(kBlockThreads), 0, stream, a.cb, sc, sky, a, cfg.refillBelow, cfg.nodeMin, cfg.leafGroupMin, cfg.stackLimit, (unsigned long long*)nullptr);
    return hipGetLastError();
}

}  // namespace rt

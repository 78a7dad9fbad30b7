// 4-wide BVH construction for the GPU layout (host code, runs at scene upload).
//
// The library does not traverse the caller's trees in the production kernel: the
// reference's median-split builder (Assets/Script/BVH/BoundingVolumeHierarchyBuilder.cs,
// with its left-size miscount at :298) yields trees that cost several times more node
// visits per ray than a surface-area-heuristic tree over the same primitives.  What must
// be kept from the caller's tree is only what decides hits: each primitive's own leaf-slot
// box and the order in which the reference would visit primitives (for ties).  Every
// inner box built here is the union of the leaf-slot boxes below it, so a primitive whose
// own box passes the reference's slab test also passes every box above it (the slab
// arithmetic is monotone in the box bounds as long as no 0 * inf occurs; rays for which it
// can occur are handed to the reference-order kernel).
//
// Builder: top-down binned SAH (32 bins, three axes) on box centroids; a node takes the
// range, splits it, then keeps splitting the part with the largest box until it has four parts
// (or every part fits a leaf; area * count as the criterion measured 1-2 % slower to trace).  Large subtrees are built by threads of their own
// (disjoint item ranges, a private node array spliced into the parent's afterwards); the tree does
// not depend on how many threads ran.
#pragma once

#include <algorithm>
#include <atomic>
#include <new>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace rt {

struct BuildBox { float lo[3], hi[3]; };

struct WideTree {
    struct Node { float lo[4][3], hi[4][3]; int32_t child[4]; };   // child: >= 0 node; < 0: ~((start << 3) | count); INT32_MIN empty
    std::vector<Node> nodes;            // node 0 is the root
    std::vector<int32_t> order;         // position in leaf order -> input item
    int depth = 0;
};

namespace widebvh {

constexpr int32_t kEmpty = INT32_MIN;
constexpr int kBins = 32;

struct Item { float c[3]; int32_t index; float lo[3], hi[3]; };   // centroid, input item, and the item's box (carried along: the scans below read memory in order)

inline float halfArea(const float lo[3], const float hi[3]) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0) || !(dy >= 0) || !(dz >= 0)) return 0.f;
    return dx * dy + dy * dz + dz * dx;
}

constexpr int kSpawnMin = 16384;     // items from which a subtree is worth a thread
constexpr int kMaxThreads = 8;

struct Builder {
    const BuildBox* boxes;
    Item* items;
    WideTree* out;
    int maxLeaf;
    int sweepMax = 0;                // ranges up to this many items are split by an exact sweep instead of bins
    int expPick = 1, expCost = 0;    // which part a node splits next: 1 the largest box (default), 0 area x count, 2 count; cost per item instead of per leaf
    std::atomic<int>* threads;       // builder threads at work, shared by the whole build (one that waits for its own helpers does not count)
    bool counted = false;            // this builder runs on a thread that `threads` counts
    int spawnLimit = kMaxThreads - 1; // helpers may start while `threads` is below this (the calling thread of a build on its own is not counted)
    std::atomic<bool>* failed;       // a builder thread ran out of memory: the caller throws std::bad_alloc after all have joined

    void bounds(int lo, int hi, float blo[3], float bhi[3]) const {
        for (int k = 0; k < 3; k++) { blo[k] = __builtin_inff(); bhi[k] = -__builtin_inff(); }
        for (int i = lo; i < hi; i++) {
            const Item& b = items[i];
            for (int k = 0; k < 3; k++) { blo[k] = std::min(blo[k], b.lo[k]); bhi[k] = std::max(bhi[k], b.hi[k]); }
        }
    }

    // Exact SAH over a small range: the items sorted along each axis, every position tried.
    int splitSweep(int lo, int hi) {
        const int n = hi - lo;
        float bestCost = __builtin_inff();
        int bestAxis = -1, bestAt = -1;
        std::vector<float> rArea((size_t)n + 1);
        for (int axis = 0; axis < 3; axis++) {
            std::stable_sort(items + lo, items + hi, [axis](const Item& a, const Item& b) { return a.c[axis] < b.c[axis]; });
            float alo[3], ahi[3];
            for (int k = 0; k < 3; k++) { alo[k] = __builtin_inff(); ahi[k] = -__builtin_inff(); }
            for (int i = n - 1; i > 0; i--) {
                const Item& bx = items[lo + i];
                for (int k = 0; k < 3; k++) { alo[k] = std::min(alo[k], bx.lo[k]); ahi[k] = std::max(ahi[k], bx.hi[k]); }
                rArea[(size_t)i] = halfArea(alo, ahi);
            }
            for (int k = 0; k < 3; k++) { alo[k] = __builtin_inff(); ahi[k] = -__builtin_inff(); }
            for (int i = 1; i < n; i++) {
                const Item& bx = items[lo + i - 1];
                for (int k = 0; k < 3; k++) { alo[k] = std::min(alo[k], bx.lo[k]); ahi[k] = std::max(ahi[k], bx.hi[k]); }
                const float cost = halfArea(alo, ahi) * (float)((i + maxLeaf - 1) / maxLeaf) + rArea[(size_t)i] * (float)((n - i + maxLeaf - 1) / maxLeaf);
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestAt = i; }
            }
        }
        if (bestAxis < 0) return lo + n / 2;
        if (bestAxis != 2) std::stable_sort(items + lo, items + hi, [bestAxis](const Item& a, const Item& b) { return a.c[bestAxis] < b.c[bestAxis]; });
        return lo + bestAt;
    }

    // split [lo,hi) in two; returns mid with lo < mid < hi
    int split(int lo, int hi) {
        const int n = hi - lo;
        if (n <= sweepMax) return splitSweep(lo, hi);
        float clo[3], chi[3];
        for (int k = 0; k < 3; k++) { clo[k] = __builtin_inff(); chi[k] = -__builtin_inff(); }
        for (int i = lo; i < hi; i++)
            for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], items[i].c[k]); chi[k] = std::max(chi[k], items[i].c[k]); }
        float bestCost = __builtin_inff();
        int bestAxis = -1, bestBin = -1;
        // The bins of the three axes in one pass over the items.  A bin is set up when its first item arrives and the
        // sweeps below visit only bins that hold items (most ranges are small: a few items in 32 bins); an empty bin
        // would only repeat the split of the bin before it, at the same cost.
        static_assert(kBins == 32, "one bit per bin");
        float blo[3][kBins][3], bhi[3][kBins][3], scale[3];
        int cnt[3][kBins];
        uint32_t filled[3] = {0u, 0u, 0u};
        for (int axis = 0; axis < 3; axis++) scale[axis] = chi[axis] > clo[axis] ? kBins / (chi[axis] - clo[axis]) : 0.f;
        for (int i = lo; i < hi; i++) {
            const Item& bx = items[i];
            for (int axis = 0; axis < 3; axis++) {
                if (!(chi[axis] > clo[axis])) continue;
                int b = (int)((bx.c[axis] - clo[axis]) * scale[axis]);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                if (!(filled[axis] >> b & 1u)) {
                    filled[axis] |= 1u << b;
                    cnt[axis][b] = 0;
                    for (int k = 0; k < 3; k++) { blo[axis][b][k] = __builtin_inff(); bhi[axis][b][k] = -__builtin_inff(); }
                }
                for (int k = 0; k < 3; k++) { blo[axis][b][k] = std::min(blo[axis][b][k], bx.lo[k]); bhi[axis][b][k] = std::max(bhi[axis][b][k], bx.hi[k]); }
                cnt[axis][b]++;
            }
        }
        for (int axis = 0; axis < 3; axis++) {
            if (!(chi[axis] > clo[axis])) continue;
            float rArea[kBins];      // at a filled bin: area and count of the items in this bin and the bins above
            int rCnt[kBins];
            float alo[3], ahi[3];
            for (int k = 0; k < 3; k++) { alo[k] = __builtin_inff(); ahi[k] = -__builtin_inff(); }
            int c = 0;
            for (uint32_t m = filled[axis]; m;) {
                const int b = 31 - __builtin_clz(m);
                m &= ~(1u << b);
                for (int k = 0; k < 3; k++) { alo[k] = std::min(alo[k], blo[axis][b][k]); ahi[k] = std::max(ahi[k], bhi[axis][b][k]); }
                c += cnt[axis][b];
                rArea[b] = halfArea(alo, ahi);
                rCnt[b] = c;
            }
            for (int k = 0; k < 3; k++) { alo[k] = __builtin_inff(); ahi[k] = -__builtin_inff(); }
            c = 0;
            for (uint32_t m = filled[axis]; m;) {
                const int b = __builtin_ctz(m);
                m &= m - 1;
                if (!m) break;      // the highest filled bin: nothing would be right of the split
                const int above = __builtin_ctz(m);
                for (int k = 0; k < 3; k++) { alo[k] = std::min(alo[k], blo[axis][b][k]); ahi[k] = std::max(ahi[k], bhi[axis][b][k]); }
                c += cnt[axis][b];
                // cost in leaf-sized groups: a leaf holds up to maxLeaf primitives
                float cost = expCost ? halfArea(alo, ahi) * (float)c + rArea[above] * (float)rCnt[above]
                                     : halfArea(alo, ahi) * (float)((c + maxLeaf - 1) / maxLeaf) +
                                       rArea[above] * (float)((rCnt[above] + maxLeaf - 1) / maxLeaf);
                if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = b; }
            }
        }
        if (bestAxis < 0) return lo + n / 2;     // all centroids coincide
        const float cmin = clo[bestAxis];
        const float bestScale = scale[bestAxis];
        auto it = std::partition(items + lo, items + hi, [&](const Item& p) {
            int b = (int)((p.c[bestAxis] - cmin) * bestScale);
            b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
            return b <= bestBin;
        });
        int mid = (int)(it - items);
        if (mid == lo || mid == hi) mid = lo + n / 2;
        return mid;
    }

    // fills node `ni` for items [lo,hi), hi - lo > maxLeaf (or the root)
    void buildNode(int ni, int lo, int hi, int depth) {
        if (depth > out->depth) out->depth = depth;
        struct Seg { int lo, hi; float blo[3], bhi[3]; };      // a part of the range and the box of its items
        Seg segs[4];
        int ns = 1;
        segs[0].lo = lo; segs[0].hi = hi;
        bounds(lo, hi, segs[0].blo, segs[0].bhi);
        while (ns < 4) {
            int pick = -1;
            float best = -1.f;
            for (int s = 0; s < ns; s++) {
                int n = segs[s].hi - segs[s].lo;
                if (n <= maxLeaf) continue;
                float score = expPick == 1 ? halfArea(segs[s].blo, segs[s].bhi) : (expPick == 2 ? (float)n : halfArea(segs[s].blo, segs[s].bhi) * (float)n);
                if (score > best) { best = score; pick = s; }
            }
            if (pick < 0) break;
            int mid = split(segs[pick].lo, segs[pick].hi);
            Seg a, b;
            a.lo = segs[pick].lo; a.hi = mid; b.lo = mid; b.hi = segs[pick].hi;
            bounds(a.lo, a.hi, a.blo, a.bhi);
            bounds(b.lo, b.hi, b.blo, b.bhi);
            for (int s = ns; s > pick + 1; s--) segs[s] = segs[s - 1];
            segs[pick] = a;
            segs[pick + 1] = b;
            ns++;
        }
        int32_t child[4] = {kEmpty, kEmpty, kEmpty, kEmpty};
        int childNode[4] = {-1, -1, -1, -1};
        for (int s = 0; s < ns; s++) {
            int n = segs[s].hi - segs[s].lo;
            if (n <= maxLeaf) child[s] = ~(int32_t)(((uint32_t)segs[s].lo << 3) | (uint32_t)n);
            else {
                childNode[s] = (int)out->nodes.size();
                child[s] = childNode[s];
                out->nodes.emplace_back();
            }
        }
        {
            WideTree::Node& nd = out->nodes[ni];
            for (int s = 0; s < 4; s++) {
                nd.child[s] = child[s];
                if (s < ns) for (int k = 0; k < 3; k++) { nd.lo[s][k] = segs[s].blo[k]; nd.hi[s][k] = segs[s].bhi[k]; }
                else for (int k = 0; k < 3; k++) { nd.lo[s][k] = __builtin_inff(); nd.hi[s][k] = __builtin_inff(); }
            }
        }
        // subtrees: the large ones on threads of their own while this thread does the rest
        struct Job { std::thread th; WideTree tree; int child; };
        Job jobs[4];
        int nJobs = 0;
        for (int s = 0; s < ns; s++) {
            if (childNode[s] < 0) continue;
            const int n = segs[s].hi - segs[s].lo;
            if (n >= kSpawnMin && threads->fetch_add(1) < spawnLimit) {
                Job& j = jobs[nJobs++];
                j.child = childNode[s];
                j.tree.nodes.emplace_back();
                const int jlo = segs[s].lo, jhi = segs[s].hi;
                WideTree* jt = &j.tree;
                try {
                    j.th = std::thread([this, jt, jlo, jhi, depth]() {
                        // (an exception must not leave a thread: it would end the process)
                        try {
                            Builder sub = *this;
                            sub.out = jt;
                            sub.counted = true;
                            sub.buildNode(0, jlo, jhi, depth + 1);
                        } catch (...) { failed->store(true); }
                        threads->fetch_sub(1);
                    });
                } catch (...) {
                    // no thread to be had: build this subtree here after all
                    threads->fetch_sub(1);
                    nJobs--;
                    j.tree.nodes.clear();
                    try { buildNode(childNode[s], jlo, jhi, depth + 1); } catch (...) { failed->store(true); }
                }
            } else {
                if (n >= kSpawnMin) threads->fetch_sub(1);
                try { buildNode(childNode[s], segs[s].lo, segs[s].hi, depth + 1); } catch (...) { failed->store(true); }
            }
        }
        // every started thread is joined before anything can leave this frame; while this thread waits its place is free
        if (nJobs > 0 && counted) threads->fetch_sub(1);
        for (int k = 0; k < nJobs; k++) jobs[k].th.join();
        if (nJobs > 0 && counted) threads->fetch_add(1);
        if (failed->load()) return;
        for (int k = 0; k < nJobs; k++) {
            // node 0 of the private array is the child itself, the others are appended
            const WideTree& t = jobs[k].tree;
            const int32_t base = (int32_t)out->nodes.size() - 1;
            out->nodes.resize(out->nodes.size() + t.nodes.size() - 1);
            for (size_t i = 0; i < t.nodes.size(); i++) {
                WideTree::Node nd = t.nodes[i];
                for (int s = 0; s < 4; s++)
                    if (nd.child[s] > 0) nd.child[s] += base;
                out->nodes[i == 0 ? (size_t)jobs[k].child : (size_t)base + i] = nd;
            }
            if (t.depth > out->depth) out->depth = t.depth;
        }
    }
};

}  // namespace widebvh

// Build a 4-wide tree over `n` boxes with at most `maxLeaf` (1..7) items per leaf.  `sharedThreads` (may be null):
// the count of builder threads at work that several builds running side by side share, so that together they stay
// within kMaxThreads (the calling thread of each build counts as one of them).
inline void buildWideTree(const BuildBox* boxes, int n, int maxLeaf, WideTree& out, std::atomic<int>* sharedThreads = nullptr) {
    out.nodes.clear();
    out.order.clear();
    out.depth = 0;
    widebvh::Builder b;
    std::vector<widebvh::Item> items((size_t)n);
    std::atomic<int> threads(0);
    std::atomic<bool> failed(false);
    b.failed = &failed;
    b.boxes = boxes;
    b.items = items.data();
    b.out = &out;
    b.maxLeaf = maxLeaf;
    if (const char* sv = getenv("RTAMD_EXP_SWEEP")) b.sweepMax = atoi(sv);
    if (const char* sv = getenv("RTAMD_EXP_PICK")) b.expPick = atoi(sv);
    if (const char* sv = getenv("RTAMD_EXP_COST")) b.expCost = atoi(sv);
    b.threads = sharedThreads ? sharedThreads : &threads;
    b.counted = sharedThreads != nullptr;
    if (sharedThreads) b.spawnLimit = widebvh::kMaxThreads;
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) { items[i].c[k] = (boxes[i].lo[k] + boxes[i].hi[k]) * 0.5f; items[i].lo[k] = boxes[i].lo[k]; items[i].hi[k] = boxes[i].hi[k]; }
        items[i].index = i;
    }
    out.nodes.emplace_back();
    if (n > 0) b.buildNode(0, 0, n, 1);
    else {
        WideTree::Node& nd = out.nodes[0];
        for (int s = 0; s < 4; s++) {
            nd.child[s] = widebvh::kEmpty;
            for (int k = 0; k < 3; k++) { nd.lo[s][k] = __builtin_inff(); nd.hi[s][k] = __builtin_inff(); }
        }
    }
    if (failed.load()) throw std::bad_alloc();
    out.order.resize(n);
    for (int i = 0; i < n; i++) out.order[i] = items[i].index;
}

}  // namespace rt

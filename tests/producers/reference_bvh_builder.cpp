// 4-wide BVH builder: the host-side producer of the TLAS / BLAS node arrays.
//
// Restates Assets/Script/BVH/BoundingVolumeHierarchyBuilder.cs (median-split mode,
// the only mode any reference caller enables: useSah defaults to false at :529 and is
// hard-wired false at :660 and :738), including the behaviours that make the
// reference's trees what they are:
//   * Segregate computes the left size from the base of the WHOLE point array
//     (:270,:298), not from range.Start;
//   * IndexOfMaxComponent is IndexOfMinComponent(-v) (Math.cs:44,53) — ties go to
//     the later axis;
//   * ranges of <= 32 points are insertion-sorted on one axis and cut into runs of
//     four (:359-394);
//   * bounds come only from Refit (:560-602), bottom-up over node indices.
//
// TEST / FIXTURE INFRASTRUCTURE, not product code: a function-by-function C++ restatement of the
// reference's C# builder, kept so that scenes can be produced without Unity exactly as the reference
// would produce them.  The builders the library itself uses are raytracing_amd/csrc/wide_bvh.h
// (binned SAH) and raytracing_amd/csrc/tlas_build.hip (GPU LBVH).

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <vector>

#include "host_internal.h"

namespace rtamd_host {

namespace {

struct float3 { float x, y, z; };
struct Aabb { float3 Min, Max; };

inline float comp(const float3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
inline float3 min3(float3 a, float3 b) { return {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)}; }
inline float3 max3(float3 a, float3 b) { return {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)}; }

const Aabb kEmptyAabb = {{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};   // Aabb.cs:21

inline void include(Aabb& a, float3 p) { a.Min = min3(a.Min, p); a.Max = max3(a.Max, p); }  // Aabb.cs:39-43

// Math.cs:44 and :53
inline int indexOfMinComponent(float3 v) { return v.x < v.y ? (v.x < v.z ? 0 : 2) : (v.y < v.z ? 1 : 2); }
inline int indexOfMaxComponent(float3 v) { return indexOfMinComponent({-v.x, -v.y, -v.z}); }

inline float3 pos(const rtamd_point_and_index& p) { return {p.Position[0], p.Position[1], p.Position[2]}; }

void setNodeEmpty(rtamd_node& n) {                                   // Node.Empty, :38-43
    for (int i = 0; i < 4; i++) {
        n.Lx[i] = n.Ly[i] = n.Lz[i] = FLT_MAX;
        n.Hx[i] = n.Hy[i] = n.Hz[i] = -FLT_MAX;
        n.Data[i] = 0;
    }
    n.Flags = 0;
}

inline void setAabb(rtamd_node& n, int i, const Aabb& a) {           // FourTransposedAabbs.cs:32-42
    n.Lx[i] = a.Min.x; n.Hx[i] = a.Max.x;
    n.Ly[i] = a.Min.y; n.Hy[i] = a.Max.y;
    n.Lz[i] = a.Min.z; n.Hz[i] = a.Max.z;
}

inline float cmin4(const float* v) { return std::fmin(std::fmin(v[0], v[1]), std::fmin(v[2], v[3])); }
inline float cmax4(const float* v) { return std::fmax(std::fmax(v[0], v[1]), std::fmax(v[2], v[3])); }

inline Aabb compoundAabb(const rtamd_node& n) {                      // FourTransposedAabbs.cs:60-64
    return {{cmin4(n.Lx), cmin4(n.Ly), cmin4(n.Lz)}, {cmax4(n.Hx), cmax4(n.Hy), cmax4(n.Hz)}};
}

void setAabbFromPoints(Aabb& aabb, const rtamd_point_and_index* pts, int length) {   // :749-758
    aabb = kEmptyAabb;
    for (int i = 0; i < length; i++) include(aabb, pos(pts[i]));
}

struct Range {                                                       // :101-117
    int Start = 0, Length = 0, Root = 0;
    Aabb Domain = kEmptyAabb;
    int Depth = 0;
};

constexpr int kMaxTreeDepth = 63;          // Constants, :79-85
constexpr int kMaxNumTreeBranches = 64;
constexpr int kSmallRangeSize = 32;
constexpr int kUnaryStackSize = 256;

struct Builder {                                                     // :96-480
    rtamd_node* nodes;
    int capacity;
    rtamd_point_and_index* points;
    int freeNodeIndex;
    bool failed = false;

    rtamd_node* getNode(int i) {
        if (i < 0 || i >= capacity) { failed = true; return &scratch; }
        return nodes + i;
    }
    rtamd_node scratch;

    void sortRange(int axis, const Range& range) {                   // :119-133
        for (int i = range.Start; i < range.Start + range.Length; ++i) {
            rtamd_point_and_index value = points[i];
            float key = value.Position[axis];
            int j = i;
            while (j > range.Start && key < points[j - 1].Position[axis]) {
                points[j] = points[j - 1];
                j--;
            }
            points[j] = value;
        }
    }

    static void computeAxisAndPivot(const Range& range, int& axis, float& pivot) {   // :141-146
        float3 ext = {range.Domain.Max.x - range.Domain.Min.x, range.Domain.Max.y - range.Domain.Min.y,
                      range.Domain.Max.z - range.Domain.Min.z};
        axis = indexOfMaxComponent(ext);
        float3 mid = {(range.Domain.Min.x + range.Domain.Max.x) / 2, (range.Domain.Min.y + range.Domain.Max.y) / 2,
                      (range.Domain.Min.z + range.Domain.Max.z) / 2};
        pivot = comp(mid, axis);
    }

    static void splitRange(const Range& range, int size, Range& l, Range& r) {       // :148-154
        l.Start = range.Start;
        l.Length = size;
        r.Start = l.Start + l.Length;
        r.Length = range.Length - l.Length;
    }

    void segregate(int axis, float pivot, Range range, int minItems, Range& lRange, Range& rRange) {  // :263-315
        Aabb lDomain = kEmptyAabb, rDomain = kEmptyAabb;
        rtamd_point_and_index* p = points;
        rtamd_point_and_index* start = p + range.Start;
        rtamd_point_and_index* end = start + range.Length - 1;
        for (;;) {
            while (start <= end && start->Position[axis] < pivot) include(lDomain, pos(*(start++)));
            while (end > start && end->Position[axis] >= pivot) include(rDomain, pos(*(end--)));
            if (start >= end) break;
            include(lDomain, pos(*end));
            include(rDomain, pos(*start));
            std::swap(*(start++), *(end--));
        }
        int lSize = (int)(start - p);           // measured from the array base, as the reference does (:298)
        int rSize = range.Length - lSize;
        if (lSize < minItems || rSize < minItems) {
            splitRange(range, range.Length / 2, lRange, rRange);
            setAabbFromPoints(lDomain, points + lRange.Start, lRange.Length);
            setAabbFromPoints(rDomain, points + rRange.Start, rRange.Length);
        } else {
            splitRange(range, lSize, lRange, rRange);
        }
        lRange.Domain = lDomain;
        rRange.Domain = rDomain;
    }

    void createChildren(Range* subRanges, int numSubRanges, int parentNodeIndex, Range* rangeStack, int& stackSize,
                        int stackCapacity) {                                          // :317-353
        int parentData[4] = {0, 0, 0, 0};
        for (int i = 0; i < numSubRanges; i++) {
            int childNodeIndex = freeNodeIndex++;
            parentData[i] = childNodeIndex;
            if (subRanges[i].Length > 4) {
                if (stackSize >= stackCapacity) { failed = true; return; }
                rangeStack[stackSize] = subRanges[i];
                rangeStack[stackSize++].Root = childNodeIndex;
            } else {
                rtamd_node* child = getNode(childNodeIndex);
                child->Flags = 1;
                for (int k = 0; k < subRanges[i].Length; k++) child->Data[k] = points[subRanges[i].Start + k].Index;
                for (int j = subRanges[i].Length; j < 4; j++) child->Data[j] = -1;
            }
        }
        rtamd_node* parent = getNode(parentNodeIndex);
        std::memcpy(parent->Data, parentData, sizeof(parentData));
        parent->Flags = 0;
    }

    void processSmallRange(Range baseRange) {                         // :359-394
        Range range = baseRange;
        int axis; float pivot;
        computeAxisAndPivot(range, axis, pivot);
        sortRange(axis, range);
        Range subRanges[4];
        int hasLeftOvers = 1;
        do {
            int numSubRanges = 0;
            while (range.Length > 4 && numSubRanges < 3) {
                subRanges[numSubRanges].Start = range.Start;
                subRanges[numSubRanges].Length = 4;
                numSubRanges++;
                range.Start += 4;
                range.Length -= 4;
            }
            if (range.Length > 0) {
                subRanges[numSubRanges].Start = range.Start;
                subRanges[numSubRanges].Length = range.Length;
                numSubRanges++;
            }
            hasLeftOvers = 0;
            // the one-slot "stack" is `range` itself: a remainder of more than four points
            // replaces it (with Root = its freshly allocated node) and the loop runs again
            int parent = range.Root;
            createChildren(subRanges, numSubRanges, parent, &range, hasLeftOvers, 1);
            if (failed) return;
        } while (hasLeftOvers > 0);
    }

    void processLargeRange(const Range& range, Range* subRanges) {    // :396-421
        for (int i = 0; i < 4; i++) subRanges[i].Depth = range.Depth + 1;
        int axis; float pivot;
        computeAxisAndPivot(range, axis, pivot);
        Range temps[2];
        segregate(axis, pivot, range, 2, temps[0], temps[1]);
        int lAxis; float lPivot;
        computeAxisAndPivot(temps[0], lAxis, lPivot);
        segregate(lAxis, lPivot, temps[0], 1, subRanges[0], subRanges[1]);
        int rAxis; float rPivot;
        computeAxisAndPivot(temps[1], rAxis, rPivot);
        segregate(rAxis, rPivot, temps[1], 1, subRanges[2], subRanges[3]);
    }

    void createInternalNodes(Range* subRanges, int numSubRanges, int root, Range* rangeStack, int& stackSize,
                             int& freeIdx) {                                          // :423-437
        int rootData[4] = {0, 0, 0, 0};
        for (int i = 0; i < numSubRanges; ++i) {
            rootData[i] = freeIdx++;
            rangeStack[stackSize] = subRanges[i];
            rangeStack[stackSize++].Root = rootData[i];
        }
        rtamd_node* rootNode = getNode(root);
        std::memcpy(rootNode->Data, rootData, sizeof(rootData));
        rootNode->Flags = 0;
    }

    void build(const Range& baseRange) {                              // :439-467
        std::vector<Range> ranges(kUnaryStackSize);
        int rangeStackSize = 1;
        ranges[0] = baseRange;
        Range subRanges[4];
        if (baseRange.Length > 4) {
            do {
                Range range = ranges[--rangeStackSize];
                if (range.Depth <= kMaxTreeDepth && range.Length > kSmallRangeSize) {
                    processLargeRange(range, subRanges);
                    createChildren(subRanges, 4, range.Root, ranges.data(), rangeStackSize, kUnaryStackSize);
                } else {
                    processSmallRange(range);
                }
                if (failed) return;
            } while (rangeStackSize > 0);
        } else {
            createChildren(ranges.data(), 1, baseRange.Root, ranges.data(), rangeStackSize, kUnaryStackSize);
        }
    }
};

void refit(rtamd_node* nodes, const rtamd_aabb* aabbs, int nodeStartIndex, int nodeEndIndex) {   // :560-602
    for (int i = nodeEndIndex; i >= nodeStartIndex; i--) {
        rtamd_node* cur = nodes + i;
        if (cur->Flags != 0) {
            for (int j = 0; j < 4; ++j) {
                Aabb aabb = kEmptyAabb;
                if (cur->Data[j] != -1) {
                    const rtamd_aabb& a = aabbs[cur->Data[j]];
                    aabb = {{a.Min[0], a.Min[1], a.Min[2]}, {a.Max[0], a.Max[1], a.Max[2]}};
                }
                setAabb(*cur, j, aabb);
            }
        } else {
            for (int j = 0; j < 4; j++) {
                Aabb aabb = kEmptyAabb;
                if (cur->Data[j] != 0) aabb = compoundAabb(nodes[cur->Data[j]]);
                setAabb(*cur, j, aabb);
            }
        }
    }
}

void refitNode(rtamd_node* nodes, int nodeIndex) {                    // :604-616
    rtamd_node* cur = nodes + nodeIndex;
    for (int j = 0; j < 4; j++) setAabb(*cur, j, compoundAabb(nodes[cur->Data[j]]));
}

}  // namespace

// BoundingVolumeHierarchy.Build, :529-558
int bvhBuild(rtamd_point_and_index* points, const rtamd_aabb* aabbs, int n, rtamd_node* nodes, int capacity,
             int* nodeCount) {
    if (capacity < 2 || n < 0) return RTAMD_ERR_INVALID_ARG;
    setNodeEmpty(nodes[0]);
    if (n > 0) {
        Builder b;
        b.nodes = nodes; b.capacity = capacity; b.points = points; b.freeNodeIndex = 2;
        Range base;
        base.Start = 0; base.Length = n; base.Root = 1; base.Depth = 0;
        setAabbFromPoints(base.Domain, points, n);
        b.build(base);
        if (b.failed || b.freeNodeIndex > capacity) return RTAMD_ERR_BAD_INDEX;
        *nodeCount = b.freeNodeIndex;
        refit(nodes, aabbs, 1, b.freeNodeIndex - 1);
    } else {
        setNodeEmpty(nodes[1]);
        *nodeCount = 2;
    }
    return RTAMD_OK;
}

// ScheduleBuildJobs, :482-527: BuildFirstNLevels (:642-728), BuildBranch per branch
// (:731-746), FinalizeTreeJob (:823-841).
int bvhBuildJobs(rtamd_point_and_index* points, const rtamd_aabb* aabbs, int n, rtamd_node* nodes, int capacity,
                 int* branchCountOut) {
    if (capacity < 2 || n <= 0) return RTAMD_ERR_INVALID_ARG;
    // The reference keeps each level in a 64-entry stack array (:647-648) and appends ranges it does not
    // split without a bound check (:690), so a level can outgrow the array.  The arrays here are large
    // enough to notice that (a level of 60 can grow by 4 plus at most the 60 ranges of the level before)
    // and the build is then refused instead of overrunning them.
    Range levelA[2 * kMaxNumTreeBranches + 8], levelB[2 * kMaxNumTreeBranches + 8];
    Range* level0 = levelA;
    Range* level1 = levelB;
    int level0Size = 1, level1Size = 0;

    Range root;
    root.Start = 0; root.Length = n; root.Root = 1; root.Depth = 0;
    setAabbFromPoints(root.Domain, points, n);
    level0[0] = root;
    int largestRangeInLastLevel = 0;
    int minTreeNodeLevel;
    const int maxNumBranchesMinusOneSplit = kMaxNumTreeBranches - 4;
    int freeNodeIndex = 2;

    Builder builder;
    builder.nodes = nodes; builder.capacity = capacity; builder.points = points; builder.freeNodeIndex = 0;

    Range subRanges[4];
    bool anySplit;
    do {
        minTreeNodeLevel = kMaxTreeDepth;
        anySplit = false;
        for (int i = 0; i < level0Size; ++i) {
            if (level0[i].Depth <= kMaxTreeDepth && level0[i].Length > kSmallRangeSize &&
                level1Size <= maxNumBranchesMinusOneSplit) {
                anySplit = true;
                builder.processLargeRange(level0[i], subRanges);
                builder.createInternalNodes(subRanges, 4, level0[i].Root, level1, level1Size, freeNodeIndex);
                for (int k = 0; k < 4; k++) largestRangeInLastLevel = std::max(largestRangeInLastLevel, subRanges[k].Length);
                minTreeNodeLevel = std::min(minTreeNodeLevel, level0[i].Depth + 1);
            } else {
                level1[level1Size++] = level0[i];
                largestRangeInLastLevel = std::max(largestRangeInLastLevel, level0[i].Length);
                minTreeNodeLevel = std::min(minTreeNodeLevel, level0[i].Depth);
            }
        }
        if (level1Size > kMaxNumTreeBranches) {
            setLastError("the reference's first-levels pass overruns its 64-entry range arrays for this input (BoundingVolumeHierarchyBuilder.cs:647-690)");
            return RTAMD_ERR_BAD_INDEX;
        }
        std::swap(level0, level1);
        level0Size = level1Size;
        level1Size = 0;
        // The reference never resets largestRangeInLastLevel (:655,:678-681,:688), so for roughly
        // 129..900 primitives its loop condition stays true after the last possible split and
        // the loop spins forever on an unchanging state.  Leaving once a pass splits nothing
        // yields exactly that fixed state.
    } while (anySplit && level0Size <= maxNumBranchesMinusOneSplit && largestRangeInLastLevel > kSmallRangeSize &&
             minTreeNodeLevel <= kMaxTreeDepth);
    if (builder.failed) return RTAMD_ERR_BAD_INDEX;

    struct RangeSizeAndIndex { int RangeIndex, RangeSize, RangeFirstNodeOffset; };
    RangeSizeAndIndex rangeMap[kMaxNumTreeBranches + 4];
    int nodeOffset = freeNodeIndex;
    for (int i = 0; i < level0Size; i++) {
        rangeMap[i] = {i, level0[i].Length, nodeOffset};
        nodeOffset += level0[i].Length;
    }
    // SortRangeMap (:625-640): insertion sort, descending size, stable
    for (int i = 0; i < level0Size; i++) {
        RangeSizeAndIndex value = rangeMap[i];
        int key = rangeMap[i].RangeSize;
        int j = i;
        while (j > 0 && key > rangeMap[j - 1].RangeSize) { rangeMap[j] = rangeMap[j - 1]; j--; }
        rangeMap[j] = value;
    }
    setNodeEmpty(nodes[0]);

    int branchCount = level0Size;
    int minBranchNodeIndex = 0;
    for (int i = 0; i < branchCount; i++) {
        const Range& range = level0[rangeMap[i].RangeIndex];
        int firstNodeIndex = rangeMap[i].RangeFirstNodeOffset;
        Builder b;
        b.nodes = nodes; b.capacity = capacity; b.points = points; b.freeNodeIndex = firstNodeIndex;
        b.build(range);
        if (b.failed || b.freeNodeIndex > capacity) return RTAMD_ERR_BAD_INDEX;
        // A branch owns as many nodes as it has primitives (:502-510).  With many coincident points
        // (every instance of a mesh carries the mesh's own centre, BuildAccelerateStructureSystem.cs:
        // 212-214) the halving fallback can need more, and the reference's branch job then writes into
        // its neighbour's nodes.  Reported instead of returning a corrupt tree.
        if (b.freeNodeIndex > firstNodeIndex + range.Length) {
            setLastError("a TLAS branch needs more nodes than the reference reserves for it (too many instances share one centre)");
            return RTAMD_ERR_BAD_INDEX;
        }
        refit(nodes, aabbs, firstNodeIndex, b.freeNodeIndex - 1);
        refitNode(nodes, range.Root);
        minBranchNodeIndex = i == 0 ? firstNodeIndex - 1 : std::min(firstNodeIndex - 1, minBranchNodeIndex);
    }
    refit(nodes, aabbs, 1, minBranchNodeIndex);
    if (branchCountOut) *branchCountOut = branchCount;
    return RTAMD_OK;
}

}  // namespace rtamd_host

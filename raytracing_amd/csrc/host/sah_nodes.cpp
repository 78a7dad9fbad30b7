// Binned-SAH tree in the reference's node format (BoundingVolumeHierarchy.Node,
// Assets/Script/BVH/BoundingVolumeHierarchyBuilder.cs:31-74): what the reference's unused SAH
// path (:166-260) aims at, interchangeable with its median builder for any consumer.  The tree
// itself comes from the library's one SAH builder (wide_bvh.h: 32 bins x 3 axes, 4-wide, leaves of
// up to four primitives); this file only writes it out as 116-byte nodes: node 0 the Empty
// sentinel, node 1 the root, inner nodes breadth first, every leaf a node of its own whose slots
// carry the primitives' own boxes.
#include <cfloat>
#include <vector>

#include "../wide_bvh.h"
#include "host_internal.h"

namespace rtamd_host {

namespace {

void emptyNode(rtamd_node& n, bool leaf) {      // Node.Empty (:38-43): +FLT_MAX / -FLT_MAX bounds
    for (int i = 0; i < 4; i++) {
        n.Lx[i] = n.Ly[i] = n.Lz[i] = FLT_MAX;
        n.Hx[i] = n.Hy[i] = n.Hz[i] = -FLT_MAX;
        n.Data[i] = leaf ? -1 : 0;
    }
    n.Flags = leaf ? 1 : 0;
}

void setSlot(rtamd_node& n, int i, const float lo[3], const float hi[3], int32_t data) {
    n.Lx[i] = lo[0]; n.Ly[i] = lo[1]; n.Lz[i] = lo[2];
    n.Hx[i] = hi[0]; n.Hy[i] = hi[1]; n.Hz[i] = hi[2];
    n.Data[i] = data;
}

}  // namespace

int bvhBuildSah(const rtamd_aabb* aabbs, int n, rtamd_node* nodes, int capacity, int* nodeCount) {
    if (capacity < 2 || n < 0) return RTAMD_ERR_INVALID_ARG;
    emptyNode(nodes[0], false);
    if (n == 0) { emptyNode(nodes[1], false); *nodeCount = 2; return RTAMD_OK; }
    std::vector<rt::BuildBox> boxes((size_t)n);
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { boxes[i].lo[k] = aabbs[i].Min[k]; boxes[i].hi[k] = aabbs[i].Max[k]; }
    rt::WideTree tree;
    try {
        rt::buildWideTree(boxes.data(), n, 4, tree);
    } catch (...) {
        setLastError("out of memory in the SAH builder");
        return RTAMD_ERR_OUT_OF_MEMORY;
    }
    // wide inner node k -> output node at[k]; leaves get the indices after all inner nodes of their level's parent
    std::vector<int32_t> at(tree.nodes.size(), -1), queue;
    int next = 1;
    at[0] = next++;
    queue.push_back(0);
    for (size_t q = 0; q < queue.size(); q++) {
        const rt::WideTree::Node& w = tree.nodes[(size_t)queue[q]];
        const int self = at[(size_t)queue[q]];
        if (self >= capacity) { setLastError("node array too short for the SAH tree"); return RTAMD_ERR_BAD_INDEX; }
        rtamd_node& out = nodes[self];
        emptyNode(out, false);
        for (int s = 0; s < 4; s++) {
            const int32_t c = w.child[s];
            if (c == rt::widebvh::kEmpty) continue;
            if (next >= capacity) { setLastError("node array too short for the SAH tree"); return RTAMD_ERR_BAD_INDEX; }
            const int child = next++;
            setSlot(out, s, w.lo[s], w.hi[s], child);
            if (c >= 0) {
                at[(size_t)c] = child;
                queue.push_back(c);
            } else {
                const uint32_t code = (uint32_t)~c;
                const int start = (int)(code >> 3), count = (int)(code & 7u);
                rtamd_node& leaf = nodes[child];
                emptyNode(leaf, true);
                for (int k = 0; k < count && k < 4; k++) {
                    const int prim = tree.order[(size_t)(start + k)];
                    setSlot(leaf, k, aabbs[prim].Min, aabbs[prim].Max, prim);
                }
            }
        }
    }
    *nodeCount = next;
    return RTAMD_OK;
}

}  // namespace rtamd_host

extern "C" int rtamd_host_bvh_build_sah(const rtamd_aabb* aabbs, int32_t n, rtamd_node* nodes, int32_t cap, int32_t* node_count) {
    if (!nodes || !node_count || (n > 0 && !aabbs)) return RTAMD_ERR_INVALID_ARG;
    return rtamd_host::bvhBuildSah(aabbs, n, nodes, cap, node_count);
}

// Scene assembly on the host: BLAS bake, entities, the per-frame acceleration-structure
// update that flattens everything into the eight upload buffers.
//
// TEST / FIXTURE INFRASTRUCTURE, not product code: a restatement of the reference's C# scene
// producers, kept so that inputs can be made without Unity exactly as the reference makes them.
//
// Restates Assets/Editor/RayTracing/Authoring/RayTraceableBaker.cs:75-149 (bake),
// Assets/Script/RayTracing/System/BuildAccelerateStructureSystem.cs:41-163 (OnUpdate),
// :202-234 (PrepareBuildTLASJob), :250-273 (BuildBLASComputeBufferJob), :283-300
// (BuildPerObjectPropertyComputeBufferJob) and
// Assets/Script/RayTracing/Component/TopLevelAccelerateStructure.cs:14-40,64-68.

#include <cfloat>
#include <cstdio>
#include <cstring>
#include <memory>

#include "host_internal.h"
#include "rtamd_testhost.h"

namespace rtamd_host {

namespace {

inline float cmin4(const float* v) { return std::fmin(std::fmin(v[0], v[1]), std::fmin(v[2], v[3])); }
inline float cmax4(const float* v) { return std::fmax(std::fmax(v[0], v[1]), std::fmax(v[2], v[3])); }

// float4x4.TransformPoint: (c0*x + c1*y + c2*z + c3).xyz
inline void transformPoint(const float* m, const float p[3], float out[3]) {
    for (int r = 0; r < 3; r++) out[r] = ((m[r] * p[0] + m[4 + r] * p[1]) + m[8 + r] * p[2]) + m[12 + r];
}

}  // namespace
}  // namespace rtamd_host

using namespace rtamd_host;

extern "C" {

int rtamd_host_bvh_build(rtamd_point_and_index* points, const rtamd_aabb* aabbs, int32_t n, rtamd_node* nodes,
                         int32_t cap, int32_t* node_count) {
    if (!nodes || !node_count || (n > 0 && (!points || !aabbs))) return RTAMD_ERR_INVALID_ARG;
    return bvhBuild(points, aabbs, n, nodes, cap, node_count);
}
int rtamd_host_bvh_build_jobs(rtamd_point_and_index* points, const rtamd_aabb* aabbs, int32_t n, rtamd_node* nodes,
                              int32_t cap, int32_t* branch_count) {
    if (!nodes || !points || !aabbs) return RTAMD_ERR_INVALID_ARG;
    return bvhBuildJobs(points, aabbs, n, nodes, cap, branch_count);
}

int rtamd_host_scene_set_tlas_builder(rtamd_host_scene* s, int32_t tree) {
    if (!s || (tree != 0 && tree != 1)) return RTAMD_ERR_INVALID_ARG;
    s->tlasTree = tree;
    return RTAMD_OK;
}
// RayTraceableBaker.cs:75-149
int rtamd_host_bake_blas(rtamd_host_scene* s, const float* positions, const float* normals, int32_t nVerts,
                         const int32_t* indices, int32_t nTris, int32_t tree) {
    if (!s || !positions || !normals || !indices || nVerts <= 0 || nTris <= 0) return RTAMD_ERR_INVALID_ARG;
    for (int i = 0; i < nTris * 3; i++)
        if (indices[i] < 0 || indices[i] >= nVerts) { setLastError("triangle index out of range"); return RTAMD_ERR_BAD_INDEX; }
    auto blas = std::make_unique<Blas>();
    blas->vertices.assign(positions, positions + 3 * (size_t)nVerts);
    blas->normals.assign(normals, normals + 3 * (size_t)nVerts);
    blas->triangles.resize(nTris);
    std::vector<rtamd_aabb> aabbs(nTris);
    std::vector<rtamd_point_and_index> pts(nTris);
    for (int i = 0; i < nTris; i++) {
        blas->triangles[i] = {indices[3 * i], indices[3 * i + 1], indices[3 * i + 2]};
        const float* a = positions + 3 * (size_t)indices[3 * i];
        const float* b = positions + 3 * (size_t)indices[3 * i + 1];
        const float* c = positions + 3 * (size_t)indices[3 * i + 2];
        for (int k = 0; k < 3; k++) {
            aabbs[i].Max[k] = std::fmax(std::fmax(a[k], b[k]), c[k]);
            aabbs[i].Min[k] = std::fmin(std::fmin(a[k], b[k]), c[k]);
            pts[i].Position[k] = (aabbs[i].Max[k] + aabbs[i].Min[k]) * 0.5f;   // Aabb.Center
        }
        pts[i].Index = i;
    }
    int cap = nTris + 64;                      // sketchNodeCount, :129
    std::vector<rtamd_node> sketch((size_t)cap);
    std::memset(sketch.data(), 0, sketch.size() * sizeof(rtamd_node));
    int nodeCount = 0;
    int rc = tree == 1 ? bvhBuildSah(aabbs.data(), nTris, sketch.data(), cap, &nodeCount)
                       : bvhBuild(pts.data(), aabbs.data(), nTris, sketch.data(), cap, &nodeCount);
    if (rc != RTAMD_OK) { setLastError("BVH build failed"); return rc; }
    blas->nodes.assign(sketch.begin(), sketch.begin() + nodeCount);
    s->blobs.push_back(std::move(blas));
    return (int)s->blobs.size() - 1;
}

int rtamd_host_add_entity(rtamd_host_scene* s, int32_t blas, const float l2w[16], const float albedo[4],
                          const float specular[4], const float emission[4]) {
    if (!s || !l2w || !albedo || !specular || !emission) return RTAMD_ERR_INVALID_ARG;
    if (blas < 0 || blas >= (int)s->blobs.size()) return RTAMD_ERR_BAD_INDEX;
    Entity e;
    e.blas = blas;
    std::memcpy(e.l2w, l2w, sizeof(e.l2w));
    std::memcpy(e.albedo, albedo, sizeof(e.albedo));
    std::memcpy(e.specular, specular, sizeof(e.specular));
    std::memcpy(e.emission, emission, sizeof(e.emission));
    s->entities.push_back(e);
    return (int)s->entities.size() - 1;
}

int rtamd_host_set_entity_transform(rtamd_host_scene* s, int32_t entity, const float l2w[16]) {
    if (!s || !l2w) return RTAMD_ERR_INVALID_ARG;
    if (entity < 0 || entity >= (int)s->entities.size()) return RTAMD_ERR_BAD_INDEX;
    std::memcpy(s->entities[entity].l2w, l2w, sizeof(float) * 16);
    return RTAMD_OK;
}

// BuildAccelerateStructureSystem.OnUpdate, :41-163
int rtamd_host_scene_update(rtamd_host_scene* s) {
    if (!s) return RTAMD_ERR_INVALID_ARG;
    int n = (int)s->entities.size();
    if (n == 0) { setLastError("no entities (ObjectsCount == 0)"); return RTAMD_ERR_NO_SCENE; }

    // PrepareBuildTLASJob, :202-234
    std::vector<rtamd_aabb> aabbs(n);
    std::vector<rtamd_point_and_index> pts(n);
    for (int e = 0; e < n; e++) {
        const Blas& blas = *s->blobs[s->entities[e].blas];
        const rtamd_node& rootNode = blas.nodes[1];
        float mn[3] = {cmin4(rootNode.Lx), cmin4(rootNode.Ly), cmin4(rootNode.Lz)};
        float mx[3] = {cmax4(rootNode.Hx), cmax4(rootNode.Hy), cmax4(rootNode.Hz)};
        float center[3], ext[3];
        for (int k = 0; k < 3; k++) {
            center[k] = (mx[k] + mn[k]) * 0.5f;
            ext[k] = (mx[k] - mn[k]) * 0.5f;
            pts[e].Position[k] = center[k];     // mesh-local centre, as the reference does (:212-214)
        }
        pts[e].Index = e;
        static const float sg[8][3] = {{1, 1, 1}, {1, 1, -1}, {1, -1, 1}, {1, -1, -1},
                                       {-1, 1, 1}, {-1, 1, -1}, {-1, -1, 1}, {-1, -1, -1}};
        float mxw[3], mnw[3];
        for (int c = 0; c < 8; c++) {
            float p[3] = {center[0] + ext[0] * sg[c][0], center[1] + ext[1] * sg[c][1], center[2] + ext[2] * sg[c][2]};
            float w[3];
            transformPoint(s->entities[e].l2w, p, w);
            for (int k = 0; k < 3; k++) {
                if (c == 0) { mxw[k] = w[k]; mnw[k] = w[k]; }
                else { mxw[k] = std::fmax(mxw[k], w[k]); mnw[k] = std::fmin(mnw[k], w[k]); }
            }
        }
        for (int k = 0; k < 3; k++) { aabbs[e].Min[k] = mnw[k]; aabbs[e].Max[k] = mxw[k]; }
    }
    // TLAS.ScheduleBuildTree: NodeCount = BodyCount + MaxNumTreeBranches (TopLevelAccelerateStructure.cs:17).
    // That is too small once the first levels take more than 64 nodes (2 + 4 + 16 + 64 = 86 with 64
    // branches): each branch is given a node range as long as its primitive range, behind those 86,
    // and the last branches then write past BodyCount + 64 (seen from about 1 280 bodies).  The array
    // here has room for the 86; the tree built is the one the reference's code describes.
    int tlasCount = n + 128;
    s->tlas.assign((size_t)tlasCount, rtamd_node{});
    int rc;
    if (s->tlasTree == 1) {
        int used = 0;
        rc = bvhBuildSah(aabbs.data(), n, s->tlas.data(), tlasCount, &used);
        s->branchCount = 0;
    } else {
        rc = bvhBuildJobs(pts.data(), aabbs.data(), n, s->tlas.data(), tlasCount, &s->branchCount);
    }
    if (rc != RTAMD_OK) { setLastError(std::string("TLAS build failed: ") + rtamd_host_last_error()); return rc; }

    // BLAS index by first use, :70-94
    std::vector<int> blasIndexOfBlob(s->blobs.size(), -1);
    std::vector<int> blobOfBlasIndex;
    for (int e = 0; e < n; e++) {
        int blob = s->entities[e].blas;
        if (blasIndexOfBlob[blob] < 0) {
            blasIndexOfBlob[blob] = (int)blobOfBlasIndex.size();
            blobOfBlasIndex.push_back(blob);
        }
    }
    int blasCount = (int)blobOfBlasIndex.size();
    s->nBlas = blasCount;
    // offsets, :98-117
    s->blasOffsets.resize(blasCount); s->vtxOffsets.resize(blasCount); s->triOffsets.resize(blasCount);
    int totalNodes = 0, totalVerts = 0, totalTris = 0;
    for (int i = 0; i < blasCount; i++) {
        const Blas& b = *s->blobs[blobOfBlasIndex[i]];
        s->blasOffsets[i] = totalNodes; totalNodes += (int)b.nodes.size();
        s->vtxOffsets[i] = totalVerts; totalVerts += (int)(b.vertices.size() / 3);
        s->triOffsets[i] = totalTris; totalTris += (int)b.triangles.size();
    }
    // BuildBLASComputeBufferJob, :250-273
    s->blasNodes.resize(totalNodes); s->vertices.resize(totalVerts); s->triangles.resize(totalTris);
    for (int i = 0; i < blasCount; i++) {
        const Blas& b = *s->blobs[blobOfBlasIndex[i]];
        std::copy(b.nodes.begin(), b.nodes.end(), s->blasNodes.begin() + s->blasOffsets[i]);
        int nv = (int)(b.vertices.size() / 3);
        for (int v = 0; v < nv; v++) {
            rtamd_vertex& o = s->vertices[s->vtxOffsets[i] + v];
            for (int k = 0; k < 3; k++) { o.position[k] = b.vertices[3 * v + k]; o.normal[k] = b.normals[3 * v + k]; }
        }
        std::copy(b.triangles.begin(), b.triangles.end(), s->triangles.begin() + s->triOffsets[i]);
    }
    // BuildPerObjectPropertyComputeBufferJob, :283-300
    s->objects.resize(n);
    for (int e = 0; e < n; e++) {
        const Entity& en = s->entities[e];
        rtamd_object_property& o = s->objects[e];
        o.BLASIndex = blasIndexOfBlob[en.blas];
        std::memcpy(o.albedo, en.albedo, sizeof(o.albedo));
        std::memcpy(o.specular, en.specular, sizeof(o.specular));
        std::memcpy(o.emission, en.emission, sizeof(o.emission));
        if (rtamd_host_inverse4x4(en.l2w, o.WorldToLocal) != 0) {
            setLastError("singular LocalToWorld");
            return RTAMD_ERR_INVALID_ARG;
        }
        std::memcpy(o.LocalToWorld, en.l2w, sizeof(o.LocalToWorld));
    }
    return RTAMD_OK;
}

}  // extern "C"

/*
 * rtamd_testhost.h — C ABI of librtamd_testhost.so: the reference's C# scene producers restated in
 * C++.  TEST AND FIXTURE INFRASTRUCTURE (inputs for tests, golden files and the synthetic
 * benchmark scenes made exactly as the reference makes them); not part of the product libraries.
 * Works on the scene container of librtamd_host.so (rtamd_host_scene_create / _buffers / _destroy).
 *
 * Reference code restated (paths relative to the reference checkout):
 *   Assets/Script/BVH/BoundingVolumeHierarchyBuilder.cs      (builder, Refit)
 *   Assets/Script/BVH/Base/Math/{Aabb,FourTransposedAabbs,Math}.cs
 *   Assets/Editor/RayTracing/Authoring/RayTraceableBaker.cs:75-149
 *   Assets/Script/RayTracing/System/BuildAccelerateStructureSystem.cs:41-163,202-300
 *   Assets/Script/RayTracing/Component/TopLevelAccelerateStructure.cs
 */
#ifndef RTAMD_TESTHOST_H
#define RTAMD_TESTHOST_H

#include "rtamd_host.h"

#ifdef __cplusplus
extern "C" {
#endif

/* BoundingVolumeHierarchy.Build (BoundingVolumeHierarchyBuilder.cs:529-558), median
 * split (useSah = false, the only mode any caller uses).  `points` is permuted in
 * place, as in the reference.  `nodes` must hold n + 64 entries
 * (RayTraceableBaker.cs:129).  Returns 0 or a negative rtamd status. */
int rtamd_host_bvh_build(rtamd_point_and_index* points, const rtamd_aabb* aabbs, int32_t n,
                         rtamd_node* nodes, int32_t nodes_capacity, int32_t* node_count);

/* BoundingVolumeHierarchy.ScheduleBuildJobs (:482-527) run to completion:
 * BuildFirstNLevelsJob, every BuildBranchesJob, FinalizeTreeJob.  This is the
 * TLAS build path (TopLevelAccelerateStructure.cs:64-68).  `nodes` must hold
 * n + 64 entries; entries the build does not reach are left zeroed.
 * branch_count receives the number of branches. */
int rtamd_host_bvh_build_jobs(rtamd_point_and_index* points, const rtamd_aabb* aabbs, int32_t n,
                              rtamd_node* nodes, int32_t nodes_capacity, int32_t* branch_count);

/* BLASBaker.BuildBLASBlobAssetReference (RayTraceableBaker.cs:75-149): copies
 * the mesh, builds per-triangle AABBs + centres and the BLAS.  tree = 0: the
 * reference's median builder, 1: the SAH builder.  Returns the blob handle
 * (>= 0) or a negative status.  Baking the same mesh once and instancing the
 * handle is the reference's blob de-duplication (:59-65). */
int rtamd_host_bake_blas(rtamd_host_scene* s, const float* positions, const float* normals,
                         int32_t n_vertices, const int32_t* indices, int32_t n_triangles, int32_t tree);

/* One entity with RayTraceableBLASComponent + RayTraceableMaterialComponent +
 * LocalToWorld (column-major).  Returns the entity index (= object index =
 * entityIndexInQuery) or a negative status. */
int rtamd_host_add_entity(rtamd_host_scene* s, int32_t blas_handle, const float local_to_world[16],
                          const float albedo[4], const float specular[4], const float emission[4]);
int rtamd_host_set_entity_transform(rtamd_host_scene* s, int32_t entity, const float local_to_world[16]);

/* BuildAccelerateStructureSystem.OnUpdate (:41-163): world AABBs, TLAS build,
 * BLAS index assignment by first use, offsets, flattening, object properties
 * (WorldToLocal = inverse(LocalToWorld)). */
int rtamd_host_scene_update(rtamd_host_scene* s);

/* Builder of the TLAS in rtamd_host_scene_update: 0 (default) the reference's three-job
 * median build (BoundingVolumeHierarchyBuilder.cs:482-527), which its own fixed-size
 * arrays limit to roughly 1 300 objects (the update then fails with a message instead
 * of overrunning them as the reference would); 1 the binned-SAH builder, any count. */
int rtamd_host_scene_set_tlas_builder(rtamd_host_scene* s, int32_t tree);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_TESTHOST_H */

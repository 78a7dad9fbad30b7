/*
 * rtamd_host.h — C ABI of the host-side helpers (CPU only, no GPU calls): librtamd_host.so.
 *
 * What a caller without Unity needs around the compute path of houshuo/RayTracing: mesh
 * sources (generators, COLLADA, subdivision), Unity's camera matrices, a 4-wide SAH tree
 * writer in the reference's node format, the flattened scene (the eight buffers of
 * rtamd_scene_upload) as a container and a file, and HDR image codecs for the sky and the
 * framebuffer.  In a Unity build the project's own C# code produces the same buffers and
 * this library is not needed.
 *
 * The reference's own producers restated (its median-split BVH builder, the edit-time BLAS
 * bake and the per-frame flattening system) are test and fixture infrastructure and live in a
 * library of their own: tests/producers/rtamd_testhost.h, librtamd_testhost.so.
 */
#ifndef RTAMD_HOST_H
#define RTAMD_HOST_H

#include <stdint.h>
#include "rtamd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Aabb (Aabb.cs:11-14): Min then Max. */
typedef struct rtamd_aabb { float Min[3]; float Max[3]; } rtamd_aabb;
/* BoundingVolumeHierarchy.PointAndIndex (BoundingVolumeHierarchyBuilder.cs:87-91). */
typedef struct rtamd_point_and_index { float Position[3]; int32_t Index; } rtamd_point_and_index;

/* A 4-wide tree over primitive boxes in the reference's node format
 * (BoundingVolumeHierarchyBuilder.cs:31-74): binned surface-area-heuristic splits (what the
 * reference's unused SAH path, :166-260, aims at).  `nodes` must hold n + 64 entries. */
int rtamd_host_bvh_build_sah(const rtamd_aabb* aabbs, int32_t n,
                             rtamd_node* nodes, int32_t nodes_capacity, int32_t* node_count);

/* ------------------------------------------------------------------ */
/* A flattened scene on the host: the eight upload buffers              */
/* ------------------------------------------------------------------ */
typedef struct rtamd_host_scene rtamd_host_scene;

rtamd_host_scene* rtamd_host_scene_create(void);
void rtamd_host_scene_destroy(rtamd_host_scene* s);

typedef struct rtamd_host_buffers {
    const rtamd_node* tlas; int32_t n_tlas_nodes;
    const int32_t* blas_offsets;
    const rtamd_node* blas; int32_t n_blas_nodes;
    const int32_t* vertices_offsets;
    const rtamd_vertex* vertices; int32_t n_vertices;
    const int32_t* triangles_offsets;
    const rtamd_int3* triangles; int32_t n_triangles;
    int32_t n_blas;
    const rtamd_object_property* objects; int32_t n_objects;
    int32_t tlas_branch_count;
} rtamd_host_buffers;

/* The eight buffers the scene holds (after a load, or an update by a producer); valid until the next load / update / destroy. */
int rtamd_host_scene_buffers(const rtamd_host_scene* s, rtamd_host_buffers* out);

/* ------------------------------------------------------------------ */
/* Camera and math                                                      */
/* ------------------------------------------------------------------ */

/* camera.cameraToWorldMatrix and camera.projectionMatrix.inverse
 * (RayTracingPass.cs:74-75) for a Unity camera: position, rotation quaternion
 * (x,y,z,w), vertical field of view in degrees, aspect = width/height, near,
 * far.  Unity / OpenGL convention: the camera looks down -Z of camera space. */
void rtamd_host_camera(const float position[3], const float rotation_xyzw[4], float fov_y_deg,
                       float aspect, float z_near, float z_far, rtamd_frame_cbuffer* out);

/* General 4x4 inverse (column-major), the stand-in for math.inverse
 * (BuildAccelerateStructureSystem.cs:297).  Returns 0, or -1 if singular. */
int rtamd_host_inverse4x4(const float m[16], float out[16]);

/* TRS matrix (column-major) from translation, rotation quaternion (x,y,z,w), scale. */
void rtamd_host_trs(const float t[3], const float q_xyzw[4], const float s[3], float out[16]);

/* ------------------------------------------------------------------ */
/* Mesh sources                                                         */
/* ------------------------------------------------------------------ */
typedef struct rtamd_host_mesh {
    float* positions;  /* 3 * n_vertices */
    float* normals;    /* 3 * n_vertices */
    int32_t n_vertices;
    int32_t* indices;  /* 3 * n_triangles */
    int32_t n_triangles;
} rtamd_host_mesh;

void rtamd_host_mesh_free(rtamd_host_mesh* m);
/* UV sphere of the given radius: `segments` around, `rings` pole to pole. */
int rtamd_host_mesh_uv_sphere(int32_t segments, int32_t rings, float radius, rtamd_host_mesh* out);
/* Unit quad in the XY plane facing -Z (Unity's built-in Quad), 2 triangles. */
int rtamd_host_mesh_quad(rtamd_host_mesh* out);
/* Unit cube centred at the origin, 12 triangles, per-face normals. */
int rtamd_host_mesh_cube(rtamd_host_mesh* out);
/* COLLADA <triangles> mesh (VERTEX/NORMAL index pairs).  A Z_UP asset is rotated to Y-up. */
int rtamd_host_mesh_load_dae(const char* path, rtamd_host_mesh* out);
/* Midpoint subdivision, every triangle -> 4, `rounds` times. */
int rtamd_host_mesh_subdivide(const rtamd_host_mesh* in, int32_t rounds, rtamd_host_mesh* out);

/* Flattened-scene file: the eight arrays behind a small header.  Save returns 0. */
int rtamd_host_scene_save(const rtamd_host_buffers* b, const char* path);
/* Load into a scene object (replaces its buffers; no entities are recreated). */
int rtamd_host_scene_load(rtamd_host_scene* s, const char* path);

/* ------------------------------------------------------------------ */
/* HDR images: sky texture in, framebuffer out                          */
/* ------------------------------------------------------------------ */

/* Reads the lat-long sky the shader samples as _SkyboxTexture
 * (Assets/Shaders/RayTracingShader.compute:22,409; the reference ships
 * the .exr files of Assets/SkyBoxes through Unity's importer) from the file itself:
 * OpenEXR (single-part scanline; NONE / RLE / ZIPS / ZIP / PIZ; HALF, FLOAT or UINT
 * R,G,B,A or Y) or Radiance RGBE (.hdr).  Returns RGBA float32, malloc'ed, to be
 * released with rtamd_host_image_free.  flip_y = 1 puts the file's bottom scanline
 * in row 0, which is Unity's texture space (v = 0 at the bottom) and what
 * rtamd_sky_upload expects; 0 keeps file order.  Returns 0, or -1 with
 * rtamd_host_last_error set (the lossy codecs are not supported). */
int rtamd_host_image_load(const char* path, int32_t flip_y, float** out_rgba, int32_t* out_w, int32_t* out_h);
void rtamd_host_image_free(float* rgba);
/* Writes RGBA float32 by extension: .exr (FLOAT RGBA, ZIP), .hdr (flat RGBE, alpha
 * dropped), .ppm (8 bits, clamped to [0,1] like the reference's UNORM8 target,
 * RayTracingPass.cs:58).  flip_y = 1: row 0 of `rgba` is the bottom of the picture. */
int rtamd_host_image_save(const char* path, const float* rgba, int32_t w, int32_t h, int32_t flip_y);

const char* rtamd_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_HOST_H */

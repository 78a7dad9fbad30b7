/*
 * rtamd.h — C ABI of the MI355X (gfx950) ray-tracing backend.
 *
 * This is the drop-in boundary for the compute path of houshuo/RayTracing:
 * the library accepts the project's own scene buffers byte-for-byte (the
 * arrays `BuildAccelerateStructureSystem` uploads with ComputeBuffer.SetData)
 * plus the per-frame constant buffer that `RayTracingPass.Execute` fills, and
 * renders with hand-written HIP kernels.  Every entry point is `extern "C"`,
 * takes plain pointers / sizes / blittable structs only, returns an int
 * status (0 = ok, negative = error) and never throws across the boundary.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   struct layouts   Assets/Script/Common/RayTraceMaterial.cs:6-35,
 *                    Assets/Script/BVH/BoundingVolumeHierarchyBuilder.cs:31-37,
 *                    Assets/Script/BVH/Base/Math/FourTransposedAabbs.cs:6-10,
 *                    mirrored in Assets/Shaders/RayTracingShader.compute:8-19,80-124
 *   scene bindings   Assets/Script/RenderPass/RayTracingPass.cs:83-97
 *   dispatch         Assets/Script/RenderPass/RayTracingPass.cs:101
 */
#ifndef RTAMD_H
#define RTAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Wire structs — byte-identical to the reference (all members 4-byte) */
/* ------------------------------------------------------------------ */

/* BoundingVolumeHierarchy.Node (BoundingVolumeHierarchyBuilder.cs:31-37,
 * RayTracingShader.compute:80-91).  116 bytes.  Lane i of every bound = child i.
 * Flags == 0: internal, Data[i] = child node index relative to the tree base,
 * 0 = empty slot.  Flags != 0: leaf, Data[i] = primitive index, -1 = empty.
 * Node 0 is a sentinel, the root is node 1. */
typedef struct rtamd_node {
    float   Lx[4], Hx[4];
    float   Ly[4], Hy[4];
    float   Lz[4], Hz[4];
    int32_t Data[4];
    int32_t Flags;
} rtamd_node;

/* Vertex (RayTraceMaterial.cs:6-10, RayTracingShader.compute:102-106). 24 bytes. */
typedef struct rtamd_vertex {
    float position[3];
    float normal[3];
} rtamd_vertex;

/* int3 triangle (RayTracableComponent.cs:15, RayTracingShader.compute:109). 12 bytes.
 * Vertex indices relative to the mesh's vertex base. */
typedef struct rtamd_int3 {
    int32_t x, y, z;
} rtamd_int3;

/* PerObjectProperty (RayTraceMaterial.cs:19-25, RayTracingShader.compute:117-123).
 * 180 bytes.  Matrices are column-major: 4 consecutive floats = one column. */
typedef struct rtamd_object_property {
    int32_t BLASIndex;
    float   albedo[4];
    float   specular[4];
    float   emission[4];
    float   WorldToLocal[16];
    float   LocalToWorld[16];
} rtamd_object_property;

/* FrameCBuffer (RayTraceMaterial.cs:28-35, RayTracingShader.compute:8-19). 144 bytes. */
typedef struct rtamd_frame_cbuffer {
    float   CameraToWorld[16];
    float   CameraInverseProjection[16];
    float   PixelOffset[2];
    float   Seed;
    int32_t ReflectionNum;
} rtamd_frame_cbuffer;

/* ------------------------------------------------------------------ */
/* Extension parameters (no counterpart in the reference cbuffer)      */
/* ------------------------------------------------------------------ */

enum {
    RTAMD_MODE_REF = 0, /* RayTracingShader.compute:413-429 exactly: primary + one mirror bounce */
    RTAMD_MODE_PT  = 1  /* bounced path tracing grown from the shader's unused roulette code (:386-411) */
};

enum {
    RTAMD_FLAG_ACCUMULATE      = 1 << 0, /* PT: add this call's samples to the context's running sum */
    RTAMD_FLAG_NO_JITTER       = 1 << 1, /* PT: use FrameCBuffer.PixelOffset instead of a random sub-pixel offset */
    RTAMD_FLAG_REFERENCE_ORDER = 1 << 2, /* traverse in the reference's exact node order on the reference's own node layout */
    RTAMD_FLAG_OUTPUT_SUM      = 1 << 3, /* PT: write the un-normalised sample sum (for cross-GPU sample sharding) */
    RTAMD_FLAG_COUNTERS        = 1 << 4, /* also fill the work counters (forces REFERENCE_ORDER so the counts are the reference's) */
    RTAMD_FLAG_SHARD_SAMPLES   = 1 << 5, /* PT on a context over several GPUs: split the samples, not the tiles, over the GPUs */
    RTAMD_FLAG_DIELECTRICS     = 1 << 6, /* PT: objects whose specular[3] is RTAMD_MATERIAL_DIELECTRIC transmit light (see below) */
    RTAMD_FLAG_DEBUG_CHECKS    = 1 << 7  /* diagnostic: run the bounds-checked variant of the production kernel (one path per lane): every node,
                                            instance, triangle and stack index is checked against the scene's sizes before it is
                                            followed; a violation ends that pixel and the call returns RTAMD_ERR_BAD_INDEX.  Same image on an
                                            intact scene, at about the speed of RTAMD_PIPELINE=persistent.  Not together with DIELECTRICS,
                                            REFERENCE_ORDER, COUNTERS or rtamd_render_begin (RTAMD_ERR_INVALID_ARG).  No counterpart in the
                                            reference, whose shader trusts its buffers. */
};

/* Dielectric extension (no counterpart in the reference: RayTraceMaterial.cs:12-17 has no
 * transmission term and Trisection culls back faces, RayTracingShader.compute:205).  It uses
 * the fourth components of albedo and specular, which the reference declares (float4) and the
 * shader never reads (.xyz only, :336-338): with RTAMD_FLAG_DIELECTRICS set, an object whose
 * specular[3] == RTAMD_MATERIAL_DIELECTRIC is glass with refractive index albedo[3] and
 * transmission tint albedo[0..2].  A path that refracted into glass tests triangles from both
 * sides until it refracts out again; at a glass surface Schlick's Fresnel term chooses between
 * mirror reflection and refraction with the segment's roulette draw.  Without the flag the
 * fourth components are ignored, as in the reference. */
#define RTAMD_MATERIAL_DIELECTRIC 2.0f

/* Work counters, summed over every ray segment of one render call. */
typedef struct rtamd_counters {
    uint64_t rays;            /* ray segments traversed (primary + every bounce) */
    uint64_t nodes;           /* BVH nodes fetched, TLAS + BLAS */
    uint64_t tris;            /* triangles tested */
    uint64_t instances;       /* BLAS entries (object property fetches) */
    uint64_t hits;            /* accepted (closer) hits */
    uint64_t max_blas_stack;  /* deepest BLAS stack seen */
    uint64_t max_tlas_stack;  /* deepest TLAS stack seen */
    uint64_t reserved;
} rtamd_counters;

typedef struct rtamd_ext_params {
    uint32_t struct_size;   /* = sizeof(rtamd_ext_params) */
    int32_t  mode;          /* RTAMD_MODE_* */
    int32_t  spp;           /* PT: samples per pixel rendered by this call (>=1) */
    int32_t  max_bounces;   /* PT: scattering events after the primary segment (segments <= max_bounces+1) */
    uint64_t seed;          /* PT: RNG seed; the stream is keyed (seed, pixel, sample, bounce, dimension) */
    int32_t  sample_offset; /* PT: index of the first sample of this call */
    int32_t  tile_x, tile_y, tile_w, tile_h; /* render only this pixel rectangle (tile_w<=0 or tile_h<=0: whole image) */
    int32_t  shard_rank, shard_count;        /* render only 32x32 tiles t with t % shard_count == shard_rank (count<=1: all) */
    uint32_t flags;         /* RTAMD_FLAG_* */
    float*   out_t;         /* optional, width*height floats: primary hit parameter t (+inf on miss) */
    rtamd_counters* out_counters; /* optional (host pointer) */
} rtamd_ext_params;

/* ------------------------------------------------------------------ */
/* Status codes                                                         */
/* ------------------------------------------------------------------ */
enum {
    RTAMD_OK                 =  0,
    RTAMD_ERR_INVALID_ARG    = -1,
    RTAMD_ERR_BAD_INDEX      = -2, /* an offset / index in the uploaded buffers is out of range */
    RTAMD_ERR_OUT_OF_MEMORY  = -3,
    RTAMD_ERR_HIP            = -4,
    RTAMD_ERR_STACK_OVERFLOW = -5, /* traversal needed more stack than the reference shader declares (:93-94) */
    RTAMD_ERR_NO_SCENE       = -6, /* render before a scene upload (the reference skips the dispatch: RayTracingPass.cs:68) */
    RTAMD_ERR_NO_DEVICE      = -7
};

typedef struct rtamd_ctx rtamd_ctx;

/* Replaces the ComputeShader / ComputeBuffer ownership of RayTracingPass.Create
 * (RayTracingPass.cs:119-125) and BuildAccelerateStructureSystem.OnCreate (:35-39). */
int rtamd_ctx_create(int device_ordinal, rtamd_ctx** out_ctx);

/* One context over several GPUs of the node, in this one process (the reference is one Unity
 * process on one GPU: RayTracingPass.cs:62-109 has nothing to replace here; this is what its
 * dispatch needs when the image is partitioned).  Every other entry point takes the context as
 * it takes an ordinary one: a scene upload replicates the scene on every device (one host
 * thread each), rtamd_render splits the frame — by default the 32x32-pixel tiles t with
 * t % n == d go to device d, with RTAMD_FLAG_SHARD_SAMPLES (PT) every device renders its share
 * of the samples — and assembles it on device_ordinals[0]: the tiles are gathered
 * (bit-identical to a one-GPU render), the sample sums reduced (equal up to fp32 summation
 * order), with RCCL over xGMI (ncclSend/ncclRecv, ncclReduce; one communicator rank per
 * device, ncclCommInitAll) or, when a device is listed twice or librccl cannot be loaded,
 * with peer copies.  out_rgba / out_prim_id / ext->out_t may be host memory or device memory
 * of any GPU.  Not available on such a context: rtamd_set_stream, rtamd_render_noalloc,
 * ext->shard_count > 1, RTAMD_FLAG_ACCUMULATE together with RTAMD_FLAG_SHARD_SAMPLES. */
int rtamd_ctx_create_multi(const int32_t* device_ordinals, int32_t n_devices, rtamd_ctx** out_ctx);

/* Number of devices of the context (1 for an ordinary one), the rank count of the RCCL
 * communicator that carries the exchange (0: peer copies), and the device time of the last
 * frame's exchange step.  Any pointer may be NULL.  No counterpart in the reference. */
int rtamd_multi_info(const rtamd_ctx* ctx, int32_t* n_devices, int32_t* rccl_ranks, float* last_exchange_ms);
/* Why RCCL is not carrying the exchange, or "" when it is.  Never NULL. */
const char* rtamd_multi_exchange_note(const rtamd_ctx* ctx);

/* Replaces RayTracingPass.Dispose (:137-145) and BuildAccelerateStructureSystem.OnDestroy (:165-176). */
int rtamd_ctx_destroy(rtamd_ctx* ctx);

/* Last error text of this context (or of ctx creation when ctx is NULL). Never NULL. */
const char* rtamd_last_error_string(const rtamd_ctx* ctx);

/* Replaces the eight ComputeBuffer.SetData calls of
 * BuildAccelerateStructureSystem.OnUpdate (BuildAccelerateStructureSystem.cs:124-158)
 * and the eight SetComputeBufferParam bindings of RayTracingPass.Execute (:87-97).
 * All pointers are host memory in the reference layouts; the library copies
 * (and re-packs for the GPU) before returning.  n_tlas_nodes may include the
 * reference's uninitialised tail (TopLevelAccelerateStructure.cs:17,30): only
 * nodes reachable from node 1 are read. */
int rtamd_scene_upload(rtamd_ctx* ctx,
                       const rtamd_node* tlas, int32_t n_tlas_nodes,
                       const int32_t* blas_offsets,
                       const rtamd_node* blas, int32_t n_blas_nodes,
                       const int32_t* vertices_offsets,
                       const rtamd_vertex* vertices, int32_t n_vertices,
                       const int32_t* triangles_offsets,
                       const rtamd_int3* triangles, int32_t n_triangles,
                       int32_t n_blas,
                       const rtamd_object_property* objects, int32_t n_objects);

/* Cheap per-frame refresh for moving objects: replaces only _TLAS and
 * _ObjectsProperty (the reference re-uploads everything each frame,
 * BuildAccelerateStructureSystem.cs:148-158; the meshes are static). n_objects
 * must equal the uploaded scene's. */
int rtamd_scene_refresh_objects(rtamd_ctx* ctx,
                                const rtamd_node* tlas, int32_t n_tlas_nodes,
                                const rtamd_object_property* objects, int32_t n_objects);

/* Who builds the library's own tree over the instances on a scene upload or an object
 * refresh.  The reference rebuilds its TLAS on the CPU every frame
 * (BuildAccelerateStructureSystem.cs:56-66, TopLevelAccelerateStructure.cs:14-40); that
 * tree is still taken from the caller and decides ties, but the production kernel walks a
 * private tree over the same instance boxes, rebuilt either on the host (binned SAH: best
 * tree, about a microsecond per instance) or on the GPU (Morton order + binary radix tree,
 * collapsed to 4-wide nodes: constant time up to ~10^5 instances).  AUTO picks the GPU
 * from 4096 instances on.  Which instance a ray hits does not depend on the choice. */
#define RTAMD_TLAS_BUILD_AUTO 0
#define RTAMD_TLAS_BUILD_HOST 1
#define RTAMD_TLAS_BUILD_DEVICE 2
int rtamd_set_tlas_build(rtamd_ctx* ctx, int32_t mode);

/* Who builds the library's own trees over the meshes at rtamd_scene_upload.  The reference bakes its
 * bottom-level trees at edit time (Assets/Editor/RayTracing/Authoring/RayTraceableBaker.cs:75-149) and the caller
 * still passes those: they give every triangle its leaf-slot box and its place in the visiting order, which decide
 * hits and ties.  The hierarchy the production kernels walk is rebuilt: on the host (binned SAH on up to 8 threads,
 * the tree that traces fastest; AUTO and HOST) or on the GPU (Morton order, binary radix tree over the triangles,
 * sibling triangles paired into leaves, collapsed to 4-wide nodes: 25 ms per million triangles against 85, frames
 * trace 4-8 % slower; for meshes that change between frames; meshes below 256 triangles stay with the host builder).  Which triangle a ray hits does not depend on the choice.
 * Takes effect at the next rtamd_scene_upload. */
#define RTAMD_BLAS_BUILD_AUTO 0
#define RTAMD_BLAS_BUILD_HOST 1
#define RTAMD_BLAS_BUILD_DEVICE 2
int rtamd_set_blas_build(rtamd_ctx* ctx, int32_t mode);

/* The last rtamd_scene_upload: wall-clock time of the whole call, the part of it spent building the meshes'
 * trees, and how many meshes' trees the GPU built.  Any pointer may be NULL.  (No counterpart in the reference.) */
int rtamd_last_upload_stats(const rtamd_ctx* ctx, float* upload_ms, float* tree_build_ms, int32_t* meshes_built_on_device);

/* Shape of that private tree after the last upload / refresh: number of 4-wide nodes,
 * number of levels, and whether it was built on the GPU (1) or the host (0).  Any
 * pointer may be NULL. */
int rtamd_tlas_info(const rtamd_ctx* ctx, int32_t* nodes, int32_t* levels, int32_t* built_on_device);

/* Replaces SetComputeTextureParam(.., "_SkyboxTexture", ..) (RayTracingPass.cs:85).
 * rgba32f: height rows of width RGBA float texels, row 0 = v 0.  Sampled
 * bilinear, repeat, mip 0 (Assets/SkyBoxes/dikhololo_night_4k.exr.meta:33,36,39-40).
 * Without a sky the miss colour is black. */
int rtamd_sky_upload(rtamd_ctx* ctx, const float* rgba32f, int32_t width, int32_t height);

/* Replaces cmd.DispatchCompute(shader, 0, RTWidth, RTHeight, 1)
 * (RayTracingPass.cs:101) together with the Result binding (:99).
 *   frame        the reference cbuffer, required.
 *   ext          NULL = RTAMD_MODE_REF with defaults.
 *   out_rgba     optional, width*height*4 floats, row-major, row 0 = uv.y -1
 *                (RayTracingShader.compute:439).  Host or device pointer.
 *   out_prim_id  optional, width*height*2 int32: (object index, triangle index)
 *                of the un-jittered primary hit, (-1,-1) on miss.  Host or device.
 * Blocking: returns after the image is complete.  Pixels outside the tile
 * rectangle / shard are left untouched. */
int rtamd_render(rtamd_ctx* ctx, const rtamd_frame_cbuffer* frame,
                 const rtamd_ext_params* ext, int32_t width, int32_t height,
                 float* out_rgba, int32_t* out_prim_id);

/* rtamd_render for callers that cannot allocate device memory: tight serving loops,
 * harnesses that forbid hipMalloc, hosts that budget device memory themselves.  No
 * counterpart in the reference.  Same arguments, same image bit for bit, but the call
 * allocates and frees nothing: its working memory (work-queue head, ray counter, the mask and
 * list of pixels handed to the reference-order kernel, the tile list, hand-over samples, and a
 * 16 MiB overflow area for the traversal stacks) is a fixed 22 303 488-byte array that each GPU
 * the library runs on holds until the process ends.  The call still synchronises its stream,
 * reads a small block back to the host and decides there whether a hand-over pass follows, so
 * it cannot be recorded into a HIP graph.
 * Limits (RTAMD_ERR_INVALID_ARG beyond them): width*height <= 3840*2160 and at most 8192
 * tiles of 32x32 pixels; spp <= 262144; out_rgba, out_prim_id and ext->out_t must be
 * device memory (or NULL); RTAMD_FLAG_ACCUMULATE only once rtamd_render has created the
 * running sum at this size.  At most 65536 pixels of a frame may need the reference-order
 * kernel (RTAMD_ERR_OUT_OF_MEMORY otherwise; typical frames need about a hundred, a REF
 * frame one row and one column).  PT mode runs one path per lane here (the path pool
 * needs buffers sized to the frame).  Traversal stacks keep 16 entries on chip and, on a
 * 256-CU part, 16 more in the fixed overflow area (32 in all; the deepest stack of the
 * benchmark scenes is 27): a pixel with a deeper stack goes to the reference-order kernel.
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two contexts,
 * two threads, two streams) race. */
int rtamd_render_noalloc(rtamd_ctx* ctx, const rtamd_frame_cbuffer* frame,
                         const rtamd_ext_params* ext, int32_t width, int32_t height,
                         float* out_rgba_device, int32_t* out_prim_id_device);

/* rtamd_render in two halves, for a host that renders frame after frame (the reference's pass
 * runs once per camera per frame, RayTracingPass.cs:62-109, and its DispatchCompute (:101) does
 * not block either: the command buffer is executed later): rtamd_render_begin queues the frame
 * and returns, rtamd_render_wait completes the oldest frame begun and returns its status; the
 * rtamd_last_* statistics then describe that frame.  At most two frames are in flight, so the
 * second frame's kernels fill the GPU while the first one's last paths finish, are summed and
 * the host looks at its counters.  Same image as rtamd_render, bit for bit (one exception:
 * with RTAMD_FLAG_ACCUMULATE a pixel that needed the per-pixel hand-over pass may receive two
 * frames' sums in the other order).  Limits (RTAMD_ERR_INVALID_ARG): outputs in device memory,
 * a separate output buffer per frame in flight; no RTAMD_FLAG_COUNTERS /
 * RTAMD_FLAG_REFERENCE_ORDER; a path-traced frame must fit one batch of paths
 * (width * height * spp up to 2^29 on a 288 GB part); not on a context over several GPUs.
 * While a frame is in flight, rtamd_render, scene / sky uploads, rtamd_accum_reset and
 * rtamd_set_stream are refused. */
int rtamd_render_begin(rtamd_ctx* ctx, const rtamd_frame_cbuffer* frame,
                       const rtamd_ext_params* ext, int32_t width, int32_t height,
                       float* out_rgba_device, int32_t* out_prim_id_device);
int rtamd_render_wait(rtamd_ctx* ctx);

/* Drop the running sum used by RTAMD_FLAG_ACCUMULATE. */
int rtamd_accum_reset(rtamd_ctx* ctx);

/* The running sum itself (width*height*4 floats: rgb sums and the sample count per pixel), out
 * to the caller and back in: with the next ext->sample_offset it is all a progressive render
 * needs to stop and be continued later, in another process or on another GPU, with the same
 * bits as if it had never stopped.  Either pointer may be host or device memory.  No
 * counterpart in the reference (it keeps no history buffer, AddShader.shader:11). */
int rtamd_accum_download(rtamd_ctx* ctx, float* sums, int32_t width, int32_t height);
int rtamd_accum_upload(rtamd_ctx* ctx, const float* sums, int32_t width, int32_t height);

/* Run kernels of this context on a caller-owned HIP stream (hipStream_t as
 * void*; NULL = the context's own stream). */
int rtamd_set_stream(rtamd_ctx* ctx, void* hip_stream);

/* Device time of the traversal kernel(s) of the last rtamd_render call,
 * measured with HIP events on the launch stream, and the number of ray
 * segments it traced. */
int rtamd_last_render_stats(const rtamd_ctx* ctx, float* kernel_ms, uint64_t* ray_segments);

/* Device time of the dominant kernel alone in the last rtamd_render call: in PT mode the
 * persistent path-tracing kernel (all its launches when the frame is rendered in several
 * batches), without the per-pixel summation of the samples and the hand-over pass; in REF
 * mode the same as kernel_ms above.  This is the duration a kernel trace lists for that
 * kernel.  No counterpart in the reference. */
int rtamd_last_trace_kernel_ms(const rtamd_ctx* ctx, float* trace_ms);

/* The part of that time spent in the hand-over pass (pixels the production kernel
 * could not settle itself and gave to the reference-order kernel), and how many
 * pixels that was — plus, in PT mode, the single paths the path-tracing kernel
 * passed on between its launches (their time is part of kernel_ms, not of
 * handover_ms).  No counterpart in the reference. */
int rtamd_last_handover_stats(const rtamd_ctx* ctx, float* handover_ms, int32_t* pixels);

/* The guard of the production kernels' distance culling.  The reference tests a leaf slot when
 * the slot's box is entered before the closest hit found so far
 * (Assets/Shaders/RayTracingShader.compute:298,327); the production kernels visit boxes in
 * another order and cull them against the closest hit times (1 + 2^-8), which returns the
 * reference's triangle as long as no triangle's own boxes are entered, as computed, more than
 * 2^-8 (relative) behind its own computed hit distance -- rounding error, a few ulps on ordinary
 * meshes, larger on extreme slivers and huge coordinates.  Every hit the kernels compute is
 * checked against a quarter of that slack (2^-10): such a ray goes to the reference-order
 * kernel, which counts the hits of this kind it computes, per mesh.  A mesh that shows at
 * least four of them in a frame and more than one per million ray segments (a single one in a
 * billion rays is the tail of ordinary rounding) is marked: from then on (until the next
 * rtamd_scene_upload) every ray entering an instance of a marked mesh goes to the
 * reference-order kernel, and the frame in which it happened is rendered a second time before
 * rtamd_render returns (not a frame with RTAMD_FLAG_ACCUMULATE, whose samples are in the
 * running sum already).
 *   hits_near_slack  such hits in the last rtamd_render call (first pass)
 *   demoted_meshes   meshes (BLAS) of the scene currently handled in reference order
 *   rerendered       1 when the last call rendered its frame twice
 * Any pointer may be NULL.  No counterpart in the reference. */
int rtamd_last_guard_stats(const rtamd_ctx* ctx, int32_t* hits_near_slack, int32_t* demoted_meshes, int32_t* rerendered);

/* Presentation step of the reference: AddShader's `Blend SrcAlpha One`
 * (Assets/Shaders/AddShader.shader:11,43-46) onto an 8-bit UNORM target
 * (RayTracingPass.cs:58): dst = saturate(dst + src.rgb * src.a), quantised to
 * 8 bits.  src_rgba: width*height*4 floats, dst_rgba8: width*height*4 bytes,
 * both host or both device. */
int rtamd_composite_add(rtamd_ctx* ctx, const float* src_rgba, uint8_t* dst_rgba8,
                        int32_t width, int32_t height);

/* Tone-mapped presentation of an HDR framebuffer (a path-traced mean, which unlike
 * the reference's single mirror bounce routinely exceeds 1): the reference has no
 * tone map — its UNORM8 target clamps at 1 (RayTracingPass.cs:58), which is
 * RTAMD_TONEMAP_CLAMP here.  Per channel: x = src * exposure; CLAMP: x;
 * REINHARD: x / (1 + x); then saturate, then x^(1/gamma) when gamma is neither 0
 * nor 1, then round(x * 255).  Alpha is saturate(src.a) * 255.  dst is overwritten
 * (use rtamd_composite_add for the reference's additive blend).  Both buffers host
 * or both device. */
#define RTAMD_TONEMAP_CLAMP 0
#define RTAMD_TONEMAP_REINHARD 1
int rtamd_present(rtamd_ctx* ctx, const float* src_rgba, uint8_t* dst_rgba8, int32_t width,
                  int32_t height, int32_t tonemap, float exposure, float gamma);

/* Library / build identification: "rtamd 0.3 gfx950 src:<fingerprint of the sources it was built from>". */
const char* rtamd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RTAMD_H */

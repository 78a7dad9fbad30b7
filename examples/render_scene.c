/* The drop-in from plain C, end to end: load a flattened scene (the eight buffers
 * BuildAccelerateStructureSystem publishes, Assets/Script/.../BuildAccelerateStructureSystem.cs:20-33, as one
 * file), upload it, render what the reference's pass renders (RayTracingPass.cs:62-109: one dispatch of CSMain,
 * Assets/Shaders/RayTracingShader.compute:431-444) and, optionally, a path-traced frame; write the frame as .ppm /
 * .hdr / .exr and the primary ids as raw int32 pairs.
 *
 *   render_scene <scene.rtscene> <out image> [width height [spp]]      (spp 0 = the reference's image)
 *
 * The camera is the reference's sample camera (Assets/Scenes/SampleScene.unity:638-640,666-667).
 * Build: gcc -std=c99 -Iinclude examples/render_scene.c -Lraytracing_amd/lib -lrtamd -lrtamd_host -o render_scene */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtamd.h"
#include "rtamd_host.h"

static int fail(const char* what, const char* why) {
    fprintf(stderr, "render_scene: %s: %s\n", what, why ? why : "?");
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 3) return fail("usage", "render_scene <scene.rtscene> <out image> [width height [spp]]");
    const int width = argc > 4 ? atoi(argv[3]) : 640, height = argc > 4 ? atoi(argv[4]) : 360;
    const int spp = argc > 5 ? atoi(argv[5]) : 0;
    if (width < 1 || height < 1 || spp < 0) return fail("arguments", "width, height >= 1, spp >= 0");

    rtamd_host_scene* scene = rtamd_host_scene_create();
    if (!scene) return fail("rtamd_host_scene_create", "out of memory");
    if (rtamd_host_scene_load(scene, argv[1]) < 0) return fail(argv[1], rtamd_host_last_error());
    rtamd_host_buffers b;
    if (rtamd_host_scene_buffers(scene, &b) < 0) return fail("rtamd_host_scene_buffers", rtamd_host_last_error());

    rtamd_ctx* ctx = NULL;
    if (rtamd_ctx_create(0, &ctx) != RTAMD_OK) return fail("rtamd_ctx_create", rtamd_last_error_string(NULL));
    if (rtamd_scene_upload(ctx, b.tlas, b.n_tlas_nodes, b.blas_offsets, b.blas, b.n_blas_nodes, b.vertices_offsets, b.vertices, b.n_vertices,
                           b.triangles_offsets, b.triangles, b.n_triangles, b.n_blas, b.objects, b.n_objects) != RTAMD_OK)
        return fail("rtamd_scene_upload", rtamd_last_error_string(ctx));

    const float position[3] = {39.34177f, 121.19705f, 167.90202f};
    const float rotation[4] = {-0.0028800666f, 0.98602176f, -0.16570693f, -0.017148614f};
    rtamd_frame_cbuffer frame;
    memset(&frame, 0, sizeof frame);
    rtamd_host_camera(position, rotation, 58.2f, (float)width / (float)height, 0.3f, 1000.0f, &frame);

    float* rgba = (float*)calloc((size_t)width * height * 4, sizeof(float));
    int32_t* ids = (int32_t*)calloc((size_t)width * height * 2, sizeof(int32_t));
    if (!rgba || !ids) return fail("calloc", "out of memory");
    int rc;
    if (spp == 0) rc = rtamd_render(ctx, &frame, NULL, width, height, rgba, ids);      /* ext == NULL: the reference's image */
    else {
        rtamd_ext_params ext;
        memset(&ext, 0, sizeof ext);
        ext.struct_size = (uint32_t)sizeof ext;
        ext.mode = RTAMD_MODE_PT;
        ext.spp = spp;
        ext.max_bounces = 8;
        ext.seed = 1;
        rc = rtamd_render(ctx, &frame, &ext, width, height, rgba, ids);
    }
    if (rc != RTAMD_OK) return fail("rtamd_render", rtamd_last_error_string(ctx));
    float ms = 0.f;
    uint64_t rays = 0;
    rtamd_last_render_stats(ctx, &ms, &rays);
    printf("%s: %dx%d, %llu ray segments, %.3f ms on the GPU (%s)\n", spp ? "path traced" : "reference image", width, height, (unsigned long long)rays, ms,
           rtamd_version());

    if (rtamd_host_image_save(argv[2], rgba, width, height, 1) < 0) return fail(argv[2], rtamd_host_last_error());
    {
        char path[4096];
        snprintf(path, sizeof path, "%s.ids", argv[2]);
        FILE* f = fopen(path, "wb");
        if (!f || fwrite(ids, sizeof(int32_t) * 2, (size_t)width * height, f) != (size_t)width * height) return fail(path, "cannot write");
        fclose(f);
        snprintf(path, sizeof path, "%s.rgba", argv[2]);
        f = fopen(path, "wb");
        if (!f || fwrite(rgba, sizeof(float) * 4, (size_t)width * height, f) != (size_t)width * height) return fail(path, "cannot write");
        fclose(f);
    }
    free(rgba);
    free(ids);
    rtamd_ctx_destroy(ctx);
    rtamd_host_scene_destroy(scene);
    return 0;
}

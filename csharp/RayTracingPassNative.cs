// Replacement for CustomRenderPass.Execute (Assets/Script/RenderPass/RayTracingPass.cs:62-109)
// that hands the scene to librtamd through P/Invoke instead of dispatching the compute shader.
// SOURCE ONLY: this repository's build environment has no C# / Unity toolchain, so the file
// is not compiled or tested here; it shows the binding a maintainer would add.
using System;
using System.Runtime.InteropServices;
using Unity.Collections;
using Unity.Collections.LowLevel.Unsafe;
using UnityEngine;

namespace Script.RayTracing
{
    public static unsafe class RtAmd
    {
        const string Lib = "rtamd";   // librtamd.so

        [StructLayout(LayoutKind.Sequential)]
        public struct ExtParams
        {
            public uint struct_size; public int mode, spp, max_bounces; public ulong seed; public int sample_offset;
            public int tile_x, tile_y, tile_w, tile_h, shard_rank, shard_count; public uint flags;
            public IntPtr out_t, out_counters;
        }

        [DllImport(Lib)] public static extern int rtamd_ctx_create(int device, out IntPtr ctx);
        // one context over several GPUs of the node (scene replicated, frame tile-sharded, gathered with RCCL inside the library)
        [DllImport(Lib)] public static extern int rtamd_ctx_create_multi(int* deviceOrdinals, int nDevices, out IntPtr ctx);
        [DllImport(Lib)] public static extern int rtamd_multi_info(IntPtr ctx, out int nDevices, out int rcclRanks, out float lastExchangeMs);
        [DllImport(Lib)] public static extern int rtamd_ctx_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern IntPtr rtamd_last_error_string(IntPtr ctx);
        [DllImport(Lib)] public static extern int rtamd_scene_upload(IntPtr ctx,
            void* tlas, int nTlas, void* blasOffsets, void* blas, int nBlasNodes,
            void* vtxOffsets, void* vertices, int nVertices, void* triOffsets, void* triangles, int nTriangles,
            int nBlas, void* objects, int nObjects);
        [DllImport(Lib)] public static extern int rtamd_scene_refresh_objects(IntPtr ctx, void* tlas, int nTlas, void* objects, int nObjects);
        [DllImport(Lib)] public static extern int rtamd_set_tlas_build(IntPtr ctx, int mode);      // 0 auto, 1 host, 2 device
        [DllImport(Lib)] public static extern int rtamd_set_blas_build(IntPtr ctx, int mode);      // 0 auto (host), 1 host, 2 device: who builds the meshes' trees at the next upload
        [DllImport(Lib)] public static extern int rtamd_last_upload_stats(IntPtr ctx, out float uploadMs, out float treeBuildMs, out int meshesBuiltOnDevice);
        [DllImport(Lib)] public static extern int rtamd_tlas_info(IntPtr ctx, out int nodes, out int levels, out int builtOnDevice);
        [DllImport(Lib)] public static extern int rtamd_sky_upload(IntPtr ctx, void* rgba32f, int w, int h);
        [DllImport(Lib)] public static extern int rtamd_render(IntPtr ctx, FrameCBuffer* frame, ExtParams* ext, int w, int h, void* outRgba, void* outPrimId);
        // rtamd_render without any device allocation (outputs must be device memory): for serving loops and hosts that budget device memory themselves
        [DllImport(Lib)] public static extern int rtamd_render_noalloc(IntPtr ctx, FrameCBuffer* frame, ExtParams* ext, int w, int h, void* outRgbaDevice, void* outPrimIdDevice);
        // rtamd_render in two halves (device-memory outputs, up to two frames in flight): begin queues the frame and returns, wait completes the oldest one
        [DllImport(Lib)] public static extern int rtamd_render_begin(IntPtr ctx, FrameCBuffer* frame, ExtParams* ext, int w, int h, void* outRgbaDevice, void* outPrimIdDevice);
        [DllImport(Lib)] public static extern int rtamd_render_wait(IntPtr ctx);
        [DllImport(Lib)] public static extern int rtamd_accum_reset(IntPtr ctx);
        // the running sum of RTAMD_FLAG_ACCUMULATE frames, out and back in (stop a progressive render and continue it later)
        [DllImport(Lib)] public static extern int rtamd_accum_download(IntPtr ctx, void* sums, int w, int h);
        [DllImport(Lib)] public static extern int rtamd_accum_upload(IntPtr ctx, void* sums, int w, int h);
        // the guard of the production kernels' distance culling: hits that came near its slack in the last render, meshes now handled in reference order, frame rendered twice
        [DllImport(Lib)] public static extern int rtamd_last_guard_stats(IntPtr ctx, out int hitsNearSlack, out int demotedMeshes, out int rerendered);
        [DllImport(Lib)] public static extern int rtamd_last_render_stats(IntPtr ctx, out float kernelMs, out ulong raySegments);
        [DllImport(Lib)] public static extern int rtamd_composite_add(IntPtr ctx, void* srcRgba32f, void* dstRgba8, int w, int h);
        [DllImport(Lib)] public static extern int rtamd_present(IntPtr ctx, void* srcRgba32f, void* dstRgba8, int w, int h, int tonemap, float exposure, float gamma);
    }

    // In BuildAccelerateStructureSystem.OnUpdate the eight ComputeBuffer.SetData calls (:124-158)
    // become one call on the NativeArrays the jobs already filled:
    //
    //   RtAmd.rtamd_scene_upload(ctx,
    //       TLAS.Nodes.GetUnsafeReadOnlyPtr(), TLAS.NodeCount,
    //       objectsBVHOffsetBuffer.GetUnsafeReadOnlyPtr(),
    //       buildComputeBufferJob.blasBVHBuffer.GetUnsafeReadOnlyPtr(), totalBVHNodeCount,
    //       objectsVerticesOffsetBuffer.GetUnsafeReadOnlyPtr(),
    //       buildComputeBufferJob.blasVerticesBuffer.GetUnsafeReadOnlyPtr(), totalVerticesCount,
    //       objectsTrianglesOffsetBuffer.GetUnsafeReadOnlyPtr(),
    //       buildComputeBufferJob.blasTriangleBuffer.GetUnsafeReadOnlyPtr(), totalTrianglesCount,
    //       blasCount,
    //       buildPerObjectPropertyComputeBufferJob.objectsProperty.GetUnsafeReadOnlyPtr(), ObjectsCount);
    //
    // (or rtamd_scene_refresh_objects on frames where only transforms / materials changed).

    public unsafe class NativeRayTracer : IDisposable
    {
        IntPtr ctx;
        Texture2D target;
        NativeArray<float> pixels;

        public NativeRayTracer(int device = 0)
        {
            if (RtAmd.rtamd_ctx_create(device, out ctx) != 0)
                throw new InvalidOperationException(Marshal.PtrToStringAnsi(RtAmd.rtamd_last_error_string(IntPtr.Zero)));
        }

        // Every listed GPU renders its share of each frame; the frame comes back whole.
        public NativeRayTracer(int[] devices)
        {
            fixed (int* d = devices)
                if (RtAmd.rtamd_ctx_create_multi(d, devices.Length, out ctx) != 0)
                    throw new InvalidOperationException(Marshal.PtrToStringAnsi(RtAmd.rtamd_last_error_string(IntPtr.Zero)));
        }

        // Body of Execute between the ObjectsCount guard (:68) and the Blit (:105).
        public Texture2D Render(Camera camera, int width, int height, int reflectionNum)
        {
            FrameCBuffer cbuffer = new FrameCBuffer
            {
                CameraInverseProjection = camera.projectionMatrix.inverse,
                CameraToWorld = camera.cameraToWorldMatrix,
                PixelOffset = new Unity.Mathematics.float2(0, 0),
                Seed = UnityEngine.Random.value,
                ReflectionNum = reflectionNum
            };
            if (target == null || target.width != width || target.height != height)
            {
                target = new Texture2D(width, height, TextureFormat.RGBAFloat, false);
                if (pixels.IsCreated) pixels.Dispose();
                pixels = new NativeArray<float>(width * height * 4, Allocator.Persistent);
            }
            int rc = RtAmd.rtamd_render(ctx, &cbuffer, null, width, height, pixels.GetUnsafePtr(), null);   // ext == null: REF mode
            if (rc != 0) Debug.LogError(Marshal.PtrToStringAnsi(RtAmd.rtamd_last_error_string(ctx)));
            target.SetPixelData(pixels, 0);
            target.Apply(false);
            return target;   // cmd.Blit(target, source, pass.settings.AddMaterial) as before
        }

        public void Dispose()
        {
            if (ctx != IntPtr.Zero) RtAmd.rtamd_ctx_destroy(ctx);
            ctx = IntPtr.Zero;
            if (pixels.IsCreated) pixels.Dispose();
        }
    }
}

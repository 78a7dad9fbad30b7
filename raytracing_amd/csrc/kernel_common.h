// Device-side structs and the shading / sampling code shared by both traversal kernels.
#pragma once

#include "device_math.h"
#include "rtamd.h"

namespace rt {

// The reference's buffers exactly as uploaded (RayTracingShader.compute:96-124).
struct RefScene {
    const rtamd_node* tlas;
    const int32_t* blasOffsets;
    const rtamd_node* blas;
    const int32_t* vtxOffsets;
    const rtamd_vertex* verts;
    const int32_t* triOffsets;
    const rtamd_int3* tris;
    const rtamd_object_property* objects;
    // The guard of the production kernels' distance culling (RT_GUARD_FACTOR, kernel_fast.h), kept by the
    // reference-order kernels, which trace what the production kernels hand over: *guardCount counts the hits
    // whose own boxes are entered, as computed, more than 2^-10 behind the hit's computed distance,
    // guardBlas[b] those that lay in BLAS b.  Both null in a render the caller asked to be in reference order.
    int32_t* guardCount;
    int32_t* guardBlas;
};

// A hit whose own boxes are entered, as computed, more than this factor behind the hit's computed distance trips
// the guard of the production kernels' distance culling (see RT_CULL_RELAX in kernel_fast.h).
#ifndef RT_GUARD_FACTOR
#define RT_GUARD_FACTOR 1.0009765625f   /* 1 + 2^-10 */
#endif

struct Sky {
    const float4* texels;
    int32_t w, h;
    float invW, invH;       // 1.0f / w, 1.0f / h
};

struct DevCounters {
    unsigned long long rays, nodes, tris, instances, hits, maxBlasStack, maxTlasStack, overflow;
};

struct RenderArgs {
    rtamd_frame_cbuffer cb;
    int32_t width, height;
    int32_t mode, spp, maxBounces;
    uint32_t seedHi, seedLo;
    int32_t sampleOffset;
    int32_t tileX, tileY, tileW, tileH;   // already clipped to the image
    int32_t shardRank, shardCount;
    uint32_t flags;
    float4* outRgba;        // may be null
    int32_t* outIds;        // may be null, 2 per pixel
    float* outT;            // may be null
    float4* accum;          // running sums (ACCUMULATE), may be null
    DevCounters* counters;  // may be null
    unsigned long long* rayCount;   // always valid
    int32_t* errorFlag;     // always valid
    // pixels the production kernel hands over to the reference-order kernel
    uint32_t* redoMask;     // one bit per pixel
    int32_t* redoList;      // pixel indices
    int32_t* redoCount;
    int32_t redoCap;
};

struct Ray { V3 origin, direction, invDir; };

__device__ __forceinline__ Ray makeRay(V3 o, V3 d) {          // CreateRay, .compute:135-143
    Ray r;
    r.origin = o;
    r.direction = d;
    r.invDir = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    return r;
}

// (zero: the z of float4(uv, 0, 1); a caller may pass a zero the compiler cannot see through, see kPool)
__device__ __forceinline__ Ray cameraRay(const rtamd_frame_cbuffer& cb, float u, float v, float zero = 0.0f) {   // .compute:145-157
    V3 origin = v3(cb.CameraToWorld[12], cb.CameraToWorld[13], cb.CameraToWorld[14]);
    const float* ip = cb.CameraInverseProjection;
    V3 d = v3(((ip[0] * u + ip[4] * v) + ip[8] * zero) + ip[12],
              ((ip[1] * u + ip[5] * v) + ip[9] * zero) + ip[13],
              ((ip[2] * u + ip[6] * v) + ip[10] * zero) + ip[14]);
    d = mulDir(cb.CameraToWorld, d);
    d = normalize(d);
    return makeRay(origin, d);
}

// The same from the camera block of a kernel whose FIRST argument is that block (by value): read from the argument
// block itself, through a pointer the compiler cannot see through, at the place where a camera ray is made.  Read
// as an ordinary argument, the block's 36 scalars sit in scalar registers for the whole kernel, and — there being
// too few of those in the traversal kernels — are shuttled through lanes of a vector register at every use.
typedef const __attribute__((address_space(4))) rtamd_frame_cbuffer* CameraArgPtr;
__device__ __forceinline__ CameraArgPtr cameraArg() {
    CameraArgPtr p = (CameraArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
__device__ __forceinline__ Ray cameraRayArg(CameraArgPtr c, float u, float v, float zero) {      // cameraRay, operation for operation
    V3 d = v3(((c->CameraInverseProjection[0] * u + c->CameraInverseProjection[4] * v) + c->CameraInverseProjection[8] * zero) + c->CameraInverseProjection[12],
              ((c->CameraInverseProjection[1] * u + c->CameraInverseProjection[5] * v) + c->CameraInverseProjection[9] * zero) + c->CameraInverseProjection[13],
              ((c->CameraInverseProjection[2] * u + c->CameraInverseProjection[6] * v) + c->CameraInverseProjection[10] * zero) + c->CameraInverseProjection[14]);
    d = v3((c->CameraToWorld[0] * d.x + c->CameraToWorld[4] * d.y) + c->CameraToWorld[8] * d.z,
           (c->CameraToWorld[1] * d.x + c->CameraToWorld[5] * d.y) + c->CameraToWorld[9] * d.z,
           (c->CameraToWorld[2] * d.x + c->CameraToWorld[6] * d.y) + c->CameraToWorld[10] * d.z);
    return makeRay(v3(c->CameraToWorld[12], c->CameraToWorld[13], c->CameraToWorld[14]), normalize(d));
}

// uv = (id.xy + offset) / (w,h) * 2 - 1.  .compute:439
__device__ __forceinline__ float pixelToNdc(int p, float offset, int extent) {
    return ((float)p + offset) / (float)extent * 2.0f - 1.0f;
}

// Culled Möller–Trumbore.  .compute:198-231.  e1 = v1 - v0, e2 = v2 - v0.
// twoSided (a ray travelling inside a dielectric, extension): either sign of the determinant is
// accepted, only |det| < EPSILON is rejected, and `back` reports det < 0; for det >= EPSILON the
// operations are the reference's.
__device__ __forceinline__ bool trisect(V3 org, V3 dir, V3 v0, V3 e1, V3 e2, float& T, float& U, float& V, bool twoSided, bool& back) {
    V3 pvec = cross(dir, e2);
    float det = dot(e1, pvec);
    if (det < RT_EPSILON) {
        if (!twoSided || det > -RT_EPSILON) return false;
    }
    back = twoSided && det < 0.0f;
    float inv_det = 1.0f / det;
    V3 tvec = org - v0;
    float u = dot(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    V3 qvec = cross(tvec, e1);
    float v = dot(dir, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot(e2, qvec) * inv_det;
    if (t > 0) { T = t; U = u; V = v; return true; }
    return false;
}

// Bilinear, repeat, mip 0.  .compute:409
__device__ __forceinline__ V3 sampleSky(const Sky& sky, float u, float v) {
    if (!sky.texels) return v3(0, 0, 0);
    float W = (float)sky.w, H = (float)sky.h;
    float x = u * W - 0.5f;
    float y = v * H - 0.5f;
    float x0f = __builtin_floorf(x), y0f = __builtin_floorf(y);
    float fx = x - x0f, fy = y - y0f;
    float xw = x0f - __builtin_floorf(x0f * sky.invW) * W;
    float yw = y0f - __builtin_floorf(y0f * sky.invH) * H;
    int x0 = (int)xw, y0 = (int)yw;
    if (x0 < 0) x0 = 0;
    if (x0 >= sky.w) x0 = sky.w - 1;
    if (y0 < 0) y0 = 0;
    if (y0 >= sky.h) y0 = sky.h - 1;
    int x1 = x0 + 1 == sky.w ? 0 : x0 + 1;
    int y1 = y0 + 1 == sky.h ? 0 : y0 + 1;
    float4 p00 = sky.texels[(size_t)y0 * sky.w + x0];
    float4 p10 = sky.texels[(size_t)y0 * sky.w + x1];
    float4 p01 = sky.texels[(size_t)y1 * sky.w + x0];
    float4 p11 = sky.texels[(size_t)y1 * sky.w + x1];
    float gx = 1.0f - fx, gy = 1.0f - fy;
    V3 r;
    r.x = (p00.x * gx + p10.x * fx) * gy + (p01.x * gx + p11.x * fx) * fy;
    r.y = (p00.y * gx + p10.y * fx) * gy + (p01.y * gx + p11.y * fx) * fy;
    r.z = (p00.z * gx + p10.z * fx) * gy + (p01.z * gx + p11.z * fx) * fy;
    return r;
}

// Miss branch of Shade.  .compute:402-410
__device__ __forceinline__ V3 shadeMiss(const Sky& sky, V3 dir) {
    float theta = acosDecl(dir.y) / -RT_PI;
    float phi = atan2Decl(dir.x, -dir.z) / -RT_PI * 0.5f;
    return sampleSky(sky, phi, theta);
}

__device__ __forceinline__ float energyOf(V3 c) {             // .compute:45-48
    const float third = 1.0f / 3.0f;
    return (c.x * third + c.y * third) + c.z * third;
}

// Uniform hemisphere sample around `normal`.  .compute:33-43, 67-76
__device__ __forceinline__ V3 sampleHemisphere(V3 normal, float u1, float u2) {
    float cosTheta = u1;
    float sinTheta = __builtin_sqrtf(fmax_(0.0f, 1.0f - cosTheta * cosTheta));
    float sphi, cphi;
    sincos2piDecl(u2, sphi, cphi);
    V3 ts = v3(cphi * sinTheta, sphi * sinTheta, cosTheta);
    V3 helper = v3(1, 0, 0);
    if (__builtin_fabsf(normal.x) > 0.99f) helper = v3(0, 0, 1);
    V3 tangent = normalize(cross(normal, helper));
    V3 binormal = normalize(cross(normal, tangent));
    return ((ts.x * tangent) + (ts.y * binormal)) + (ts.z * normal);
}

// One scattering event of the path tracer: the specular / diffuse roulette the shader's
// Shade sketches (.compute:391-399).  Returns false when the path ends.
__device__ __forceinline__ bool scatter(V3 albedoIn, V3 spec, V3 normal, V3 inDir, uint32_t key, uint32_t bounce,
                                        V3& energy, V3& outDir) {
    V3 alb = v3(fmin_(1.0f - spec.x, albedoIn.x), fmin_(1.0f - spec.y, albedoIn.y), fmin_(1.0f - spec.z, albedoIn.z));
    float specChance = energyOf(spec);
    float diffChance = energyOf(alb);
    float sum = specChance + diffChance;
    if (!(sum > 0.0f)) return false;
    specChance = specChance / sum;
    diffChance = diffChance / sum;
    float roulette = rngDraw(key, 2u + 3u * bounce);
    if (roulette < specChance) {
        outDir = reflect(inDir, normal);
        energy = energy * (((1.0f / specChance) * spec) * saturate(dot(normal, outDir)));
    } else {
        float u1 = rngDraw(key, 3u + 3u * bounce);
        float u2 = rngDraw(key, 4u + 3u * bounce);
        outDir = sampleHemisphere(normal, u1, u2);
        energy = energy * ((((1.0f / diffChance) * 2.0f) * alb) * saturate(dot(normal, outDir)));
    }
    return energy.x > 0.0f || energy.y > 0.0f || energy.z > 0.0f;
}

// What the path tracer needs of a hit surface.  normal already points against the side that was hit.
struct Surf { V3 position, normal, albedo, specular, emission; float albedoW, specularW; bool back; int32_t object, triangle; };

// Next segment of a path at surface `s`: false when the path ends.  DIEL: glass (specular.w ==
// RTAMD_MATERIAL_DIELECTRIC, index of refraction albedo.w, tint albedo.xyz) reflects with Schlick's
// Fresnel probability and refracts otherwise; `inside` follows the path into and out of the glass.
template <bool DIEL>
__device__ __forceinline__ bool scatterSurface(const Surf& s, V3 inDir, uint32_t key, uint32_t bounce, V3& energy, V3& outOrigin, V3& outDir,
                                               bool& inside) {
    if (DIEL && s.specularW == RTAMD_MATERIAL_DIELECTRIC) {
        const float ior = s.albedoW;
        const float eta = s.back ? ior : 1.0f / ior;
        const float cosi = saturate(-dot(inDir, s.normal));
        float r0 = (1.0f - ior) / (1.0f + ior);
        r0 = r0 * r0;
        const float sin2t = (eta * eta) * (1.0f - cosi * cosi);
        float F = 1.0f;
        if (!(sin2t > 1.0f)) {
            const float m = 1.0f - cosi;
            const float m2 = m * m;
            F = r0 + (1.0f - r0) * ((m2 * m2) * m);
        }
        const float roulette = rngDraw(key, 2u + 3u * bounce);
        if (roulette < F) {
            outDir = reflect(inDir, s.normal);
            outOrigin = s.position + s.normal * 0.001f;
        } else {
            const float cost = __builtin_sqrtf(fmax_(0.0f, 1.0f - sin2t));
            outDir = normalize((eta * inDir) + ((eta * cosi - cost) * s.normal));
            outOrigin = s.position + s.normal * -0.001f;
            energy = energy * s.albedo;
            inside = !s.back;
        }
        return energy.x > 0.0f || energy.y > 0.0f || energy.z > 0.0f;
    }
    if (!scatter(s.albedo, s.specular, s.normal, inDir, key, bounce, energy, outDir)) return false;
    outOrigin = s.position + s.normal * 0.001f;
    return true;
}

__device__ __forceinline__ bool pixelSelected(const RenderArgs& a, int x, int y) {
    if (x < a.tileX || y < a.tileY || x >= a.tileX + a.tileW || y >= a.tileY + a.tileH) return false;
    if (a.shardCount > 1) {
        int tilesX = (a.width + 31) / 32;
        int tile = (y / 32) * tilesX + (x / 32);
        if (tile % a.shardCount != a.shardRank) return false;
    }
    return true;
}

// Final write of a path-traced pixel: plain mean, running mean with the context's
// accumulation buffer, or the raw sum.
__device__ __forceinline__ void writePathPixel(const RenderArgs& a, size_t pix, V3 sum) {
    float n = (float)a.spp;
    if (a.accum) {
        float4 acc = a.accum[pix];
        acc.x += sum.x; acc.y += sum.y; acc.z += sum.z; acc.w += n;
        a.accum[pix] = acc;
        sum = v3(acc.x, acc.y, acc.z);
        n = acc.w;
    }
    if (a.outRgba) {
        if (a.flags & RTAMD_FLAG_OUTPUT_SUM) a.outRgba[pix] = make_float4(sum.x, sum.y, sum.z, n);
        else a.outRgba[pix] = make_float4(sum.x / n, sum.y / n, sum.z / n, 1.0f);
    }
}

}  // namespace rt

// Arithmetic of the ray-tracing kernels, spelled out operation by operation.
//
// The reference shader (Assets/Shaders/RayTracingShader.compute) leaves the bit-level
// meaning of dot / mul / normalize / reflect / min / max / acos / atan2 / sin / cos to
// the HLSL compiler and the GPU vendor.  This backend fixes one meaning — IEEE-754
// binary32, round to nearest even, no fused multiply-add, correctly rounded division
// and square root, subnormals kept — so that a CPU evaluation and the gfx950
// evaluation of the same ray give the same bits.  Every `/` of the shader text is a true division
// (uv / float2(width,height) :439, / -PI :407-408, specChance /= sum :395) and normalize is
// v / sqrt(dot(v,v)); replacing them by multiplications with rounded reciprocals was measured to move
// a few dozen primary-hit ids per 1080p frame, so it is not done.  This file must be compiled with
// -ffp-contract=off; hipcc's default correctly rounded f32 divide/sqrt is relied on.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 v3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 normalize(V3 v) {          // v / length(v): three true divisions
    float l = __builtin_sqrtf(dot(v, v));
    return {v.x / l, v.y / l, v.z / l};
}
__device__ __forceinline__ V3 reflect(V3 i, V3 n) { return i - (2.0f * dot(n, i)) * n; }
__device__ __forceinline__ float fmin_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ float saturate(float x) { return fmin_(fmax_(x, 0.0f), 1.0f); }

// mul(float4x4 M, float4(v, 1)).xyz and mul(M, float4(v, 0)).xyz, M column-major in memory
// (4 consecutive floats = one column), rows summed left to right.
__device__ __forceinline__ V3 mulPoint(const float* m, V3 v) {
    return {((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12],
            ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13],
            ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14]};
}
__device__ __forceinline__ V3 mulDir(const float* m, V3 v) {
    return {(m[0] * v.x + m[4] * v.y) + m[8] * v.z,
            (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
            (m[2] * v.x + m[6] * v.y) + m[10] * v.z};
}

#define RT_PI 3.14159265f          /* RayTracingShader.compute:24 */
#define RT_EPSILON 1e-8f           /* RayTracingShader.compute:27 */
#define RT_INF __builtin_inff()

// acos on [-1,1]: sqrt(1-|x|) * P7(|x|), mirrored for x < 0.
__device__ __forceinline__ float acosDecl(float x) {
    float a = __builtin_fabsf(x);
    float p = -0.0012624911f;
    p = p * a + 0.0066700901f;
    p = p * a + -0.0170881256f;
    p = p * a + 0.0308918810f;
    p = p * a + -0.0501743046f;
    p = p * a + 0.0889789874f;
    p = p * a + -0.2145988016f;
    p = p * a + 1.5707963050f;
    float r = __builtin_sqrtf(fmax_(1.0f - a, 0.0f)) * p;
    return x < 0.0f ? RT_PI - r : r;
}

// atan2(y, x): odd polynomial for atan on [0,1], unfolded over the octants.
__device__ __forceinline__ float atan2Decl(float y, float x) {
    float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    float mx = fmax_(ax, ay), mn = fmin_(ax, ay);
    float a = mx > 0.0f ? mn / mx : 0.0f;
    float s = a * a;
    float p = 0.0028662257f;
    p = p * s + -0.0161657367f;
    p = p * s + 0.0429096138f;
    p = p * s + -0.0752896400f;
    p = p * s + 0.1065626393f;
    p = p * s + -0.1420889944f;
    p = p * s + 0.1999355085f;
    p = p * s + -0.3333314528f;
    p = p * s + 1.0f;
    float r = a * p;
    if (ay > ax) r = 1.57079633f - r;
    if (x < 0.0f) r = RT_PI - r;
    if (y < 0.0f) r = -r;
    return r;
}

// sin and cos of 2*pi*u, u in [0,1): quadrant split, Taylor polynomials on [0, pi/2).
__device__ __forceinline__ void sincos2piDecl(float u, float& s, float& c) {
    float t = u * 4.0f;
    float q = __builtin_floorf(t);
    float a = (t - q) * 1.57079633f;
    float a2 = a * a;
    float sp = -2.50521084e-8f;
    sp = sp * a2 + 2.75573192e-6f;
    sp = sp * a2 + -1.98412698e-4f;
    sp = sp * a2 + 8.33333333e-3f;
    sp = sp * a2 + -1.66666667e-1f;
    sp = sp * a2 + 1.0f;
    float sn = a * sp;
    float cp = 2.08767570e-9f;
    cp = cp * a2 + -2.75573192e-7f;
    cp = cp * a2 + 2.48015873e-5f;
    cp = cp * a2 + -1.38888889e-3f;
    cp = cp * a2 + 4.16666667e-2f;
    cp = cp * a2 + -0.5f;
    cp = cp * a2 + 1.0f;
    int qi = (int)q & 3;
    if (qi == 0) { s = sn; c = cp; }
    else if (qi == 1) { s = cp; c = -sn; }
    else if (qi == 2) { s = -sn; c = -cp; }
    else { s = -cp; c = sn; }
}

// Counter-based random numbers: every draw is a pure function of
// (seed, pixel, sample, dimension), so the image does not depend on how pixels or
// samples are distributed over lanes, waves or GPUs.
__device__ __forceinline__ uint32_t pcg(uint32_t x) {
    uint32_t s = x * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
__device__ __forceinline__ uint32_t rngKey(uint32_t seedHi, uint32_t seedLo, uint32_t pixel, uint32_t sample) {
    uint32_t k = pcg(seedHi);
    k = pcg(k ^ seedLo);
    k = pcg(k ^ sample);
    k = pcg(k ^ pixel);
    return k;
}
__device__ __forceinline__ float rngDraw(uint32_t key, uint32_t dim) {
    return (float)(pcg(key + dim * 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
}

}  // namespace rt

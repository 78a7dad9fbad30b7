// rt_oracle.cpp — CPU restatement of the reference ray-tracing compute path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library includes, links or
// calls this file; it is the checker the GPU kernels are compared against (and
// the scalar CPU baseline that bench.py times next to the GPU number).
//
// PARITY UNPINNED: the reference ships no tests, golden images or known-answer
// vectors, and neither its HLSL kernel nor its C# host code can be compiled in
// this environment (no Unity, dotnet, mono, dxc).  This file therefore follows
// the reference *text* line by line and DECLARES the floating-point semantics
// the HLSL compiler would otherwise choose:
//   * IEEE-754 binary32, round-to-nearest-even, no fused multiply-add
//     (build with -ffp-contract=off), subnormals kept;
//   * dot(a,b)   = (a.x*b.x + a.y*b.y) + a.z*b.z
//   * mul(M,v)   = per row ((m0*v.x + m1*v.y) + m2*v.z) [+ m3 when w == 1]
//   * normalize  = v / sqrt(dot(v,v))  (three true divisions, true sqrt; HLSL's own form is v * rsqrt)
//   * every `/` in the shader text is a true division (image size :439, -PI :407-408, roulette sum :395)
//   * reflect    = i - (2*dot(n,i))*n
//   * min / max  = fminf / fmaxf (a NaN operand is ignored)
//   * 1.#INF     = +infinity
//   * acos, atan2, sin, cos = the fixed polynomials below (only +,-,*,/,sqrt,floor)
//   * texture SampleLevel = fp32 bilinear, texel centres at +0.5, repeat wrap.
//
// Files followed (relative to the reference checkout):
//   Assets/Shaders/RayTracingShader.compute   (cited per function below)
//   Assets/Script/Common/RayTraceMaterial.cs:6-35               (struct layouts)
//   Assets/Script/BVH/BoundingVolumeHierarchyBuilder.cs:31-74,844-880 (Node, CheckIntegrity)
//   Assets/Script/BVH/Base/Math/FourTransposedAabbs.cs:6-64
//   Assets/Script/RenderPass/RayTracingPass.cs:54-109           (dispatch, cbuffer)
//
// Build: g++ -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------- layouts --
struct Node {               // BoundingVolumeHierarchyBuilder.cs:31-37 / .compute:80-91
    float Lx[4], Hx[4], Ly[4], Hy[4], Lz[4], Hz[4];
    int32_t Data[4];
    int32_t Flags;
};
struct Vertex { float position[3]; float normal[3]; };   // RayTraceMaterial.cs:6-10
struct Int3 { int32_t x, y, z; };                          // .compute:109
struct ObjectProperty {                                    // RayTraceMaterial.cs:19-25
    int32_t BLASIndex;
    float albedo[4], specular[4], emission[4];
    float WorldToLocal[16], LocalToWorld[16];
};
struct FrameCBuffer {                                      // RayTraceMaterial.cs:28-35
    float CameraToWorld[16], CameraInverseProjection[16];
    float PixelOffset[2];
    float Seed;
    int32_t ReflectionNum;
};
static_assert(sizeof(Node) == 116, "Node stride (BuildAccelerateStructureSystem.cs:125)");
static_assert(sizeof(Vertex) == 24, "Vertex stride (:131)");
static_assert(sizeof(Int3) == 12, "int3 stride (:137)");
static_assert(sizeof(ObjectProperty) == 180, "PerObjectProperty stride (:144)");
static_assert(sizeof(FrameCBuffer) == 144, "FrameCBuffer size (RayTracingPass.cs:83)");

struct Scene {
    const Node* tlas; int32_t nTlas;
    const int32_t* blasOffsets;
    const Node* blas; int32_t nBlasNodes;
    const int32_t* vtxOffsets;
    const Vertex* verts; int32_t nVerts;
    const int32_t* triOffsets;
    const Int3* tris; int32_t nTris;
    int32_t nBlas;
    const ObjectProperty* objects; int32_t nObjects;
    const float* sky; int32_t skyW, skyH;
};

struct Params {
    int32_t mode;            // 0 REF, 1 PT
    int32_t spp;
    int32_t max_bounces;
    uint64_t seed;
    int32_t sample_offset;
    int32_t tile_x, tile_y, tile_w, tile_h;
    int32_t shard_rank, shard_count;
    uint32_t flags;          // 1 accumulate(ignored here) 2 no-jitter 8 output-sum 64 dielectrics
    int32_t nthreads;
};

struct Counters {
    uint64_t rays, nodes, tris, instances, hits, max_blas_stack, max_tlas_stack, overflow;
};

struct float3 { float x, y, z; };
static inline float3 f3(float x, float y, float z) { return {x, y, z}; }
static inline float3 operator+(float3 a, float3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline float3 operator-(float3 a, float3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline float3 operator*(float3 a, float3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline float3 operator*(float s, float3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline float3 operator*(float3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline float dot(float3 a, float3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float3 cross(float3 a, float3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Every `/` of the shader text is a true IEEE division (uv / float2(width,height) .compute:439,
// / -PI :407-408, specChance /= sum :395) and normalize(v) = v / sqrt(dot(v,v)).
// RT_ORACLE_RECIPROCAL_DIVISIONS builds the other reading — a division by a constant or per-frame
// value is a multiplication by its rounded reciprocal and normalize is v * (1 / sqrt(dot)), the forms
// an HLSL compiler may emit — only to COUNT how many primary ids, t values and 8-bit REF pixels the
// choice moves (tests/test_oracle.py): a few dozen primary ids per 1080p frame, which is why the
// literal reading is the definition and the kernels follow it.
static inline float3 normalize(float3 v) {
#ifndef RT_ORACLE_RECIPROCAL_DIVISIONS
    float l = std::sqrt(dot(v, v));
    return {v.x / l, v.y / l, v.z / l};
#else
    float r = 1.0f / std::sqrt(dot(v, v));
    return {v.x * r, v.y * r, v.z * r};
#endif
}
// fminf / fmaxf semantics (the non-NaN operand wins) without the libm call.
static inline float ofmin(float a, float b) { float r = a < b ? a : b; return b != b ? a : r; }
static inline float ofmax(float a, float b) { float r = a > b ? a : b; return b != b ? a : r; }
static inline float3 reflect(float3 i, float3 n) { return i - (2.0f * dot(n, i)) * n; }
static inline float saturate(float x) { return ofmin(ofmax(x, 0.0f), 1.0f); }

// mul(float4x4 M, float4(v, w)).xyz with M column-major in memory.
static inline float3 mulPoint(const float* m, float3 v) {
    return {((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12],
            ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13],
            ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14]};
}
static inline float3 mulDir(const float* m, float3 v) {
    return {(m[0] * v.x + m[4] * v.y) + m[8] * v.z,
            (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
            (m[2] * v.x + m[6] * v.y) + m[10] * v.z};
}

static const float PI = 3.14159265f;       // .compute:24
static const float NEG_INV_PI = -0.318309886f;   // 1 / -PI
static const float EPSILON = 1e-8f;        // .compute:27
static const float INF = std::numeric_limits<float>::infinity();

// ------------------------------------------------- declared transcendental --
// acos on [-1,1]: sqrt(1-|x|) * P7(|x|), reflected for x < 0 (Abramowitz & Stegun 4.4.46).
static inline float acosDecl(float x) {
    float a = std::fabs(x);
    float p = -0.0012624911f;
    p = p * a + 0.0066700901f;
    p = p * a + -0.0170881256f;
    p = p * a + 0.0308918810f;
    p = p * a + -0.0501743046f;
    p = p * a + 0.0889789874f;
    p = p * a + -0.2145988016f;
    p = p * a + 1.5707963050f;
    float r = std::sqrt(ofmax(1.0f - a, 0.0f)) * p;
    return x < 0.0f ? PI - r : r;
}
// atan2(y,x): atan on [0,1] by the odd polynomial of A&S 4.4.49, octant-unfolded.
static inline float atan2Decl(float y, float x) {
    float ax = std::fabs(x), ay = std::fabs(y);
    float mx = ofmax(ax, ay), mn = ofmin(ax, ay);
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
    if (x < 0.0f) r = PI - r;
    if (y < 0.0f) r = -r;
    return r;
}
// sin and cos of 2*pi*u for u in [0,1): quadrant split then Taylor polynomials on [0,pi/2).
static inline void sincos2piDecl(float u, float& s, float& c) {
    float t = u * 4.0f;
    float q = std::floor(t);
    float a = (t - q) * 1.57079633f;
    float a2 = a * a;
    float sp = -2.50521084e-8f;                 // -1/11!
    sp = sp * a2 + 2.75573192e-6f;              //  1/9!
    sp = sp * a2 + -1.98412698e-4f;             // -1/7!
    sp = sp * a2 + 8.33333333e-3f;              //  1/5!
    sp = sp * a2 + -1.66666667e-1f;             // -1/3!
    sp = sp * a2 + 1.0f;
    float sn = a * sp;
    float cp = 2.08767570e-9f;                  //  1/12!
    cp = cp * a2 + -2.75573192e-7f;             // -1/10!
    cp = cp * a2 + 2.48015873e-5f;              //  1/8!
    cp = cp * a2 + -1.38888889e-3f;             // -1/6!
    cp = cp * a2 + 4.16666667e-2f;              //  1/4!
    cp = cp * a2 + -0.5f;
    cp = cp * a2 + 1.0f;
    int qi = (int)q & 3;
    if (qi == 0) { s = sn; c = cp; }
    else if (qi == 1) { s = cp; c = -sn; }
    else if (qi == 2) { s = -sn; c = -cp; }
    else { s = -cp; c = sn; }
}

// ------------------------------------------------------------------- RNG ---
// Counter-based: every draw is a pure function of (seed, pixel, sample, dimension).
// Replaces the shader's stateful sin-hash `rand` (.compute:50-55), whose result depends
// on evaluation order and on the vendor's sin().
static inline uint32_t pcg(uint32_t x) {
    uint32_t s = x * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
    return (w >> 22) ^ w;
}
static inline uint32_t rngKey(uint64_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t k = pcg((uint32_t)(seed >> 32));
    k = pcg(k ^ (uint32_t)seed);
    k = pcg(k ^ sample);
    k = pcg(k ^ pixel);
    return k;
}
static inline float rngDraw(uint32_t key, uint32_t dim) {
    return (float)(pcg(key + dim * 0x9E3779B9u) >> 8) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------ rays ---
struct Ray { float3 origin, direction, energy, invDir; };     // .compute:127-133

static inline Ray CreateRay(float3 origin, float3 direction) { // .compute:135-143
    Ray r;
    r.origin = origin;
    r.direction = direction;
    r.energy = f3(1.0f, 1.0f, 1.0f);
    r.invDir = f3(1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z);
    return r;
}

// CSMain: uv = (id.xy + _PixelOffset) / (w,h) * 2 - 1.  .compute:439
static inline float pixelToNdc(int p, float offset, int extent) {
#ifndef RT_ORACLE_RECIPROCAL_DIVISIONS
    return ((float)p + offset) / (float)extent * 2.0f - 1.0f;
#else
    return ((float)p + offset) * (1.0f / (float)extent) * 2.0f - 1.0f;
#endif
}

static inline Ray CreateCameraRay(const FrameCBuffer& cb, float u, float v) { // .compute:145-157
    float3 origin = f3(cb.CameraToWorld[12], cb.CameraToWorld[13], cb.CameraToWorld[14]);
    const float* ip = cb.CameraInverseProjection;
    // mul(_CameraInverseProjection, float4(uv, 0, 1)).xyz
    float3 d = f3(((ip[0] * u + ip[4] * v) + ip[8] * 0.0f) + ip[12],
                  ((ip[1] * u + ip[5] * v) + ip[9] * 0.0f) + ip[13],
                  ((ip[2] * u + ip[6] * v) + ip[10] * 0.0f) + ip[14]);
    d = mulDir(cb.CameraToWorld, d);
    d = normalize(d);
    return CreateRay(origin, d);
}

struct RayHit {                                               // .compute:160-168
    float3 position; float distance; float3 normal, albedo, specular, emission;
    int32_t object, triangle;  // added: identity of the accepted hit
    // added for the dielectric extension: the unused fourth components of albedo / specular
    // (RayTraceMaterial.cs:12-17 declares float4s, the shader reads .xyz only) and whether the
    // triangle was hit from behind (only rays inside a dielectric test back faces)
    float albedoW, specularW;
    bool backface;
};
static inline RayHit CreateRayHit() {                         // .compute:169-179
    RayHit h;
    h.position = f3(0, 0, 0); h.distance = INF; h.normal = f3(0, 0, 0);
    h.albedo = f3(0, 0, 0); h.specular = f3(0, 0, 0); h.emission = f3(0, 0, 0);
    h.object = -1; h.triangle = -1;
    h.albedoW = 0.0f; h.specularW = 0.0f; h.backface = false;
    return h;
}

// Möller–Trumbore with back-face culling.  .compute:198-231
// twoSided (dielectric extension, a ray travelling inside glass): the determinant may have either
// sign, only |det| < EPSILON is rejected; `back` reports det < 0.  For det >= EPSILON the
// arithmetic is the reference's, operation for operation.
static inline bool Trisection(const Ray& ray, float3 v0, float3 v1, float3 v2, float& T, float& U, float& V,
                              bool twoSided = false, bool* back = nullptr) {
    float3 v0v1 = v1 - v0;
    float3 v0v2 = v2 - v0;
    float3 pvec = cross(ray.direction, v0v2);
    float det = dot(v0v1, pvec);
    if (det < EPSILON) {
        if (!twoSided || det > -EPSILON) return false;
    }
    if (back) *back = det < 0.0f;
    float inv_det = 1.0f / det;
    float3 tvec = ray.origin - v0;
    float u = dot(tvec, pvec) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    float3 qvec = cross(tvec, v0v1);
    float v = dot(ray.direction, qvec) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = dot(v0v2, qvec) * inv_det;
    if (t > 0) { T = t; U = u; V = v; return true; }
    return false;
}

// Four-box slab test.  .compute:233-263 (no lx<=hx emptiness test, unlike the
// commented C# original FourTransposedAabbs.cs:167).
static inline void Raycast(const Node& n, const Ray& ray, float maxFraction, bool hit[4]) {
    for (int i = 0; i < 4; i++) {
        float lx = n.Lx[i] - ray.origin.x, hx = n.Hx[i] - ray.origin.x;
        float nearXt = lx * ray.invDir.x, farXt = hx * ray.invDir.x;
        float ly = n.Ly[i] - ray.origin.y, hy = n.Hy[i] - ray.origin.y;
        float nearYt = ly * ray.invDir.y, farYt = hy * ray.invDir.y;
        float lz = n.Lz[i] - ray.origin.z, hz = n.Hz[i] - ray.origin.z;
        float nearZt = lz * ray.invDir.z, farZt = hz * ray.invDir.z;
        float nearX = ofmin(nearXt, farXt), farX = ofmax(nearXt, farXt);
        float nearY = ofmin(nearYt, farYt), farY = ofmax(nearYt, farYt);
        float nearZ = ofmin(nearZt, farZt), farZ = ofmax(nearZt, farZt);
        float nearMax = ofmax(ofmax(ofmax(nearX, nearY), nearZ), 0.0f);
        float farMin = ofmin(ofmin(ofmin(farX, farY), farZ), maxFraction);
        hit[i] = nearMax <= farMin;
    }
}

static const int MAX_BLAS_STACK = 64 * 4;   // MAX_BLAS_STACK_INT4 = 64  (.compute:93)
static const int MAX_TLAS_STACK = 8 * 4;    // MAX_TLAS_STACK_INT4 = 8   (.compute:94)
static const int ORACLE_STACK = 4096;       // the oracle keeps going past the shader's limit and reports it

// Diagnostic, not part of the reference: the largest (entry distance of the accepted triangle's own
// leaf-slot box - its hit distance) / hit distance seen by any accepted hit of the last render.  In
// exact arithmetic it is <= 0 (a triangle lies inside its box); what is left is rounding.  A kernel
// that culls boxes by distance with a relative slack must keep the slack above this.
static std::mutex g_gapLock;
static float g_maxEntryGap = 0.0f;

static inline float slotNear(const Node& n, int i, const Ray& ray) {
    float nearXt = (n.Lx[i] - ray.origin.x) * ray.invDir.x, farXt = (n.Hx[i] - ray.origin.x) * ray.invDir.x;
    float nearYt = (n.Ly[i] - ray.origin.y) * ray.invDir.y, farYt = (n.Hy[i] - ray.origin.y) * ray.invDir.y;
    float nearZt = (n.Lz[i] - ray.origin.z) * ray.invDir.z, farZt = (n.Hz[i] - ray.origin.z) * ray.invDir.z;
    return ofmax(ofmax(ofmax(ofmin(nearXt, farXt), ofmin(nearYt, farYt)), ofmin(nearZt, farZt)), 0.0f);
}

struct Tracer {
    const Scene& sc;
    Counters cnt;
    float maxEntryGap = 0.0f;
    explicit Tracer(const Scene& s) : sc(s) { std::memset(&cnt, 0, sizeof(cnt)); }
    ~Tracer() {
        if (maxEntryGap > 0.0f) { std::lock_guard<std::mutex> hold(g_gapLock); if (maxEntryGap > g_maxEntryGap) g_maxEntryGap = maxEntryGap; }
    }

    // .compute:277-346
    void IntersectBLAS(const Ray& ray, int index, RayHit& bestHit, bool twoSided = false) {
        const ObjectProperty& property = sc.objects[index];
        int blasIndex = property.BLASIndex;
        int blasOffset = sc.blasOffsets[blasIndex];
        int triangleOffset = sc.triOffsets[blasIndex];
        int vertexOffset = sc.vtxOffsets[blasIndex];
        const float* worldToLocal = property.WorldToLocal;
        const float* localToWorld = property.LocalToWorld;
        Ray localRay = CreateRay(mulPoint(worldToLocal, ray.origin), mulDir(worldToLocal, ray.direction));
        cnt.instances++;

        static thread_local int stack[ORACLE_STACK];
        int top = 0;
        stack[top++] = 1;
        do {
            top--;
            int topIndex = stack[top];
            const Node& node = sc.blas[topIndex + blasOffset];
            cnt.nodes++;
            bool hitMask[4];
            Raycast(node, localRay, bestHit.distance, hitMask);
            if (node.Flags == 0) {                       // internal
                for (int i = 0; i < 4; i++) {
                    int childNodeIndex = node.Data[i];
                    if (childNodeIndex != 0 && hitMask[i]) {
                        if (top < ORACLE_STACK) stack[top] = childNodeIndex;
                        top++;
                        if ((uint64_t)top > cnt.max_blas_stack) cnt.max_blas_stack = top;
                        if (top > MAX_BLAS_STACK) cnt.overflow++;
                        if (top >= ORACLE_STACK) top = ORACLE_STACK - 1;
                    }
                }
            } else {                                     // leaf
                for (int i = 0; i < 4; i++) {
                    int triangleIndex = node.Data[i];
                    if (triangleIndex != -1 && hitMask[i]) {
                        Int3 aTriangle = sc.tris[triangleIndex + triangleOffset];
                        const Vertex& v0 = sc.verts[aTriangle.x + vertexOffset];
                        const Vertex& v1 = sc.verts[aTriangle.y + vertexOffset];
                        const Vertex& v2 = sc.verts[aTriangle.z + vertexOffset];
                        cnt.tris++;
                        float t, u, v;
                        bool back = false;
                        if (Trisection(localRay, f3(v0.position[0], v0.position[1], v0.position[2]),
                                       f3(v1.position[0], v1.position[1], v1.position[2]),
                                       f3(v2.position[0], v2.position[1], v2.position[2]), t, u, v, twoSided, &back)) {
                            if (t > 0 && t < bestHit.distance) {
                                cnt.hits++;
                                { const float gap = (slotNear(node, i, localRay) - t) / t; if (gap > maxEntryGap) maxEntryGap = gap; }
                                bestHit.distance = t;
                                bestHit.position = mulPoint(localToWorld, localRay.origin + t * localRay.direction);
                                float w = 1 - u - v;
                                float3 n0 = f3(v0.normal[0], v0.normal[1], v0.normal[2]);
                                float3 n1 = f3(v1.normal[0], v1.normal[1], v1.normal[2]);
                                float3 n2 = f3(v2.normal[0], v2.normal[1], v2.normal[2]);
                                float3 nl = ((u * n1) + (v * n2)) + (w * n0);
                                bestHit.normal = normalize(mulDir(localToWorld, nl));
                                bestHit.albedo = f3(property.albedo[0], property.albedo[1], property.albedo[2]);
                                bestHit.specular = f3(property.specular[0], property.specular[1], property.specular[2]);
                                bestHit.emission = f3(property.emission[0], property.emission[1], property.emission[2]);
                                bestHit.object = index;
                                bestHit.triangle = triangleIndex;
                                bestHit.albedoW = property.albedo[3]; bestHit.specularW = property.specular[3];
                                bestHit.backface = back;
                            }
                        }
                    }
                }
            }
        } while (top > 0);
    }

    // .compute:348-384
    RayHit IntersectTLAS(const Ray& ray, bool twoSided = false) {
        RayHit bestHit = CreateRayHit();
        cnt.rays++;
        int stack[ORACLE_STACK / 8];
        const int cap = ORACLE_STACK / 8;
        int top = 0;
        stack[top++] = 1;
        do {
            top--;
            int topIndex = stack[top];
            const Node& node = sc.tlas[topIndex];
            cnt.nodes++;
            bool hitMask[4];
            Raycast(node, ray, bestHit.distance, hitMask);
            for (int i = 0; i < 4; i++) {
                if (!hitMask[i]) continue;
                if (node.Flags == 0) {
                    if (node.Data[i] != 0) {
                        if (top < cap) stack[top] = node.Data[i];
                        top++;
                        if ((uint64_t)top > cnt.max_tlas_stack) cnt.max_tlas_stack = top;
                        if (top > MAX_TLAS_STACK) cnt.overflow++;
                        if (top >= cap) top = cap - 1;
                    }
                } else {
                    if (node.Data[i] != -1) IntersectBLAS(ray, node.Data[i], bestHit, twoSided);
                }
            }
        } while (top > 0);
        return bestHit;
    }
};

// _SkyboxTexture.SampleLevel(sampler, uv, 0): bilinear, repeat.  .compute:409
static inline float3 sampleSky(const Scene& sc, float u, float v) {
    if (!sc.sky || sc.skyW <= 0 || sc.skyH <= 0) return f3(0, 0, 0);
    float x = u * (float)sc.skyW - 0.5f;
    float y = v * (float)sc.skyH - 0.5f;
    float x0f = std::floor(x), y0f = std::floor(y);
    float fx = x - x0f, fy = y - y0f;
    // wrap in float first so that huge |u| cannot overflow the int conversion
    float W = (float)sc.skyW, H = (float)sc.skyH;
    float xw = x0f - std::floor(x0f * (1.0f / W)) * W;
    float yw = y0f - std::floor(y0f * (1.0f / H)) * H;
    int x0 = (int)xw, y0 = (int)yw;
    if (x0 < 0) x0 = 0; if (x0 >= sc.skyW) x0 = sc.skyW - 1;
    if (y0 < 0) y0 = 0; if (y0 >= sc.skyH) y0 = sc.skyH - 1;
    int x1 = x0 + 1 == sc.skyW ? 0 : x0 + 1;
    int y1 = y0 + 1 == sc.skyH ? 0 : y0 + 1;
    const float* p00 = sc.sky + 4 * ((size_t)y0 * sc.skyW + x0);
    const float* p10 = sc.sky + 4 * ((size_t)y0 * sc.skyW + x1);
    const float* p01 = sc.sky + 4 * ((size_t)y1 * sc.skyW + x0);
    const float* p11 = sc.sky + 4 * ((size_t)y1 * sc.skyW + x1);
    float gx = 1.0f - fx, gy = 1.0f - fy;
    float3 r;
    r.x = (p00[0] * gx + p10[0] * fx) * gy + (p01[0] * gx + p11[0] * fx) * fy;
    r.y = (p00[1] * gx + p10[1] * fx) * gy + (p01[1] * gx + p11[1] * fx) * fy;
    r.z = (p00[2] * gx + p10[2] * fx) * gy + (p01[2] * gx + p11[2] * fx) * fy;
    return r;
}

// Miss branch of Shade.  .compute:402-410
static inline float3 shadeMiss(const Scene& sc, float3 dir) {
#ifndef RT_ORACLE_RECIPROCAL_DIVISIONS
    float theta = acosDecl(dir.y) / -PI;
    float phi = atan2Decl(dir.x, -dir.z) / -PI * 0.5f;
#else
    float theta = acosDecl(dir.y) * NEG_INV_PI;
    float phi = atan2Decl(dir.x, -dir.z) * NEG_INV_PI * 0.5f;
#endif
    return sampleSky(sc, phi, theta);
}

// Trace, REF mode.  .compute:413-429 with Shade (:386-411) taking `ray` by value, so
// none of its energy / direction updates survive: a hit returns hit.emission, a miss
// the sky colour along the reflected direction.
static inline void traceRef(Tracer& tr, const Ray& cameraRay, float out[4], RayHit& cameraHit) {
    cameraHit = tr.IntersectTLAS(cameraRay);
    if (cameraHit.distance < INF) {
        Ray ray = CreateRay(cameraHit.position, reflect(cameraRay.direction, cameraHit.normal));
        RayHit hit = tr.IntersectTLAS(ray);
        float3 shade = hit.distance < INF ? hit.emission : shadeMiss(tr.sc, ray.direction);
        float3 result = f3(0, 0, 0) + ray.energy * shade;
        out[0] = result.x; out[1] = result.y; out[2] = result.z; out[3] = 1.0f;
    } else {
        out[0] = out[1] = out[2] = out[3] = 0.0f;
    }
}

static inline float energyOf(float3 c) {                     // .compute:45-48
    const float third = 1.0f / 3.0f;
    return (c.x * third + c.y * third) + c.z * third;
}

// .compute:33-43
static inline void GetTangentSpace(float3 normal, float3& tangent, float3& binormal) {
    float3 helper = f3(1, 0, 0);
    if (std::fabs(normal.x) > 0.99f) helper = f3(0, 0, 1);
    tangent = normalize(cross(normal, helper));
    binormal = normalize(cross(normal, tangent));
}

// Uniform (not cosine-weighted) hemisphere sample.  .compute:67-76
static inline float3 SampleHemisphere(float3 normal, float u1, float u2) {
    float cosTheta = u1;
    float sinTheta = std::sqrt(ofmax(0.0f, 1.0f - cosTheta * cosTheta));
    float sphi, cphi;
    sincos2piDecl(u2, sphi, cphi);
    float3 ts = f3(cphi * sinTheta, sphi * sinTheta, cosTheta);
    float3 tangent, binormal;
    GetTangentSpace(normal, tangent, binormal);
    // mul(tangentSpaceDir, float3x3(tangent, binormal, normal))
    return ((ts.x * tangent) + (ts.y * binormal)) + (ts.z * normal);
}

// PT mode: the path loop the shader's unused code sketches — specular / diffuse roulette
// (.compute:391-399 and the tutorial it cites), emission return (:400), sky on miss
// (:405-409), origin offset n*0.001 (:397).
//
// Dielectric extension (`dielectrics`, flag 64; nothing in the reference transmits light: its material
// has no such term, RayTraceMaterial.cs:12-17, and its triangle test culls back faces, :205): an
// object whose specular.w is 2 is glass with refractive index albedo.w and transmission tint
// albedo.xyz.  A ray that refracted into glass is `inside` until it refracts out again, and only
// such rays test triangles from both sides; a triangle hit from behind is shaded with its normal
// flipped.  At a glass surface: Schlick's Fresnel term F (1 on total internal reflection), the
// segment's roulette draw picks mirror reflection (probability F, weight 1, origin offset along the
// normal) or refraction (weight tint, direction by Snell's law, renormalised, origin offset against
// the normal).
static const float DIELECTRIC_TAG = 2.0f;

static inline float3 tracePath(Tracer& tr, Ray ray, uint32_t key, int maxBounces, bool dielectrics = false) {
    float3 L = f3(0, 0, 0);
    float3 energy = f3(1, 1, 1);
    bool inside = false;
    for (int b = 0;; b++) {
        RayHit hit = tr.IntersectTLAS(ray, inside);
        if (!(hit.distance < INF)) {
            L = L + energy * shadeMiss(tr.sc, ray.direction);
            break;
        }
        if (hit.backface) hit.normal = f3(-hit.normal.x, -hit.normal.y, -hit.normal.z);
        L = L + energy * hit.emission;
        if (b >= maxBounces) break;
        if (dielectrics && hit.specularW == DIELECTRIC_TAG) {
            const float ior = hit.albedoW;
            const float eta = hit.backface ? ior : 1.0f / ior;
            const float cosi = saturate(-dot(ray.direction, hit.normal));
            float r0 = (1.0f - ior) / (1.0f + ior);
            r0 = r0 * r0;
            const float sin2t = (eta * eta) * (1.0f - cosi * cosi);
            float F = 1.0f;
            if (!(sin2t > 1.0f)) {
                const float m = 1.0f - cosi;
                const float m2 = m * m;
                F = r0 + (1.0f - r0) * ((m2 * m2) * m);
            }
            const float roulette = rngDraw(key, 2u + 3u * (uint32_t)b);
            float3 dir, origin;
            if (roulette < F) {
                dir = reflect(ray.direction, hit.normal);
                origin = hit.position + hit.normal * 0.001f;
            } else {
                const float cost = std::sqrt(ofmax(0.0f, 1.0f - sin2t));
                dir = normalize((eta * ray.direction) + ((eta * cosi - cost) * hit.normal));
                origin = hit.position + hit.normal * -0.001f;
                energy = energy * hit.albedo;
                inside = !hit.backface;
            }
            if (!(energy.x > 0.0f || energy.y > 0.0f || energy.z > 0.0f)) break;
            ray = CreateRay(origin, dir);
            continue;
        }
        float3 spec = hit.specular;
        float3 alb = f3(ofmin(1.0f - spec.x, hit.albedo.x), ofmin(1.0f - spec.y, hit.albedo.y),
                        ofmin(1.0f - spec.z, hit.albedo.z));
        float specChance = energyOf(spec);
        float diffChance = energyOf(alb);
        float sum = specChance + diffChance;
        if (!(sum > 0.0f)) break;
#ifndef RT_ORACLE_RECIPROCAL_DIVISIONS
        specChance = specChance / sum;
        diffChance = diffChance / sum;
#else
        float rsum = 1.0f / sum;
        specChance = specChance * rsum;
        diffChance = diffChance * rsum;
#endif
        float roulette = rngDraw(key, 2u + 3u * (uint32_t)b);
        float3 dir;
        if (roulette < specChance) {
            dir = reflect(ray.direction, hit.normal);
            // ray.energy *= (1.0f / specChance) * hit.specular * sdot(hit.normal, ray.direction)
            energy = energy * (((1.0f / specChance) * spec) * saturate(dot(hit.normal, dir)));
        } else {
            float u1 = rngDraw(key, 3u + 3u * (uint32_t)b);
            float u2 = rngDraw(key, 4u + 3u * (uint32_t)b);
            dir = SampleHemisphere(hit.normal, u1, u2);
            // ray.energy *= (1.0f / diffChance) * 2 * hit.albedo * sdot(hit.normal, ray.direction)
            energy = energy * ((((1.0f / diffChance) * 2.0f) * alb) * saturate(dot(hit.normal, dir)));
        }
        if (!(energy.x > 0.0f || energy.y > 0.0f || energy.z > 0.0f)) break;
        ray = CreateRay(hit.position + hit.normal * 0.001f, dir);
    }
    return L;
}

static inline bool pixelSelected(const Params& p, int x, int y, int width, int height) {
    int tx = 0, ty = 0, tw = width, th = height;
    if (p.tile_w > 0 && p.tile_h > 0) { tx = p.tile_x; ty = p.tile_y; tw = p.tile_w; th = p.tile_h; }
    if (x < tx || y < ty || x >= tx + tw || y >= ty + th) return false;
    if (p.shard_count > 1) {
        int tilesX = (width + 31) / 32;
        int tile = (y / 32) * tilesX + (x / 32);
        if (tile % p.shard_count != p.shard_rank) return false;
    }
    return true;
}

static void renderRows(const Scene& sc, const FrameCBuffer& cb, const Params& p, int width, int height,
                       int y0, int y1, float* rgba, int32_t* ids, float* tout, Counters* outCnt) {
    Tracer tr(sc);
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < width; x++) {
            if (!pixelSelected(p, x, y, width, height)) continue;
            size_t pix = (size_t)y * width + x;
            // CSMain: uv = (id.xy + _PixelOffset) / (w,h) * 2 - 1.  .compute:439
            if (p.mode == 0 || ids || tout) {
                float u = pixelToNdc(x, cb.PixelOffset[0], width);
                float v = pixelToNdc(y, cb.PixelOffset[1], height);
                Ray cam = CreateCameraRay(cb, u, v);
                float px[4];
                RayHit h;
                if (p.mode == 0) {
                    traceRef(tr, cam, px, h);
                    if (rgba) std::memcpy(rgba + 4 * pix, px, sizeof(px));
                } else {
                    Tracer side(sc);   // ID pass of PT mode is not part of the counted work
                    h = side.IntersectTLAS(cam);
                }
                if (ids) { ids[2 * pix] = h.object; ids[2 * pix + 1] = h.triangle; }
                if (tout) tout[pix] = h.distance;
            }
            if (p.mode == 1) {
                float3 sum = f3(0, 0, 0);
                for (int s = 0; s < p.spp; s++) {
                    uint32_t key = rngKey(p.seed, (uint32_t)pix, (uint32_t)(p.sample_offset + s));
                    float jx = cb.PixelOffset[0], jy = cb.PixelOffset[1];
                    if (!(p.flags & 2u)) { jx = rngDraw(key, 0u); jy = rngDraw(key, 1u); }
                    float u = pixelToNdc(x, jx, width);
                    float v = pixelToNdc(y, jy, height);
                    Ray cam = CreateCameraRay(cb, u, v);
                    sum = sum + tracePath(tr, cam, key, p.max_bounces, (p.flags & 64u) != 0);
                }
                if (rgba) {
                    if (p.flags & 8u) {
                        rgba[4 * pix] = sum.x; rgba[4 * pix + 1] = sum.y; rgba[4 * pix + 2] = sum.z;
                        rgba[4 * pix + 3] = (float)p.spp;
                    } else {
                        float n = (float)p.spp;
                        rgba[4 * pix] = sum.x / n; rgba[4 * pix + 1] = sum.y / n; rgba[4 * pix + 2] = sum.z / n;
                        rgba[4 * pix + 3] = 1.0f;
                    }
                }
            }
        }
    }
    *outCnt = tr.cnt;
}

}  // namespace

extern "C" {

// Render `width` x `height` pixels with the reference semantics (mode 0) or the path
// tracer (mode 1).  Any of rgba / ids / tout / counters may be NULL.  Pixels outside
// the tile rectangle / shard are left untouched.  Returns 0, or 1 if some traversal
// needed more stack than the reference shader declares (results are still those of
// an unbounded stack).
int oracle_render(const Scene* sc, const FrameCBuffer* cb, const Params* p, int width, int height,
                  float* rgba, int32_t* ids, float* tout, Counters* counters) {
    int nt = p->nthreads > 0 ? p->nthreads : 1;
    if (nt > height) nt = height;
    { std::lock_guard<std::mutex> hold(g_gapLock); g_maxEntryGap = 0.0f; }
    std::vector<Counters> cnts(nt);
    if (nt == 1) {
        renderRows(*sc, *cb, *p, width, height, 0, height, rgba, ids, tout, &cnts[0]);
    } else {
        // interleaved row blocks so that every thread sees a similar mix of sky and geometry
        std::vector<std::thread> th;
        const int block = 4;
        for (int t = 0; t < nt; t++) {
            th.emplace_back([&, t]() {
                Counters acc; std::memset(&acc, 0, sizeof(acc));
                for (int y0 = t * block; y0 < height; y0 += nt * block) {
                    Counters c;
                    int y1 = y0 + block < height ? y0 + block : height;
                    renderRows(*sc, *cb, *p, width, height, y0, y1, rgba, ids, tout, &c);
                    acc.rays += c.rays; acc.nodes += c.nodes; acc.tris += c.tris;
                    acc.instances += c.instances; acc.hits += c.hits; acc.overflow += c.overflow;
                    if (c.max_blas_stack > acc.max_blas_stack) acc.max_blas_stack = c.max_blas_stack;
                    if (c.max_tlas_stack > acc.max_tlas_stack) acc.max_tlas_stack = c.max_tlas_stack;
                }
                cnts[t] = acc;
            });
        }
        for (auto& t : th) t.join();
    }
    Counters tot; std::memset(&tot, 0, sizeof(tot));
    for (auto& c : cnts) {
        tot.rays += c.rays; tot.nodes += c.nodes; tot.tris += c.tris; tot.instances += c.instances;
        tot.hits += c.hits; tot.overflow += c.overflow;
        if (c.max_blas_stack > tot.max_blas_stack) tot.max_blas_stack = c.max_blas_stack;
        if (c.max_tlas_stack > tot.max_tlas_stack) tot.max_tlas_stack = c.max_tlas_stack;
    }
    if (counters) *counters = tot;
    return tot.overflow ? 1 : 0;
}

// See g_maxEntryGap: of the last oracle_render call (0 when no accepted hit had a box entry behind it).
float oracle_last_max_entry_gap(void) {
    std::lock_guard<std::mutex> hold(g_gapLock);
    return g_maxEntryGap;
}

// Closest front-facing hit of the un-jittered primary ray by testing EVERY triangle of
// EVERY object (no BVH): the ground truth the traversal must reproduce.  ties[pix] = number
// of other (object, triangle) pairs whose t equals the winner's bit for bit; where it is
// zero the winner does not depend on visiting order.
int oracle_bruteforce_primary(const Scene* sc, const FrameCBuffer* cb, int width, int height,
                              int32_t* ids, float* tout, int32_t* ties) {
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            size_t pix = (size_t)y * width + x;
            float u = pixelToNdc(x, cb->PixelOffset[0], width);
            float v = pixelToNdc(y, cb->PixelOffset[1], height);
            Ray ray = CreateCameraRay(*cb, u, v);
            float best = INF; int bo = -1, bt = -1, nt = 0;
            for (int o = 0; o < sc->nObjects; o++) {
                const ObjectProperty& pr = sc->objects[o];
                int b = pr.BLASIndex;
                int t0 = sc->triOffsets[b];
                int t1 = b + 1 < sc->nBlas ? sc->triOffsets[b + 1] : sc->nTris;
                int vo = sc->vtxOffsets[b];
                Ray lr = CreateRay(mulPoint(pr.WorldToLocal, ray.origin), mulDir(pr.WorldToLocal, ray.direction));
                for (int t = t0; t < t1; t++) {
                    Int3 tri = sc->tris[t];
                    const float* a = sc->verts[tri.x + vo].position;
                    const float* bb = sc->verts[tri.y + vo].position;
                    const float* c = sc->verts[tri.z + vo].position;
                    float T, U, V;
                    if (Trisection(lr, f3(a[0], a[1], a[2]), f3(bb[0], bb[1], bb[2]), f3(c[0], c[1], c[2]), T, U, V)) {
                        if (T < best) { best = T; bo = o; bt = t - t0; nt = 0; }
                        else if (T == best) nt++;
                    }
                }
            }
            if (ids) { ids[2 * pix] = bo; ids[2 * pix + 1] = bt; }
            if (tout) tout[pix] = best;
            if (ties) ties[pix] = nt;
        }
    return 0;
}

// Restatement of BoundingVolumeHierarchy.CheckIntegrity (BoundingVolumeHierarchyBuilder.cs:
// 844-880) plus "every primitive index in [0,nPrims) appears in exactly one leaf slot".
// Returns 0 if sound, else: 1 valid-data/valid-aabb mismatch, 2 parent does not contain
// child, 3 primitive missing or duplicated, 4 index out of range, 5 cycle / too deep.
int oracle_check_integrity(const Node* nodes, int nNodes, int nPrims) {
    std::vector<int> seen(nPrims > 0 ? nPrims : 0, 0);
    struct Item { int node, parent, child, depth; };
    std::vector<Item> st;
    st.push_back({1, 0, 0, 0});
    if (nNodes < 2) return 4;
    int visited = 0;
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        if (it.node <= 0 || it.node >= nNodes) return 4;
        if (++visited > nNodes || it.depth > 4096) return 5;
        const Node& parent = nodes[it.parent];
        const Node& node = nodes[it.node];
        for (int i = 0; i < 4; i++) {
            int data = node.Data[i];
            bool leaf = node.Flags != 0;
            bool validData = leaf ? data != -1 : data != 0;
            bool validAabb = node.Lx[i] <= node.Hx[i] && node.Ly[i] <= node.Hy[i] && node.Lz[i] <= node.Hz[i];
            if (validData != validAabb) return 1;
            if (validData) {
                if (it.parent != 0) {
                    int c = it.child;
                    bool contains = parent.Lx[c] <= node.Lx[i] && parent.Ly[c] <= node.Ly[i] && parent.Lz[c] <= node.Lz[i] &&
                                    parent.Hx[c] >= node.Hx[i] && parent.Hy[c] >= node.Hy[i] && parent.Hz[c] >= node.Hz[i];
                    if (!contains) return 2;
                }
                if (!leaf) st.push_back({data, it.node, i, it.depth + 1});
                else {
                    if (data < 0 || data >= nPrims) return 4;
                    seen[data]++;
                }
            }
        }
    }
    for (int i = 0; i < nPrims; i++) if (seen[i] != 1) return 3;
    return 0;
}

// The declared transcendental functions, exported so tests can bound their error.
float oracle_acos(float x) { return acosDecl(x); }
float oracle_atan2(float y, float x) { return atan2Decl(y, x); }
void oracle_sincos2pi(float u, float* s, float* c) { sincos2piDecl(u, *s, *c); }
float oracle_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t dim) {
    return rngDraw(rngKey(seed, pixel, sample), dim);
}

int oracle_struct_sizes(int32_t out[5]) {
    out[0] = (int32_t)sizeof(Node); out[1] = (int32_t)sizeof(Vertex); out[2] = (int32_t)sizeof(Int3);
    out[3] = (int32_t)sizeof(ObjectProperty); out[4] = (int32_t)sizeof(FrameCBuffer);
    return 0;
}

}  // extern "C"

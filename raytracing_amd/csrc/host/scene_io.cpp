// Host-side helpers that are not restatements of reference code: error text, 4x4 inverse and TRS
// matrices, Unity's camera matrices, and the flattened-scene file.

#include <cfloat>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <vector>

#include "host_internal.h"

namespace rtamd_host {

static thread_local std::string g_lastError;
void setLastError(const std::string& msg) { g_lastError = msg; }
const std::string& lastError() { return g_lastError; }

}  // namespace rtamd_host

using namespace rtamd_host;

extern "C" {

const char* rtamd_host_last_error(void) { return lastError().c_str(); }

// General inverse by cofactors (column-major storage; the formula is layout-agnostic
// because inverse(transpose(M)) == transpose(inverse(M))).
int rtamd_host_inverse4x4(const float m[16], float inv[16]) {
    float t[16];
    t[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    t[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    t[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    t[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    t[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    t[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    t[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    t[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    t[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    t[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    t[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    t[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    t[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    t[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    t[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    t[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * t[0] + m[1] * t[4] + m[2] * t[8] + m[3] * t[12];
    if (det == 0.0f || det != det) return -1;
    float id = 1.0f / det;
    for (int i = 0; i < 16; i++) inv[i] = t[i] * id;
    return 0;
}

void rtamd_host_trs(const float t[3], const float q[4], const float s[3], float out[16]) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float r00 = 1 - 2 * (y * y + z * z), r01 = 2 * (x * y - z * w), r02 = 2 * (x * z + y * w);
    float r10 = 2 * (x * y + z * w), r11 = 1 - 2 * (x * x + z * z), r12 = 2 * (y * z - x * w);
    float r20 = 2 * (x * z - y * w), r21 = 2 * (y * z + x * w), r22 = 1 - 2 * (x * x + y * y);
    out[0] = r00 * s[0]; out[1] = r10 * s[0]; out[2] = r20 * s[0]; out[3] = 0;
    out[4] = r01 * s[1]; out[5] = r11 * s[1]; out[6] = r21 * s[1]; out[7] = 0;
    out[8] = r02 * s[2]; out[9] = r12 * s[2]; out[10] = r22 * s[2]; out[11] = 0;
    out[12] = t[0]; out[13] = t[1]; out[14] = t[2]; out[15] = 1;
}

// camera.cameraToWorldMatrix = TRS(position, rotation, 1) * scale(1,1,-1);
// camera.projectionMatrix = OpenGL perspective.  RayTracingPass.cs:74-75.
void rtamd_host_camera(const float position[3], const float rot[4], float fovDeg, float aspect, float zn, float zf,
                       rtamd_frame_cbuffer* out) {
    std::memset(out, 0, sizeof(*out));
    const float one[3] = {1, 1, 1};
    float trs[16];
    rtamd_host_trs(position, rot, one, trs);
    std::memcpy(out->CameraToWorld, trs, sizeof(trs));
    for (int r = 0; r < 4; r++) out->CameraToWorld[8 + r] = -trs[8 + r];
    float f = 1.0f / (float)std::tan((double)fovDeg * 3.14159265358979323846 / 360.0);
    float P[16] = {0};
    P[0] = f / aspect; P[5] = f; P[10] = -(zf + zn) / (zf - zn); P[11] = -1.0f; P[14] = -2.0f * zf * zn / (zf - zn);
    rtamd_host_inverse4x4(P, out->CameraInverseProjection);
    out->PixelOffset[0] = 0; out->PixelOffset[1] = 0;   // RayTracingPass.cs:76
    out->Seed = 0; out->ReflectionNum = 1;
}

// ----------------------------------------------------------- scene container ----
// A flattened scene held on the host: the eight upload buffers, filled by rtamd_host_scene_load (or by a
// producer outside this library) and handed out as pointers.
rtamd_host_scene* rtamd_host_scene_create(void) { return new (std::nothrow) rtamd_host_scene(); }

void rtamd_host_scene_destroy(rtamd_host_scene* s) { delete s; }

int rtamd_host_scene_buffers(const rtamd_host_scene* s, rtamd_host_buffers* out) {
    if (!s || !out) return RTAMD_ERR_INVALID_ARG;
    if (s->objects.empty()) return RTAMD_ERR_NO_SCENE;
    out->tlas = s->tlas.data(); out->n_tlas_nodes = (int)s->tlas.size();
    out->blas_offsets = s->blasOffsets.data();
    out->blas = s->blasNodes.data(); out->n_blas_nodes = (int)s->blasNodes.size();
    out->vertices_offsets = s->vtxOffsets.data();
    out->vertices = s->vertices.data(); out->n_vertices = (int)s->vertices.size();
    out->triangles_offsets = s->triOffsets.data();
    out->triangles = s->triangles.data(); out->n_triangles = (int)s->triangles.size();
    out->n_blas = s->nBlas;
    out->objects = s->objects.data(); out->n_objects = (int)s->objects.size();
    out->tlas_branch_count = s->branchCount;
    return RTAMD_OK;
}

// ----------------------------------------------------------- scene file ----
static const char kMagic[8] = {'R', 'T', 'A', 'M', 'D', 'S', 'C', '1'};

int rtamd_host_scene_save(const rtamd_host_buffers* b, const char* path) {
    if (!b || !path) return RTAMD_ERR_INVALID_ARG;
    FILE* f = std::fopen(path, "wb");
    if (!f) { setLastError(std::string("cannot open ") + path); return RTAMD_ERR_INVALID_ARG; }
    int32_t hdr[8] = {b->n_tlas_nodes, b->n_blas_nodes, b->n_vertices, b->n_triangles, b->n_blas, b->n_objects,
                      b->tlas_branch_count, 0};
    bool ok = std::fwrite(kMagic, 1, 8, f) == 8 && std::fwrite(hdr, 4, 8, f) == 8;
    auto put = [&](const void* p, size_t sz, size_t n) { if (ok && n) ok = std::fwrite(p, sz, n, f) == n; };
    put(b->tlas, sizeof(rtamd_node), b->n_tlas_nodes);
    put(b->blas_offsets, 4, b->n_blas);
    put(b->blas, sizeof(rtamd_node), b->n_blas_nodes);
    put(b->vertices_offsets, 4, b->n_blas);
    put(b->vertices, sizeof(rtamd_vertex), b->n_vertices);
    put(b->triangles_offsets, 4, b->n_blas);
    put(b->triangles, sizeof(rtamd_int3), b->n_triangles);
    put(b->objects, sizeof(rtamd_object_property), b->n_objects);
    std::fclose(f);
    return ok ? RTAMD_OK : RTAMD_ERR_INVALID_ARG;
}

int rtamd_host_scene_load(rtamd_host_scene* s, const char* path) {
    if (!s || !path) return RTAMD_ERR_INVALID_ARG;
    FILE* f = std::fopen(path, "rb");
    if (!f) { setLastError(std::string("cannot open ") + path); return RTAMD_ERR_INVALID_ARG; }
    char magic[8]; int32_t hdr[8];
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, kMagic, 8) == 0 && std::fread(hdr, 4, 8, f) == 8;
    for (int i = 0; ok && i < 6; i++) ok = hdr[i] >= 0 && hdr[i] < (1 << 30);
    if (ok) {
        // the counts must account for the rest of the file exactly, before anything is sized by them
        const uint64_t need = (uint64_t)hdr[0] * sizeof(rtamd_node) + (uint64_t)hdr[1] * sizeof(rtamd_node) + (uint64_t)hdr[2] * sizeof(rtamd_vertex) +
                              (uint64_t)hdr[3] * sizeof(rtamd_int3) + (uint64_t)hdr[4] * 12u + (uint64_t)hdr[5] * sizeof(rtamd_object_property);
        ok = std::fseek(f, 0, SEEK_END) == 0;
        const long size = ok ? std::ftell(f) : -1;
        ok = ok && size >= 40 && (uint64_t)(size - 40) == need && std::fseek(f, 40, SEEK_SET) == 0;
    }
    int rc = RTAMD_OK;
    if (ok) {
        try {
            // into buffers of its own first: a file that ends early leaves the scene as it was
            std::vector<rtamd_node> tlas(hdr[0]), blasNodes(hdr[1]);
            std::vector<rtamd_vertex> vertices(hdr[2]);
            std::vector<rtamd_int3> triangles(hdr[3]);
            std::vector<int32_t> blasOffsets(hdr[4]), vtxOffsets(hdr[4]), triOffsets(hdr[4]);
            std::vector<rtamd_object_property> objects(hdr[5]);
            auto get = [&](void* p, size_t sz, size_t n) { if (ok && n) ok = std::fread(p, sz, n, f) == n; };
            get(tlas.data(), sizeof(rtamd_node), hdr[0]);
            get(blasOffsets.data(), 4, hdr[4]);
            get(blasNodes.data(), sizeof(rtamd_node), hdr[1]);
            get(vtxOffsets.data(), 4, hdr[4]);
            get(vertices.data(), sizeof(rtamd_vertex), hdr[2]);
            get(triOffsets.data(), 4, hdr[4]);
            get(triangles.data(), sizeof(rtamd_int3), hdr[3]);
            get(objects.data(), sizeof(rtamd_object_property), hdr[5]);
            if (ok) {
                s->tlas.swap(tlas); s->blasNodes.swap(blasNodes); s->vertices.swap(vertices); s->triangles.swap(triangles);
                s->blasOffsets.swap(blasOffsets); s->vtxOffsets.swap(vtxOffsets); s->triOffsets.swap(triOffsets);
                s->objects.swap(objects); s->nBlas = hdr[4]; s->branchCount = hdr[6];
            }
        } catch (const std::bad_alloc&) {
            setLastError("out of memory reading a scene file");
            rc = RTAMD_ERR_OUT_OF_MEMORY;
        }
    }
    std::fclose(f);
    if (rc != RTAMD_OK) return rc;
    if (!ok) { setLastError("bad scene file"); return RTAMD_ERR_INVALID_ARG; }
    return RTAMD_OK;
}

}  // extern "C"

// Mutation fuzzer of the readers of untrusted files — sky images (OpenEXR incl. PIZ, Radiance RGBE), COLLADA meshes,
// the flattened-scene file: built with the host sources and -fsanitize=address,undefined, it feeds
// rtamd_host_image_load / rtamd_host_mesh_load_dae / rtamd_host_scene_load randomly damaged copies of the seed
// files and expects a result or an error code — never an out-of-bounds access, an overflow, an uncaught exception.
//   fuzz_readers <iterations> <rng seed> <work file> <seed file (.exr | .hdr | .dae | .rtscene)>...
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtamd_host.h"

static uint64_t state = 88172645463325252ull;
static uint32_t rnd() {
    state ^= state << 13; state ^= state >> 7; state ^= state << 17;
    return (uint32_t)(state >> 16);
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: fuzz_readers <iterations> <rng seed> <work file> <seed file>...\n"); return 2; }
    const long iterations = atol(argv[1]);
    state ^= (uint64_t)atoll(argv[2]) * 0x9E3779B97F4A7C15ull;
    const std::string work = argv[3];
    std::vector<std::vector<unsigned char>> seeds;
    std::vector<std::string> ext;
    for (int i = 4; i < argc; i++) {
        FILE* f = fopen(argv[i], "rb");
        if (!f) { fprintf(stderr, "cannot read %s\n", argv[i]); return 2; }
        std::vector<unsigned char> b;
        unsigned char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + n);
        fclose(f);
        seeds.push_back(b);
        const char* dot = strrchr(argv[i], '.');
        ext.push_back(dot ? dot : ".dae");
    }
    long ok = 0, refused = 0;
    for (long it = 0; it < iterations; it++) {
        const size_t k = rnd() % seeds.size();
        const bool text = ext[k] == ".dae";
        std::vector<unsigned char> b = seeds[k];
        const int edits = 1 + (int)(rnd() % 4);
        for (int e = 0; e < edits && !b.empty(); e++) {
            const uint32_t kind = rnd() % 8;
            // (binary file: most edits in the header, where the counts and offsets are)
            const size_t at = (ext[k] == ".rtscene" && rnd() % 2) ? rnd() % (b.size() < 256 ? b.size() : 256) : rnd() % b.size();
            const unsigned char any = text ? (unsigned char)("0123456789 -.e<>/\"#a"[rnd() % 20]) : (unsigned char)rnd();
            if (kind < 3) b[at] = any;
            else if (kind == 3) b[at] ^= (unsigned char)(1u << (rnd() % 8));
            else if (kind == 4) { for (int j = 0; j < 4 && at + j < b.size(); j++) b[at + j] = text ? any : (unsigned char)(rnd() % 3 == 0 ? 0xff : rnd()); }
            else if (kind == 5) b.resize(at);
            else if (kind == 6) b.insert(b.begin() + at, (size_t)(1 + rnd() % 8), any);
            else { const size_t n = 1 + rnd() % 8; if (at + n <= b.size()) b.erase(b.begin() + at, b.begin() + at + n); }
        }
        const std::string path = work + ext[k];
        FILE* f = fopen(path.c_str(), "wb");
        if (!f) { fprintf(stderr, "cannot write %s\n", path.c_str()); return 2; }
        if (!b.empty()) fwrite(b.data(), 1, b.size(), f);
        fclose(f);
        if (ext[k] == ".exr" || ext[k] == ".hdr") {
            float* rgba = nullptr;
            int32_t w = 0, h = 0;
            if (rtamd_host_image_load(path.c_str(), (int32_t)(rnd() & 1u), &rgba, &w, &h) == 0) {
                volatile float sink = 0.f;       // every value the reader says it returned
                for (size_t i = 0; i < (size_t)w * (size_t)h * 4; i++) sink = sink + rgba[i];
                rtamd_host_image_free(rgba);
                ok++;
            } else refused++;
        } else if (text) {
            rtamd_host_mesh m;
            memset(&m, 0, sizeof m);
            if (rtamd_host_mesh_load_dae(path.c_str(), &m) == 0) {
                volatile double sink = 0;
                for (int i = 0; i < 3 * m.n_vertices; i++) sink = sink + m.positions[i] + m.normals[i];
                for (int i = 0; i < 3 * m.n_triangles; i++) sink = sink + m.positions[3 * m.indices[i]];     // every index addresses a vertex
                rtamd_host_mesh_free(&m);
                ok++;
            } else refused++;
        } else {
            rtamd_host_scene* s = rtamd_host_scene_create();
            if (rtamd_host_scene_load(s, path.c_str()) == 0) {
                rtamd_host_buffers hb;
                if (rtamd_host_scene_buffers(s, &hb) == 0) {
                    volatile double sink = 0;
                    for (int i = 0; i < hb.n_vertices; i++) sink = sink + hb.vertices[i].position[0];
                    for (int i = 0; i < hb.n_triangles; i++) sink = sink + hb.triangles[i].x;
                    for (int i = 0; i < hb.n_blas_nodes; i++) sink = sink + ((const float*)&hb.blas[i])[0];
                    for (int i = 0; i < hb.n_tlas_nodes; i++) sink = sink + ((const float*)&hb.tlas[i])[0];
                    for (int i = 0; i < hb.n_objects; i++) sink = sink + ((const float*)&hb.objects[i])[0];
                    for (int i = 0; i < hb.n_blas; i++) sink = sink + hb.blas_offsets[i] + hb.vertices_offsets[i] + hb.triangles_offsets[i];
                }
                ok++;
            } else refused++;
            rtamd_host_scene_destroy(s);
        }
    }
    printf("%ld files: %ld read, %ld refused\n", iterations, ok, refused);
    return 0;
}

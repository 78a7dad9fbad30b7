// Mesh sources for scenes built without Unity: generators for the built-in
// primitives the reference's prefabs use (Sphere, Quad, Cube), a COLLADA reader for
// Assets/Models/*.dae (<triangles> with VERTEX / NORMAL index pairs, e.g.
// Assets/Models/bunny.dae:64-67), and midpoint subdivision to reach large triangle counts.
//
// Winding convention everywhere: for a triangle (a,b,c) the vector cross(b-a, c-a)
// points out of the surface, which is the side the shader's culled Möller–Trumbore
// test (Assets/Shaders/RayTracingShader.compute:204-208) accepts rays from.

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <stdexcept>
#include <vector>
#include <string>

#include "host_internal.h"

namespace rtamd_host {
namespace {

struct MeshBuf {
    std::vector<float> pos, nrm;
    std::vector<int32_t> idx;
};

int emit(const MeshBuf& m, rtamd_host_mesh* out) {
    size_t nv = m.pos.size() / 3, nt = m.idx.size() / 3;
    out->positions = (float*)std::malloc(sizeof(float) * 3 * (nv ? nv : 1));
    out->normals = (float*)std::malloc(sizeof(float) * 3 * (nv ? nv : 1));
    out->indices = (int32_t*)std::malloc(sizeof(int32_t) * 3 * (nt ? nt : 1));
    if (!out->positions || !out->normals || !out->indices) {
        rtamd_host_mesh_free(out);
        return RTAMD_ERR_OUT_OF_MEMORY;
    }
    if (nv) {
        std::memcpy(out->positions, m.pos.data(), sizeof(float) * 3 * nv);
        std::memcpy(out->normals, m.nrm.data(), sizeof(float) * 3 * nv);
    }
    if (nt) std::memcpy(out->indices, m.idx.data(), sizeof(int32_t) * 3 * nt);
    out->n_vertices = (int32_t)nv;
    out->n_triangles = (int32_t)nt;
    return RTAMD_OK;
}

void pushTriOutward(MeshBuf& m, int a, int b, int c, const float outward[3]) {
    const float* A = &m.pos[3 * a]; const float* B = &m.pos[3 * b]; const float* C = &m.pos[3 * c];
    float e1[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]}, e2[3] = {C[0] - A[0], C[1] - A[1], C[2] - A[2]};
    float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    float l2 = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    if (!(l2 > 0.0f)) return;   // degenerate (pole)
    float d = n[0] * outward[0] + n[1] * outward[1] + n[2] * outward[2];
    if (d < 0) std::swap(b, c);
    m.idx.push_back(a); m.idx.push_back(b); m.idx.push_back(c);
}

std::string readFile(const char* path) {
    std::string s;
    FILE* f = std::fopen(path, "rb");
    if (!f) return s;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n > 0) { s.resize((size_t)n); if (std::fread(&s[0], 1, (size_t)n, f) != (size_t)n) s.clear(); }
    std::fclose(f);
    return s;
}

std::string attr(const std::string& tag, const char* name) {
    std::string key = std::string(name) + "=\"";
    size_t p = tag.find(key);
    if (p == std::string::npos) return "";
    p += key.size();
    size_t e = tag.find('"', p);
    return e == std::string::npos ? "" : tag.substr(p, e - p);
}

}  // namespace
}  // namespace rtamd_host

using namespace rtamd_host;

extern "C" {

void rtamd_host_mesh_free(rtamd_host_mesh* m) {
    if (!m) return;
    std::free(m->positions); std::free(m->normals); std::free(m->indices);
    std::memset(m, 0, sizeof(*m));
}

int rtamd_host_mesh_uv_sphere(int32_t segments, int32_t rings, float radius, rtamd_host_mesh* out) {
    if (!out || segments < 3 || rings < 2 || !(radius > 0)) return RTAMD_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof(*out));
    MeshBuf m;
    const double pi = 3.14159265358979323846;
    for (int i = 0; i <= rings; i++) {
        double th = pi * i / rings;
        for (int j = 0; j <= segments; j++) {
            double ph = 2.0 * pi * (j % segments) / segments;
            float n[3] = {(float)(std::sin(th) * std::cos(ph)), (float)std::cos(th), (float)(std::sin(th) * std::sin(ph))};
            if (i == 0) { n[0] = 0; n[1] = 1; n[2] = 0; }
            if (i == rings) { n[0] = 0; n[1] = -1; n[2] = 0; }
            for (int k = 0; k < 3; k++) { m.pos.push_back(radius * n[k]); m.nrm.push_back(n[k]); }
        }
    }
    int stride = segments + 1;
    for (int i = 0; i < rings; i++)
        for (int j = 0; j < segments; j++) {
            int a = i * stride + j, b = (i + 1) * stride + j, c = (i + 1) * stride + j + 1, d = i * stride + j + 1;
            float o[3];
            for (int k = 0; k < 3; k++) o[k] = m.pos[3 * a + k] + m.pos[3 * b + k] + m.pos[3 * c + k] + m.pos[3 * d + k];
            pushTriOutward(m, a, b, c, o);
            pushTriOutward(m, a, c, d, o);
        }
    return emit(m, out);
}

int rtamd_host_mesh_quad(rtamd_host_mesh* out) {
    if (!out) return RTAMD_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof(*out));
    MeshBuf m;
    const float p[4][3] = {{-0.5f, -0.5f, 0}, {0.5f, -0.5f, 0}, {-0.5f, 0.5f, 0}, {0.5f, 0.5f, 0}};
    for (int i = 0; i < 4; i++) {
        for (int k = 0; k < 3; k++) m.pos.push_back(p[i][k]);
        m.nrm.push_back(0); m.nrm.push_back(0); m.nrm.push_back(-1);
    }
    const int t[6] = {0, 3, 1, 3, 0, 2};
    m.idx.assign(t, t + 6);
    return emit(m, out);
}

int rtamd_host_mesh_cube(rtamd_host_mesh* out) {
    if (!out) return RTAMD_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof(*out));
    MeshBuf m;
    for (int axis = 0; axis < 3; axis++)
        for (int sgn = -1; sgn <= 1; sgn += 2) {
            int u = (axis + 1) % 3, v = (axis + 2) % 3;
            int base = (int)(m.pos.size() / 3);
            float n[3] = {0, 0, 0};
            n[axis] = (float)sgn;
            for (int c = 0; c < 4; c++) {
                float p[3];
                p[axis] = 0.5f * sgn;
                p[u] = (c & 1) ? 0.5f : -0.5f;
                p[v] = (c & 2) ? 0.5f : -0.5f;
                for (int k = 0; k < 3; k++) { m.pos.push_back(p[k]); m.nrm.push_back(n[k]); }
            }
            pushTriOutward(m, base, base + 1, base + 3, n);
            pushTriOutward(m, base, base + 3, base + 2, n);
        }
    return emit(m, out);
}

}  // extern "C"

namespace {

// "#id" -> "id" (a source reference); anything else as it is
std::string afterHash(const std::string& ref) { return (!ref.empty() && ref[0] == '#') ? ref.substr(1) : ref; }

// numbers between two positions of the text (strtof / strtol stop at the first character that is not part of one)
template <typename T, typename Parse>
void parseNumbers(const std::string& x, size_t b, size_t e, std::vector<T>& out, Parse parse) {
    if (b == std::string::npos || e == std::string::npos || b >= e || e > x.size()) return;
    const std::string piece = x.substr(b, e - b);      // (a copy with its own terminator: parsing cannot run past `e`)
    const char* c = piece.c_str();
    for (;;) {
        char* next = nullptr;
        const T v = parse(c, &next);
        if (next == c) break;
        out.push_back(v);
        c = next;
    }
}

int loadDae(const char* path, rtamd_host_mesh* out) {
    const size_t npos = std::string::npos;
    std::string x = readFile(path);
    if (x.empty()) { setLastError(std::string("cannot read ") + path); return RTAMD_ERR_INVALID_ARG; }
    bool zUp = x.find("<up_axis>Z_UP</up_axis>") != npos;
    size_t mesh = x.find("<mesh>");
    size_t meshEnd = mesh == npos ? npos : x.find("</mesh>", mesh);
    if (mesh == npos || meshEnd == npos) { setLastError("no <mesh>"); return RTAMD_ERR_INVALID_ARG; }

    // sources: id -> floats
    std::map<std::string, std::vector<float>> sources;
    size_t p = mesh;
    while ((p = x.find("<source ", p)) != npos && p < meshEnd) {
        size_t te = x.find('>', p);
        size_t se = te == npos ? npos : x.find("</source>", te);
        if (se == npos || se > meshEnd) { setLastError("unterminated <source>"); return RTAMD_ERR_INVALID_ARG; }
        std::string id = attr(x.substr(p, te - p), "id");
        size_t fa = x.find("<float_array", te);
        if (fa != npos && fa < se) {
            size_t open = x.find('>', fa), e = x.find("</float_array>", fa);
            if (open == npos || e == npos || open > e || e > se) { setLastError("unterminated <float_array>"); return RTAMD_ERR_INVALID_ARG; }
            parseNumbers<float>(x, open + 1, e, sources[id], [](const char* c, char** next) { return std::strtof(c, next); });
        }
        p = se;
    }
    // <vertices id=..><input semantic="POSITION" source="#..."/>
    std::map<std::string, std::string> vertexPos;
    p = mesh;
    while ((p = x.find("<vertices ", p)) != npos && p < meshEnd) {
        size_t te = x.find('>', p);
        size_t in = te == npos ? npos : x.find("<input", te);
        size_t ie = in == npos ? npos : x.find('>', in);
        if (ie == npos || ie > meshEnd) { setLastError("unterminated <vertices>"); return RTAMD_ERR_INVALID_ARG; }
        std::string id = attr(x.substr(p, te - p), "id");
        std::string tag = x.substr(in, ie - in);
        if (attr(tag, "semantic") == "POSITION") vertexPos[id] = afterHash(attr(tag, "source"));
        p = ie;
    }
    size_t tr = x.find("<triangles", mesh);
    if (tr == npos || tr > meshEnd) { setLastError("no <triangles>"); return RTAMD_ERR_INVALID_ARG; }
    size_t trEnd = x.find("</triangles>", tr);
    if (trEnd == npos || trEnd > meshEnd) { setLastError("unterminated <triangles>"); return RTAMD_ERR_INVALID_ARG; }
    int stride = 0, offV = -1, offN = -1;
    std::string posSrc, nrmSrc;
    p = tr;
    while ((p = x.find("<input", p)) != npos && p < trEnd) {
        size_t ie = x.find('>', p);
        if (ie == npos || ie > trEnd) { setLastError("unterminated <input>"); return RTAMD_ERR_INVALID_ARG; }
        std::string tag = x.substr(p, ie - p);
        int off = std::atoi(attr(tag, "offset").c_str());
        if (off < 0 || off > 63) { setLastError("<input> offset out of range"); return RTAMD_ERR_INVALID_ARG; }
        std::string sem = attr(tag, "semantic"), src = afterHash(attr(tag, "source"));
        if (sem == "VERTEX") { offV = off; posSrc = vertexPos.count(src) ? vertexPos[src] : src; }
        if (sem == "NORMAL") { offN = off; nrmSrc = src; }
        if (off + 1 > stride) stride = off + 1;
        p = ie;
    }
    if (offV < 0 || !sources.count(posSrc)) { setLastError("no VERTEX input"); return RTAMD_ERR_INVALID_ARG; }
    const std::vector<float>& P = sources[posSrc];
    const std::vector<float>* N = (offN >= 0 && sources.count(nrmSrc)) ? &sources[nrmSrc] : nullptr;
    size_t pb = x.find("<p>", tr);
    size_t pe = pb == npos ? npos : x.find("</p>", pb);
    if (pb == npos || pe == npos || pe > trEnd) { setLastError("no <p>"); return RTAMD_ERR_INVALID_ARG; }
    std::vector<long> raw;
    parseNumbers<long>(x, pb + 3, pe, raw, [](const char* c, char** next) { return std::strtol(c, next, 10); });
    MeshBuf m;
    std::map<std::pair<int, int>, int> weld;   // (position index, normal index) -> vertex, first-seen order
    size_t corners = raw.size() / (size_t)stride;
    for (size_t c = 0; c < corners; c++) {
        const long vl = raw[c * stride + offV];
        const long nl = N ? raw[c * stride + offN] : -1;
        if (vl < 0 || (unsigned long)vl >= P.size() / 3 || (N && (nl < 0 || (unsigned long)nl >= N->size() / 3))) {
            setLastError("index out of range in <p>");
            return RTAMD_ERR_BAD_INDEX;
        }
        const int vi = (int)vl, ni = (int)nl;
        auto key = std::make_pair(vi, ni);
        auto it = weld.find(key);
        int id;
        if (it == weld.end()) {
            id = (int)(m.pos.size() / 3);
            weld[key] = id;
            float px = P[3 * vi], py = P[3 * vi + 1], pz = P[3 * vi + 2];
            float nx = 0, ny = 1, nz = 0;
            if (N) { nx = (*N)[3 * ni]; ny = (*N)[3 * ni + 1]; nz = (*N)[3 * ni + 2]; }
            if (zUp) {   // rotate -90 degrees about X: (x,y,z) -> (x, z, -y); keeps winding
                float t = py; py = pz; pz = -t;
                t = ny; ny = nz; nz = -t;
            }
            m.pos.push_back(px); m.pos.push_back(py); m.pos.push_back(pz);
            m.nrm.push_back(nx); m.nrm.push_back(ny); m.nrm.push_back(nz);
        } else {
            id = it->second;
        }
        m.idx.push_back(id);
    }
    m.idx.resize(m.idx.size() / 3 * 3);
    if (m.idx.empty()) { setLastError("no triangle in <p>"); return RTAMD_ERR_INVALID_ARG; }
    return emit(m, out);
}

}  // namespace

extern "C" {

int rtamd_host_mesh_load_dae(const char* path, rtamd_host_mesh* out) {
    if (!path || !out) return RTAMD_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof(*out));
    // (a damaged file is an error code, whatever it makes the standard library throw)
    try {
        return loadDae(path, out);
    } catch (const std::bad_alloc&) {
        setLastError("out of memory reading a COLLADA file");
        return RTAMD_ERR_OUT_OF_MEMORY;
    } catch (const std::exception& ex) {
        setLastError(std::string("COLLADA file: ") + ex.what());
        return RTAMD_ERR_INVALID_ARG;
    }
}

int rtamd_host_mesh_subdivide(const rtamd_host_mesh* in, int32_t rounds, rtamd_host_mesh* out) {
    if (!in || !out || rounds < 0) return RTAMD_ERR_INVALID_ARG;
    MeshBuf cur;
    cur.pos.assign(in->positions, in->positions + 3 * (size_t)in->n_vertices);
    cur.nrm.assign(in->normals, in->normals + 3 * (size_t)in->n_vertices);
    cur.idx.assign(in->indices, in->indices + 3 * (size_t)in->n_triangles);
    for (int r = 0; r < rounds; r++) {
        MeshBuf nx;
        nx.pos = cur.pos; nx.nrm = cur.nrm;
        std::map<std::pair<int, int>, int> mid;
        auto midpoint = [&](int a, int b) {
            auto key = a < b ? std::make_pair(a, b) : std::make_pair(b, a);
            auto it = mid.find(key);
            if (it != mid.end()) return it->second;
            int id = (int)(nx.pos.size() / 3);
            float n[3];
            for (int k = 0; k < 3; k++) {
                nx.pos.push_back((cur.pos[3 * key.first + k] + cur.pos[3 * key.second + k]) * 0.5f);
                n[k] = cur.nrm[3 * key.first + k] + cur.nrm[3 * key.second + k];
            }
            float l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (!(l > 0)) { n[0] = cur.nrm[3 * key.first]; n[1] = cur.nrm[3 * key.first + 1]; n[2] = cur.nrm[3 * key.first + 2]; l = 1; }
            for (int k = 0; k < 3; k++) nx.nrm.push_back(n[k] / l);
            mid[key] = id;
            return id;
        };
        size_t nt = cur.idx.size() / 3;
        nx.idx.reserve(nt * 12);
        for (size_t t = 0; t < nt; t++) {
            int a = cur.idx[3 * t], b = cur.idx[3 * t + 1], c = cur.idx[3 * t + 2];
            int ab = midpoint(a, b), bc = midpoint(b, c), ca = midpoint(c, a);
            const int tri[12] = {a, ab, ca, ab, b, bc, ca, bc, c, ab, bc, ca};
            nx.idx.insert(nx.idx.end(), tri, tri + 12);
        }
        cur = std::move(nx);
    }
    std::memset(out, 0, sizeof(*out));
    return emit(cur, out);
}

}  // extern "C"

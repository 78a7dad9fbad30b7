// HDR image files for the sky texture and for framebuffer dumps.
//
// The reference samples `_SkyboxTexture` (Assets/Shaders/RayTracingShader.compute:22,409), a
// Unity-imported lat-long .exr (Assets/SkyBoxes/*.exr.meta; the pixel data is not part of
// the checkout).  Outside Unity the texture has to be read from the file itself:
//   * OpenEXR, single-part scanline files, HALF / FLOAT / UINT channels named R,G,B,A (or Y),
//     compression NONE, RLE, ZIPS, ZIP, PIZ (exr_piz.cpp).  The lossy codecs are reported as
//     unsupported.
//   * Radiance RGBE (.hdr / .pic), flat and new-style run-length scanlines, -Y +X order.
// Image files store the top scanline first; Unity's texture space has v = 0 at the bottom
// row, so loaders and writers take `flip_y` (1 = convert between the two).
// Writers: .exr (FLOAT RGBA, ZIP), .hdr (flat RGBE), .ppm (8-bit, values clamped to [0,1]).

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <stdexcept>

#include "host_internal.h"

namespace {

using rtamd_host::setLastError;

bool readFile(const char* path, std::vector<uint8_t>& out) {
    FILE* f = std::fopen(path, "rb");
    if (!f) { setLastError(std::string("cannot open ") + path); return false; }
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) { std::fclose(f); setLastError("cannot size file"); return false; }
    out.resize((size_t)n);
    size_t got = n ? std::fread(out.data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    if (got != (size_t)n) { setLastError("short read"); return false; }
    return true;
}

float halfToFloat(uint16_t h) {
    uint32_t sign = (uint32_t)(h >> 15) << 31;
    uint32_t exp = (h >> 10) & 31u, man = h & 1023u;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 1024u));
            bits = sign | (uint32_t)(127 - 15 - e) << 23 | (man & 1023u) << 13;
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
    else bits = sign | (exp + 112u) << 23 | man << 13;
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

struct Reader {
    const uint8_t* p; size_t n, at = 0; bool ok = true;
    bool need(size_t k) { if (k > n || at > n - k) { ok = false; return false; } return true; }   // (no wrap-around for a hostile `at`)
    uint8_t u8() { if (!need(1)) return 0; return p[at++]; }
    int32_t i32() { if (!need(4)) return 0; int32_t v; std::memcpy(&v, p + at, 4); at += 4; return v; }
    uint64_t u64() { if (!need(8)) return 0; uint64_t v; std::memcpy(&v, p + at, 8); at += 8; return v; }
    std::string str(size_t maxLen = 256) {
        std::string s;
        while (ok) {
            if (!need(1)) break;
            char c = (char)p[at++];
            if (!c) break;
            s.push_back(c);
            if (s.size() > maxLen) { ok = false; break; }
        }
        return s;
    }
};

// Undo OpenEXR's byte predictor and even/odd de-interleave (ZIP, ZIPS and RLE blocks).
void exrUnfilter(std::vector<uint8_t>& buf, std::vector<uint8_t>& tmp) {
    const size_t n = buf.size();
    for (size_t i = 1; i < n; i++) buf[i] = (uint8_t)(buf[i - 1] + buf[i] - 128);
    tmp.resize(n);
    const size_t half = (n + 1) / 2;
    for (size_t i = 0; i < n; i++) tmp[i] = (i & 1) ? buf[half + i / 2] : buf[i / 2];
    buf.swap(tmp);
}

bool exrRleDecode(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expect) {
    out.clear();
    size_t i = 0;
    while (i < n) {
        int8_t c = (int8_t)src[i++];
        if (c < 0) {
            size_t k = (size_t)(-(int)c);
            if (i + k > n) return false;
            out.insert(out.end(), src + i, src + i + k);
            i += k;
        } else {
            if (i >= n) return false;
            out.insert(out.end(), (size_t)c + 1, src[i++]);
        }
        if (out.size() > expect) return false;
    }
    return out.size() == expect;
}

struct ExrChannel { std::string name; int type; int xs, ys; };

bool loadExr(const std::vector<uint8_t>& file, bool flip, std::vector<float>& rgba, int& W, int& H) {
    Reader r{file.data(), file.size()};
    if (r.i32() != 20000630) { setLastError("not an OpenEXR file"); return false; }
    int32_t version = r.i32();
    if ((version & 0xff) != 2) { setLastError("unsupported OpenEXR version"); return false; }
    if (version & 0x1a00) { setLastError("tiled / multi-part / deep OpenEXR files are not supported"); return false; }
    std::vector<ExrChannel> ch;
    int compression = -1, lineOrder = 0;
    int dw[4] = {0, 0, -1, -1};
    for (;;) {
        std::string name = r.str();
        if (!r.ok) { setLastError("truncated OpenEXR header"); return false; }
        if (name.empty()) break;
        std::string type = r.str();
        int32_t size = r.i32();
        if (!r.ok || size < 0 || !r.need((size_t)size)) { setLastError("truncated OpenEXR header"); return false; }
        Reader a{file.data() + r.at, (size_t)size};
        if (name == "channels" && type == "chlist") {
            for (;;) {
                std::string cn = a.str();
                if (!a.ok || cn.empty()) break;
                ExrChannel c;
                c.name = cn; c.type = a.i32(); a.u8(); a.u8(); a.u8(); a.u8(); c.xs = a.i32(); c.ys = a.i32();
                if (!a.ok) { setLastError("bad channel list"); return false; }
                ch.push_back(c);
            }
        } else if (name == "compression") compression = a.u8();
        else if (name == "dataWindow") { for (int i = 0; i < 4; i++) dw[i] = a.i32(); }
        else if (name == "lineOrder") lineOrder = a.u8();
        r.at += (size_t)size;
    }
    (void)lineOrder;   // chunks carry their own y
    const int64_t W64 = (int64_t)dw[2] - (int64_t)dw[0] + 1, H64 = (int64_t)dw[3] - (int64_t)dw[1] + 1;
    if (W64 <= 0 || H64 <= 0 || W64 > (1ll << 28) || H64 > (1ll << 28) || W64 * H64 > (1ll << 28)) { setLastError("bad OpenEXR data window"); return false; }
    W = (int)W64; H = (int)H64;
    if (ch.empty()) { setLastError("OpenEXR file without channels"); return false; }
    int linesPerBlock;
    switch (compression) {
        case 0: case 1: case 2: linesPerBlock = 1; break;
        case 3: linesPerBlock = 16; break;
        case 4: linesPerBlock = 32; break;
        default: setLastError("unsupported OpenEXR compression (NONE, RLE, ZIPS, ZIP and PIZ are read; re-save lossy files)"); return false;
    }
    size_t lineBytes = 0;
    std::vector<size_t> chOffset;
    for (const ExrChannel& c : ch) {
        if (c.xs != 1 || c.ys != 1) { setLastError("sub-sampled OpenEXR channels are not supported"); return false; }
        if (c.type < 0 || c.type > 2) { setLastError("bad OpenEXR pixel type"); return false; }
        chOffset.push_back(lineBytes);
        lineBytes += (size_t)W * (c.type == 1 ? 2 : 4);
    }
    // destination component of each channel; a lone Y fills R, G and B
    std::vector<int> dst;
    bool haveA = false, haveY = false, haveRgb = false;
    for (const ExrChannel& c : ch) {
        int d = -1;
        if (c.name == "R") { d = 0; haveRgb = true; }
        else if (c.name == "G") { d = 1; haveRgb = true; }
        else if (c.name == "B") { d = 2; haveRgb = true; }
        else if (c.name == "A") { d = 3; haveA = true; }
        else if (c.name == "Y") { d = 4; haveY = true; }
        dst.push_back(d);
    }
    if (!haveRgb && !haveY) { setLastError("OpenEXR file has no R/G/B or Y channel"); return false; }
    const int nChunks = (H + linesPerBlock - 1) / linesPerBlock;
    std::vector<uint64_t> offsets((size_t)nChunks);
    for (int i = 0; i < nChunks; i++) offsets[i] = r.u64();
    if (!r.ok) { setLastError("truncated OpenEXR offset table"); return false; }
    rgba.assign((size_t)W * H * 4, 0.f);
    if (!haveA) for (size_t i = 0; i < (size_t)W * H; i++) rgba[4 * i + 3] = 1.f;
    std::vector<uint8_t> block, tmp;
    for (int ci = 0; ci < nChunks; ci++) {
        Reader c{file.data(), file.size()};
        if (offsets[ci] >= (uint64_t)file.size()) { setLastError("OpenEXR chunk offset beyond the end of the file"); return false; }
        c.at = (size_t)offsets[ci];
        int32_t y0 = c.i32();
        int32_t size = c.i32();
        if (!c.ok || size < 0 || !c.need((size_t)size)) { setLastError("truncated OpenEXR chunk"); return false; }
        const int64_t row64 = (int64_t)y0 - (int64_t)dw[1];
        const int row0 = (int)row64;
        if (row64 < 0 || row64 >= H) { setLastError("OpenEXR chunk outside the data window"); return false; }
        int rows = H - row0 < linesPerBlock ? H - row0 : linesPerBlock;
        size_t expect = lineBytes * (size_t)rows;
        const uint8_t* src = file.data() + c.at;
        if ((size_t)size == expect || compression == 0) {
            if ((size_t)size != expect) { setLastError("bad uncompressed OpenEXR chunk size"); return false; }
            block.assign(src, src + size);
        } else if (compression == 4) {
            std::vector<int> words;
            for (const ExrChannel& k : ch) words.push_back(k.type == 1 ? 1 : 2);
            std::string why;
            if (!rtamd_host::exrPizDecode(src, (size_t)size, words, W, rows, block, why)) { setLastError("bad PIZ data in OpenEXR chunk: " + why); return false; }
            if (block.size() != expect) { setLastError("bad PIZ data in OpenEXR chunk"); return false; }
        } else if (compression == 1) {
            if (!exrRleDecode(src, (size_t)size, block, expect)) { setLastError("bad RLE data in OpenEXR chunk"); return false; }
            exrUnfilter(block, tmp);
        } else {
            block.resize(expect);
            uLongf outLen = (uLongf)expect;
            if (uncompress(block.data(), &outLen, src, (uLong)size) != Z_OK || outLen != expect) {
                setLastError("bad zlib data in OpenEXR chunk");
                return false;
            }
            exrUnfilter(block, tmp);
        }
        for (int rr = 0; rr < rows; rr++) {
            const int row = row0 + rr;
            const int outRow = flip ? H - 1 - row : row;
            float* o = rgba.data() + (size_t)outRow * W * 4;
            const uint8_t* line = block.data() + lineBytes * (size_t)rr;
            for (size_t k = 0; k < ch.size(); k++) {
                if (dst[k] < 0) continue;
                const uint8_t* p = line + chOffset[k];
                for (int x = 0; x < W; x++) {
                    float v;
                    if (ch[k].type == 1) { uint16_t h; std::memcpy(&h, p + 2 * x, 2); v = halfToFloat(h); }
                    else if (ch[k].type == 2) std::memcpy(&v, p + 4 * x, 4);
                    else { uint32_t u; std::memcpy(&u, p + 4 * x, 4); v = (float)u; }
                    if (dst[k] == 4) { if (!haveRgb) { o[4 * x] = v; o[4 * x + 1] = v; o[4 * x + 2] = v; } }
                    else o[4 * x + dst[k]] = v;
                }
            }
        }
    }
    return true;
}

bool loadHdr(const std::vector<uint8_t>& file, bool flip, std::vector<float>& rgba, int& W, int& H) {
    size_t at = 0;
    auto line = [&](std::string& s) {
        s.clear();
        while (at < file.size() && file[at] != '\n') s.push_back((char)file[at++]);
        if (at >= file.size()) return false;
        at++;
        return true;
    };
    std::string s;
    if (!line(s) || (s.rfind("#?", 0) != 0)) { setLastError("not a Radiance HDR file"); return false; }
    bool rgbe = false;
    for (;;) {
        if (!line(s)) { setLastError("truncated Radiance header"); return false; }
        if (s.empty() || s == "\r") break;
        if (s.rfind("FORMAT=32-bit_rle_rgbe", 0) == 0) rgbe = true;
    }
    if (!rgbe) { setLastError("Radiance file is not 32-bit_rle_rgbe"); return false; }
    if (!line(s)) { setLastError("truncated Radiance header"); return false; }
    int h = 0, w = 0;
    if (std::sscanf(s.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || (int64_t)w * h > (1ll << 28)) {
        setLastError("unsupported Radiance resolution line (only -Y H +X W)");
        return false;
    }
    W = w; H = h;
    rgba.assign((size_t)W * H * 4, 1.f);
    std::vector<uint8_t> scan((size_t)W * 4);
    for (int y = 0; y < H; y++) {
        bool rle = false;
        if (W >= 8 && W < 32768 && at + 4 <= file.size() && file[at] == 2 && file[at + 1] == 2 && !(file[at + 2] & 0x80)) {
            if (((int)file[at + 2] << 8 | file[at + 3]) != W) { setLastError("Radiance scanline width mismatch"); return false; }
            rle = true;
            at += 4;
        }
        if (rle) {
            for (int c = 0; c < 4; c++) {
                int x = 0;
                while (x < W) {
                    if (at >= file.size()) { setLastError("truncated Radiance data"); return false; }
                    int count = file[at++];
                    if (count > 128) {
                        count -= 128;
                        if (at >= file.size() || x + count > W) { setLastError("bad Radiance run"); return false; }
                        uint8_t v = file[at++];
                        for (int k = 0; k < count; k++) scan[(size_t)(x++) * 4 + c] = v;
                    } else {
                        if (count == 0 || at + (size_t)count > file.size() || x + count > W) { setLastError("bad Radiance run"); return false; }
                        for (int k = 0; k < count; k++) scan[(size_t)(x++) * 4 + c] = file[at++];
                    }
                }
            }
        } else {
            if (at + (size_t)W * 4 > file.size()) { setLastError("truncated Radiance data"); return false; }
            std::memcpy(scan.data(), file.data() + at, (size_t)W * 4);
            at += (size_t)W * 4;
        }
        float* o = rgba.data() + (size_t)(flip ? H - 1 - y : y) * W * 4;
        for (int x = 0; x < W; x++) {
            const uint8_t* p = scan.data() + (size_t)x * 4;
            // value = mantissa * 2^(e - 136); e = 0 is black
            float f = p[3] ? std::ldexp(1.0f, (int)p[3] - 136) : 0.f;
            o[4 * x] = p[0] * f; o[4 * x + 1] = p[1] * f; o[4 * x + 2] = p[2] * f; o[4 * x + 3] = 1.f;
        }
    }
    return true;
}

bool endsWith(const std::string& s, const char* suffix) {
    size_t n = std::strlen(suffix);
    if (s.size() < n) return false;
    for (size_t i = 0; i < n; i++) {
        char a = s[s.size() - n + i], b = suffix[i];
        if (a >= 'A' && a <= 'Z') a = (char)(a - 'A' + 'a');
        if (a != b) return false;
    }
    return true;
}

void put32(std::vector<uint8_t>& o, int32_t v) { uint8_t b[4]; std::memcpy(b, &v, 4); o.insert(o.end(), b, b + 4); }
void putStr(std::vector<uint8_t>& o, const char* s) { o.insert(o.end(), s, s + std::strlen(s) + 1); }
void putAttr(std::vector<uint8_t>& o, const char* name, const char* type, const void* data, int32_t size) {
    putStr(o, name); putStr(o, type); put32(o, size);
    o.insert(o.end(), (const uint8_t*)data, (const uint8_t*)data + size);
}

bool saveExr(const char* path, const float* rgba, int W, int H, bool flip) {
    std::vector<uint8_t> o;
    put32(o, 20000630); put32(o, 2);
    {
        std::vector<uint8_t> cl;
        for (const char* n : {"A", "B", "G", "R"}) {     // alphabetical, as the format requires
            putStr(cl, n); put32(cl, 2); put32(cl, 0); put32(cl, 1); put32(cl, 1);
        }
        cl.push_back(0);
        putAttr(o, "channels", "chlist", cl.data(), (int32_t)cl.size());
    }
    uint8_t comp = 3, order = 0;
    putAttr(o, "compression", "compression", &comp, 1);
    int32_t win[4] = {0, 0, W - 1, H - 1};
    putAttr(o, "dataWindow", "box2i", win, 16);
    putAttr(o, "displayWindow", "box2i", win, 16);
    putAttr(o, "lineOrder", "lineOrder", &order, 1);
    float one = 1.f, centre[2] = {0.f, 0.f};
    putAttr(o, "pixelAspectRatio", "float", &one, 4);
    putAttr(o, "screenWindowCenter", "v2f", centre, 8);
    putAttr(o, "screenWindowWidth", "float", &one, 4);
    o.push_back(0);
    const int lines = 16, nChunks = (H + lines - 1) / lines;
    const size_t tableAt = o.size();
    o.resize(o.size() + (size_t)nChunks * 8);
    std::vector<uint8_t> raw, filt;
    const int compOfChannel[4] = {3, 2, 1, 0};
    for (int ci = 0; ci < nChunks; ci++) {
        uint64_t off = o.size();
        std::memcpy(o.data() + tableAt + (size_t)ci * 8, &off, 8);
        const int y0 = ci * lines, rows = H - y0 < lines ? H - y0 : lines;
        raw.resize((size_t)rows * W * 16);
        size_t at = 0;
        for (int rr = 0; rr < rows; rr++) {
            const float* src = rgba + (size_t)(flip ? H - 1 - (y0 + rr) : y0 + rr) * W * 4;
            for (int c = 0; c < 4; c++)
                for (int x = 0; x < W; x++) { std::memcpy(raw.data() + at, src + 4 * x + compOfChannel[c], 4); at += 4; }
        }
        // even/odd split, then the byte predictor
        const size_t n = raw.size(), half = (n + 1) / 2;
        filt.resize(n);
        for (size_t i = 0; i < n; i++) filt[(i & 1) ? half + i / 2 : i / 2] = raw[i];
        for (size_t i = n; i-- > 1;) filt[i] = (uint8_t)(filt[i] - filt[i - 1] + 128);
        uLongf bound = compressBound((uLong)n);
        std::vector<uint8_t> z(bound);
        if (compress2(z.data(), &bound, filt.data(), (uLong)n, 6) != Z_OK) { setLastError("zlib failure"); return false; }
        put32(o, y0);
        if (bound < n) { put32(o, (int32_t)bound); o.insert(o.end(), z.begin(), z.begin() + bound); }
        else { put32(o, (int32_t)n); o.insert(o.end(), raw.begin(), raw.end()); }
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) { setLastError(std::string("cannot create ") + path); return false; }
    bool ok = std::fwrite(o.data(), 1, o.size(), f) == o.size();
    std::fclose(f);
    if (!ok) setLastError("short write");
    return ok;
}

bool saveHdr(const char* path, const float* rgba, int W, int H, bool flip) {
    FILE* f = std::fopen(path, "wb");
    if (!f) { setLastError(std::string("cannot create ") + path); return false; }
    std::fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", H, W);
    std::vector<uint8_t> scan((size_t)W * 4);
    bool ok = true;
    for (int y = 0; y < H && ok; y++) {
        const float* src = rgba + (size_t)(flip ? H - 1 - y : y) * W * 4;
        for (int x = 0; x < W; x++) {
            float r = src[4 * x], g = src[4 * x + 1], b = src[4 * x + 2];
            r = r > 0.f ? r : 0.f; g = g > 0.f ? g : 0.f; b = b > 0.f ? b : 0.f;
            float m = r > g ? (r > b ? r : b) : (g > b ? g : b);
            uint8_t* p = scan.data() + (size_t)x * 4;
            if (!(m >= 1e-32f) || !(m < 1e38f)) { p[0] = p[1] = p[2] = p[3] = 0; continue; }
            int e;
            float frac = std::frexp(m, &e);             // m = frac * 2^e, frac in [0.5, 1)
            float scale = frac * 256.0f / m;
            p[0] = (uint8_t)(r * scale); p[1] = (uint8_t)(g * scale); p[2] = (uint8_t)(b * scale);
            p[3] = (uint8_t)(e + 128);
        }
        ok = std::fwrite(scan.data(), 1, scan.size(), f) == scan.size();
    }
    std::fclose(f);
    if (!ok) setLastError("short write");
    return ok;
}

bool savePpm(const char* path, const float* rgba, int W, int H, bool flip) {
    FILE* f = std::fopen(path, "wb");
    if (!f) { setLastError(std::string("cannot create ") + path); return false; }
    std::fprintf(f, "P6\n%d %d\n255\n", W, H);
    std::vector<uint8_t> scan((size_t)W * 3);
    bool ok = true;
    for (int y = 0; y < H && ok; y++) {
        const float* src = rgba + (size_t)(flip ? H - 1 - y : y) * W * 4;
        for (int x = 0; x < W; x++)
            for (int c = 0; c < 3; c++) {
                float v = src[4 * x + c];
                v = v > 0.f ? (v < 1.f ? v : 1.f) : 0.f;      // UNORM8 render target: values clamp at 1 (RayTracingPass.cs:58)
                scan[(size_t)x * 3 + c] = (uint8_t)std::floor(v * 255.0f + 0.5f);
            }
        ok = std::fwrite(scan.data(), 1, scan.size(), f) == scan.size();
    }
    std::fclose(f);
    if (!ok) setLastError("short write");
    return ok;
}

}  // namespace

extern "C" {

int rtamd_host_image_load(const char* path, int32_t flip_y, float** out_rgba, int32_t* out_w, int32_t* out_h) {
    if (!path || !out_rgba || !out_w || !out_h) { setLastError("NULL argument"); return -1; }
    *out_rgba = nullptr; *out_w = *out_h = 0;
    // (a damaged file is an error code, whatever sizes it claims)
    try {
        std::vector<uint8_t> file;
        if (!readFile(path, file)) return -1;
        std::vector<float> rgba;
        int W = 0, H = 0;
        bool ok;
        if (file.size() >= 4 && file[0] == 0x76 && file[1] == 0x2f && file[2] == 0x31 && file[3] == 0x01) ok = loadExr(file, flip_y != 0, rgba, W, H);
        else if (file.size() >= 2 && file[0] == '#' && file[1] == '?') ok = loadHdr(file, flip_y != 0, rgba, W, H);
        else { setLastError("unknown image format (expected OpenEXR or Radiance RGBE)"); ok = false; }
        if (!ok) return -1;
        float* p = (float*)std::malloc(rgba.size() * sizeof(float));
        if (!p) { setLastError("out of memory"); return -1; }
        std::memcpy(p, rgba.data(), rgba.size() * sizeof(float));
        *out_rgba = p; *out_w = W; *out_h = H;
        return 0;
    } catch (const std::bad_alloc&) {
        setLastError("out of memory reading an image");
        return -1;
    } catch (const std::exception& ex) {
        setLastError(std::string("image file: ") + ex.what());
        return -1;
    }
}

void rtamd_host_image_free(float* rgba) { std::free(rgba); }

int rtamd_host_image_save(const char* path, const float* rgba, int32_t w, int32_t h, int32_t flip_y) {
    if (!path || !rgba || w <= 0 || h <= 0) { setLastError("NULL argument or bad size"); return -1; }
    std::string p(path);
    bool ok;
    if (endsWith(p, ".exr")) ok = saveExr(path, rgba, w, h, flip_y != 0);
    else if (endsWith(p, ".hdr") || endsWith(p, ".pic")) ok = saveHdr(path, rgba, w, h, flip_y != 0);
    else if (endsWith(p, ".ppm")) ok = savePpm(path, rgba, w, h, flip_y != 0);
    else { setLastError("unknown image extension (use .exr, .hdr or .ppm)"); ok = false; }
    return ok ? 0 : -1;
}

}  // extern "C"

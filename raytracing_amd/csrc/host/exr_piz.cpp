// OpenEXR PIZ blocks (compression 4): range compaction through a bitmap of the 16-bit values in use,
// a two-dimensional Haar-like wavelet per channel, and a canonical Huffman code with one run-length symbol.
// This is the codec HDR skies are most often saved with; the reference's sky texture
// (Assets/Shaders/RayTracingShader.compute:22,409; Assets/SkyBoxes/*.exr.meta) is such a file.
//
// Layout of a compressed block:
//   u16 minNonZero, u16 maxNonZero       first / last byte of the bitmap that is stored
//   bitmap[minNonZero .. maxNonZero]     bit v set = value v occurs (value 0 is always taken to occur)
//   i32 length                           bytes of Huffman data that follow
//   Huffman data:  u32 im, u32 iM, u32 tableLength, u32 nBits, u32 reserved,
//                  code lengths of the symbols im..iM, 6 bits each (59..62 = run of 2..5 zero lengths,
//                  63 + 8 bits = run of 6..261), then nBits bits of codes, most significant bit first.
//                  Symbol iM is "repeat the last value `next 8 bits` more times".
// Decoded words are channel-major ([channel][row][x][word of the pixel]); the caller's layout is
// row-major ([row][channel][x]), little-endian.
#include <cstring>

#include "host_internal.h"

namespace {

constexpr int kEncBits = 16, kDecBits = 14;
constexpr int kEncSize = (1 << kEncBits) + 1;      // 65536 values + the run-length symbol
constexpr int kDecSize = 1 << kDecBits;
constexpr int kShortZeroRun = 59, kLongZeroRun = 63, kShortestLongRun = 2 + kLongZeroRun - kShortZeroRun;

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint64_t c = 0; int lc = 0; bool ok = true;
    uint32_t bits(int n) {
        while (lc < n) {
            if (p >= end) { ok = false; return 0; }
            c = (c << 8) | *p++;
            lc += 8;
        }
        lc -= n;
        return (uint32_t)((c >> lc) & ((1ull << n) - 1ull));
    }
};

// code lengths -> canonical codes: within a length codes count upwards in symbol order, and the codes of a
// longer length come (as bit strings, left-aligned) before those of a shorter one.  code[i] = code << 6 | length
bool canonicalCodes(std::vector<uint64_t>& code) {
    uint64_t n[59];
    std::memset(n, 0, sizeof(n));
    for (uint64_t l : code) { if (l > 58) return false; n[l]++; }
    uint64_t c = 0;
    for (int i = 58; i > 0; i--) {
        const uint64_t nc = (c + n[i]) >> 1;
        n[i] = c;
        c = nc;
    }
    for (uint64_t& h : code) {
        const int l = (int)h;
        if (l > 0) h = (uint64_t)l | (n[l]++ << 6);
    }
    return true;
}

struct DecEntry { int len = 0; int lit = 0; std::vector<int> longer; };   // len > 0: a code of <= 14 bits for symbol lit

bool huffmanDecode(const uint8_t* src, size_t n, std::vector<uint16_t>& out, size_t nOut, std::string& why) {
    out.clear();
    if (n == 0) { if (nOut) { why = "PIZ block without Huffman data"; return false; } return true; }
    if (n < 20) { why = "truncated PIZ Huffman header"; return false; }
    uint32_t im, iM, nBits;
    std::memcpy(&im, src, 4); std::memcpy(&iM, src + 4, 4); std::memcpy(&nBits, src + 12, 4);
    if (im >= (uint32_t)kEncSize || iM >= (uint32_t)kEncSize || im > iM) { why = "bad PIZ Huffman symbol range"; return false; }
    std::vector<uint64_t> code((size_t)kEncSize, 0);
    BitReader br{src + 20, src + n};
    for (uint32_t s = im; s <= iM; s++) {
        const uint32_t l = br.bits(6);
        if (!br.ok) { why = "truncated PIZ Huffman table"; return false; }
        if (l == (uint32_t)kLongZeroRun || l >= (uint32_t)kShortZeroRun) {
            uint32_t run = l == (uint32_t)kLongZeroRun ? br.bits(8) + (uint32_t)kShortestLongRun : l - (uint32_t)kShortZeroRun + 2u;
            if (!br.ok) { why = "truncated PIZ Huffman table"; return false; }
            if (s + run > iM + 1) { why = "PIZ Huffman table runs past its last symbol"; return false; }
            s += run - 1;       // (the lengths are zero already)
        } else code[s] = l;
    }
    if (!canonicalCodes(code)) { why = "bad PIZ Huffman code length"; return false; }
    const uint8_t* data = br.p;       // the table ends on a byte boundary: what is left in the bit buffer is padding
    if ((uint64_t)nBits > 8ull * (uint64_t)(src + n - data)) { why = "PIZ Huffman data shorter than its bit count"; return false; }

    std::vector<DecEntry> table((size_t)kDecSize);
    for (uint32_t s = im; s <= iM; s++) {
        const uint64_t c = code[s] >> 6;
        const int l = (int)(code[s] & 63);
        if (l == 0) continue;
        if (c >> l) { why = "bad PIZ Huffman code"; return false; }
        if (l > kDecBits) {
            DecEntry& e = table[(size_t)(c >> (l - kDecBits))];
            if (e.len) { why = "bad PIZ Huffman code"; return false; }
            e.longer.push_back((int)s);
        } else {
            const size_t first = (size_t)(c << (kDecBits - l)), count = (size_t)1 << (kDecBits - l);
            for (size_t i = first; i < first + count; i++) {
                if (table[i].len || !table[i].longer.empty()) { why = "bad PIZ Huffman code"; return false; }
                table[i].len = l; table[i].lit = (int)s;
            }
        }
    }

    out.reserve(nOut);
    const int runSymbol = (int)iM;
    const uint8_t* in = data;
    const uint8_t* const inEnd = data + (nBits + 7u) / 8u;
    uint64_t c = 0;
    int lc = 0;
    auto emit = [&](int sym) -> bool {
        if (sym == runSymbol) {
            if (lc < 8) {
                if (in >= inEnd) { why = "truncated PIZ Huffman data"; return false; }
                c = (c << 8) | *in++; lc += 8;
            }
            lc -= 8;
            const size_t run = (size_t)((c >> lc) & 0xffu);
            if (out.empty()) { why = "PIZ run before any value"; return false; }
            if (out.size() + run > nOut) { why = "PIZ block decodes to too many values"; return false; }
            out.insert(out.end(), run, out.back());
        } else {
            if (out.size() >= nOut) { why = "PIZ block decodes to too many values"; return false; }
            out.push_back((uint16_t)sym);
        }
        return true;
    };
    while (in < inEnd) {
        c = (c << 8) | *in++; lc += 8;
        while (lc >= kDecBits) {
            const DecEntry& e = table[(size_t)((c >> (lc - kDecBits)) & (uint64_t)(kDecSize - 1))];
            if (e.len) {
                lc -= e.len;
                if (!emit(e.lit)) return false;
            } else {
                if (e.longer.empty()) { why = "bad code in PIZ Huffman data"; return false; }
                bool found = false;
                for (int s : e.longer) {
                    const int l = (int)(code[(size_t)s] & 63);
                    while (lc < l && in < inEnd) { c = (c << 8) | *in++; lc += 8; }
                    if (lc >= l && (code[(size_t)s] >> 6) == ((c >> (lc - l)) & ((1ull << l) - 1ull))) {
                        lc -= l;
                        if (!emit(s)) return false;
                        found = true;
                        break;
                    }
                }
                if (!found) { why = "bad code in PIZ Huffman data"; return false; }
            }
        }
    }
    // what is left: the padding bits of the last byte go, short codes remain
    const int pad = (int)((8u - nBits) & 7u);
    c >>= pad;
    lc -= pad;
    while (lc > 0) {
        const DecEntry& e = table[(size_t)((c << (kDecBits - lc)) & (uint64_t)(kDecSize - 1))];
        if (!e.len || e.len > lc) { why = "bad code in PIZ Huffman data"; return false; }
        lc -= e.len;
        if (!emit(e.lit)) return false;
    }
    if (out.size() != nOut) { why = "PIZ block decodes to too few values"; return false; }
    return true;
}

// inverse of the wavelet's basic step: (average, difference) -> the two values
inline void unAvgDiff14(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int ls = (int16_t)l, hs = (int16_t)h;
    const int ai = ls + (hs & 1) + (hs >> 1);
    a = (uint16_t)(int16_t)ai;
    b = (uint16_t)(int16_t)(ai - hs);
}
// the same modulo 2^16, for data that uses more than 14 bits
inline void unAvgDiff16(uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) {
    const int m = l, d = h;
    const int bb = (m - (d >> 1)) & 0xffff;
    const int aa = (d + bb - 0x8000) & 0xffff;
    b = (uint16_t)bb;
    a = (uint16_t)aa;
}

// in: nx x ny values, ox / oy apart in x / y; mx: largest value the data held before the transform
void waveletDecode(uint16_t* in, int nx, int ox, int ny, int oy, uint16_t mx) {
    const bool w14 = mx < (1 << 14);
    const int n = nx > ny ? ny : nx;
    int p = 1;
    while (p <= n) p <<= 1;
    p >>= 1;
    int p2 = p;
    p >>= 1;
    auto step = [&](uint16_t l, uint16_t h, uint16_t& a, uint16_t& b) { if (w14) unAvgDiff14(l, h, a, b); else unAvgDiff16(l, h, a, b); };
    while (p >= 1) {
        const long oy1 = (long)oy * p, ox1 = (long)ox * p, ox2 = (long)ox * p2;
        long y = 0;
        for (; y <= ny - p2; y += p2) {
            uint16_t* px = in + (long)oy * y;
            long x = 0;
            for (; x <= nx - p2; x += p2, px += ox2) {
                uint16_t* p01 = px + ox1;
                uint16_t* p10 = px + oy1;
                uint16_t* p11 = p10 + ox1;
                uint16_t i00, i01, i10, i11;
                step(*px, *p10, i00, i10);
                step(*p01, *p11, i01, i11);
                step(i00, i01, *px, *p01);
                step(i10, i11, *p10, *p11);
            }
            if (nx & p) {       // a column without a partner: vertical step only
                uint16_t* p10 = px + oy1;
                uint16_t i00;
                step(*px, *p10, i00, *p10);
                *px = i00;
            }
        }
        if (ny & p) {           // a row without a partner: horizontal steps only
            uint16_t* px = in + (long)oy * y;
            for (long x = 0; x <= nx - p2; x += p2, px += ox2) {
                uint16_t* p01 = px + ox1;
                uint16_t i00;
                step(*px, *p01, i00, *p01);
                *px = i00;
            }
        }
        p2 = p;
        p >>= 1;
    }
}

}  // namespace

namespace rtamd_host {

bool exrPizDecode(const uint8_t* src, size_t n, const std::vector<int>& chanWords, int width, int rows, std::vector<uint8_t>& out, std::string& why) {
    size_t wordsPerRow = 0;
    for (int w : chanWords) wordsPerRow += (size_t)w * (size_t)width;
    const size_t nWords = wordsPerRow * (size_t)rows;
    if (n < 4) { why = "truncated PIZ block"; return false; }
    uint16_t minNonZero, maxNonZero;
    std::memcpy(&minNonZero, src, 2); std::memcpy(&maxNonZero, src + 2, 2);
    size_t at = 4;
    std::vector<uint8_t> bitmap(8192, 0);
    if (maxNonZero >= 8192) { why = "bad PIZ bitmap range"; return false; }
    if (minNonZero <= maxNonZero) {
        const size_t len = (size_t)maxNonZero - minNonZero + 1;
        if (n - at < len) { why = "truncated PIZ bitmap"; return false; }
        std::memcpy(bitmap.data() + minNonZero, src + at, len);
        at += len;
    }
    // rank -> value, for the values in use (0 always is)
    std::vector<uint16_t> lut(65536, 0);
    int k = 0;
    for (int v = 0; v < 65536; v++)
        if (v == 0 || (bitmap[(size_t)v >> 3] & (1u << (v & 7)))) lut[(size_t)k++] = (uint16_t)v;
    const uint16_t maxValue = (uint16_t)(k - 1);
    if (n - at < 4) { why = "truncated PIZ block"; return false; }
    int32_t length;
    std::memcpy(&length, src + at, 4);
    at += 4;
    if (length < 0 || (size_t)length > n - at) { why = "bad PIZ Huffman length"; return false; }
    std::vector<uint16_t> words;
    if (!huffmanDecode(src + at, (size_t)length, words, nWords, why)) return false;
    // per channel: one transform per 16-bit word of the pixel type
    size_t start = 0;
    std::vector<size_t> chanStart;
    for (int w : chanWords) {
        chanStart.push_back(start);
        for (int j = 0; j < w; j++) waveletDecode(words.data() + start + j, width, w, rows, width * w, maxValue);
        start += (size_t)w * (size_t)width * (size_t)rows;
    }
    for (uint16_t& v : words) v = lut[v];
    // channel-major -> row-major
    out.resize(nWords * 2);
    size_t o = 0;
    for (int r = 0; r < rows; r++)
        for (size_t c = 0; c < chanWords.size(); c++) {
            const size_t cnt = (size_t)chanWords[c] * (size_t)width;
            std::memcpy(out.data() + o, words.data() + chanStart[c] + cnt * (size_t)r, cnt * 2);
            o += cnt * 2;
        }
    return true;
}

}  // namespace rtamd_host

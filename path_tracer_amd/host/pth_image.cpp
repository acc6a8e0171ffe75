// pth_image.cpp -- texture file decode (PNG, baseline JPEG) and PNG encode for the host mirror.
// The reference decodes through gltf 1.4.1 -> image 0.25.6 -> png 0.17.16 / zune-jpeg 0.4.14
// (material.rs:64-65) and saves through image::save (renderer.rs:805-815); none of those crates
// is vendored, so the published formats are implemented directly: PNG (RFC 2083, zlib inflate
// from the system libz) and JPEG (ITU T.81 baseline / extended sequential Huffman, JFIF YCbCr,
// libjpeg-style triangle chroma upsampling).  Texel-level parity with zune-jpeg is unpinned
// (IDCT rounding may differ by 1 LSB).
#include "pth_image.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace pth {

namespace {

bool read_file(const std::string &path, std::vector<uint8_t> &out, std::string &err)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open " + path; return false; }
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    if (n < 0) { err = "cannot size " + path; return false; }      // not seekable (a directory, a pipe)
    f.seekg(0);
    out.resize((size_t)n);
    if (n && !f.read((char *)out.data(), n)) { err = "cannot read " + path; return false; }
    return true;
}

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint32_t be16(const uint8_t *p) { return ((uint32_t)p[0] << 8) | p[1]; }

// ------------------------------------------------------------------ PNG
bool decode_png(const std::vector<uint8_t> &d, Image &img, std::string &err)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() < 8 || memcmp(d.data(), sig, 8) != 0) { err = "not a PNG"; return false; }
    size_t p = 8;
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool have_ihdr = false, done = false;
    while (!done && p + 12 <= d.size()) {
        uint32_t len = be32(&d[p]);
        const uint8_t *type = &d[p + 4];
        if (p + 12 + (size_t)len > d.size()) { err = "PNG: truncated chunk"; return false; }
        const uint8_t *data = &d[p + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) { err = "PNG: bad IHDR"; return false; }
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) done = true;
        p += 12 + (size_t)len;
    }
    if (!have_ihdr || w == 0 || h == 0) { err = "PNG: missing IHDR"; return false; }
    if (interlace) { err = "PNG: interlaced images are not supported"; return false; }
    if (depth != 8) { err = "PNG: only 8-bit channels are supported (the reference maps 16-bit to non-sRGB formats, graphics.rs:496-503)"; return false; }
    int spp;   // samples per pixel in the file
    switch (ctype) {
    case 2: spp = 3; break;
    case 6: spp = 4; break;
    case 3: spp = 1; break;
    default: err = "PNG: grayscale textures are not supported (R8 / R8G8 formats, graphics.rs:499-501)"; return false;
    }
    size_t stride = (size_t)w * spp;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    int zr = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || raw_len != raw.size()) { err = "PNG: inflate failed"; return false; }
    // unfilter in place
    std::vector<uint8_t> pix(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *src = &raw[(stride + 1) * y];
        uint8_t *cur = &pix[stride * y];
        const uint8_t *up = y ? &pix[stride * (y - 1)] : nullptr;
        int ft = src[0];
        ++src;
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= (size_t)spp ? cur[x - spp] : 0;
            int b = up ? up[x] : 0;
            int c = (up && x >= (size_t)spp) ? up[x - spp] : 0;
            int v = src[x];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: {
                int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
                v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: err = "PNG: bad filter type"; return false;
            }
            cur[x] = (uint8_t)v;
        }
    }
    img.width = w; img.height = h;
    if (ctype == 3) {
        bool alpha = !trns.empty();
        img.channels = alpha ? 4 : 3;
        img.pixels.resize((size_t)w * h * img.channels);
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            size_t k = pix[i];
            if (k * 3 + 2 >= plte.size()) { err = "PNG: palette index out of range"; return false; }
            uint8_t *o = &img.pixels[i * img.channels];
            o[0] = plte[k * 3]; o[1] = plte[k * 3 + 1]; o[2] = plte[k * 3 + 2];
            if (alpha) o[3] = k < trns.size() ? trns[k] : 255;
        }
    } else {
        img.channels = spp;
        img.pixels.swap(pix);
    }
    return true;
}

// ------------------------------------------------------------------ JPEG
struct HuffTable {
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    uint16_t fast[512];      // (symbol << 8) | length for codes of <= 9 bits, 0 = slow path
    // false: the code lengths over-subscribe the code space (more codes of some length than a prefix code has
    // room for) -- such a table would index past `fast` and is rejected
    bool build()
    {
        int code = 0, k = 0;
        for (int i = 0; i < 512; ++i) fast[i] = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k; mincode[l] = code;
            if (code + (int)bits[l] > (1 << l)) return false;      // Kraft inequality violated
            for (int i = 0; i < bits[l]; ++i, ++k, ++code) {
                if (l <= 9) {
                    int first = code << (9 - l), n = 1 << (9 - l);
                    for (int j = 0; j < n; ++j) fast[first + j] = (uint16_t)((vals[k] << 8) | l);
                }
            }
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
        return true;
    }
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0; int n = 0;
    int marker = 0;          // pending marker hit while filling (0 = none)
    void fill()
    {
        while (n <= 24) {
            uint8_t b = 0;
            if (!marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    uint8_t b2 = (p + 1 < end) ? p[1] : 0xD9;
                    if (b2 == 0) { p += 2; }
                    else { marker = b2; b = 0; }   // feed zeros until the caller consumes the marker
                } else ++p;
            }
            acc |= (uint32_t)b << (24 - n);
            n += 8;
        }
    }
    int peek(int k) { if (n < k) fill(); return (int)(acc >> (32 - k)); }
    void skip(int k) { acc <<= k; n -= k; }
    int get(int k) { if (k == 0) return 0; int v = peek(k); skip(k); return v; }
    void reset() { acc = 0; n = 0; marker = 0; }
};

inline int huff_decode(BitReader &br, const HuffTable &t)
{
    int look = br.peek(9);
    uint16_t f = t.fast[look];
    if (f) { br.skip(f & 0xFF); return f >> 8; }
    int code = br.peek(16);
    for (int l = 10; l <= 16; ++l) {
        int c = code >> (16 - l);
        if (t.maxcode[l] >= 0 && c <= t.maxcode[l] && c >= t.mincode[l]) { br.skip(l); return t.vals[t.valptr[l] + c - t.mincode[l]]; }
    }
    // codes of <= 9 bits that missed the fast table cannot exist; treat as corrupt
    br.skip(16);
    return -1;
}

inline int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                             15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct IdctTable {
    double c[8][8];   // c[x][u] = 0.5 * C(u) * cos((2x+1) u pi / 16)
    IdctTable()
    {
        for (int x = 0; x < 8; ++x)
            for (int u = 0; u < 8; ++u)
                c[x][u] = 0.5 * (u == 0 ? std::sqrt(0.5) : 1.0) * std::cos((2 * x + 1) * u * M_PI / 16.0);
    }
};

void idct_block(const int *coef, uint8_t *out, size_t stride)
{
    static const IdctTable T;
    double tmp[64];
    for (int u = 0; u < 8; ++u)          // columns
        for (int y = 0; y < 8; ++y) {
            double s = 0.0;
            for (int v = 0; v < 8; ++v) if (coef[v * 8 + u]) s += T.c[y][v] * coef[v * 8 + u];
            tmp[y * 8 + u] = s;
        }
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            double s = 0.0;
            for (int u = 0; u < 8; ++u) s += T.c[x][u] * tmp[y * 8 + u];
            int v = (int)std::floor(s + 128.5);
            out[y * stride + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
}

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int pred = 0;
    size_t stride = 0, rows = 0;      // padded plane
    uint32_t cw = 0, ch = 0;          // real (downsampled) size
    std::vector<uint8_t> plane;
};

// libjpeg "fancy" (triangle filter) upsampling of one component plane to full resolution
void upsample(const Component &c, int hmax, int vmax, uint32_t W, uint32_t H, std::vector<uint8_t> &out)
{
    out.resize((size_t)W * H);
    int hs = hmax / c.h, vs = vmax / c.v;
    auto px = [&](int x, int y) -> int {
        x = x < 0 ? 0 : (x >= (int)c.cw ? (int)c.cw - 1 : x);
        y = y < 0 ? 0 : (y >= (int)c.ch ? (int)c.ch - 1 : y);
        return c.plane[(size_t)y * c.stride + x];
    };
    if (hs == 1 && vs == 1) {
        for (uint32_t y = 0; y < H; ++y) memcpy(&out[(size_t)y * W], &c.plane[(size_t)y * c.stride], W);
    } else if (hs == 2 && vs == 1) {
        for (uint32_t y = 0; y < H; ++y)
            for (uint32_t x = 0; x < W; ++x) {
                int i = (int)(x >> 1);
                int v;
                if (x & 1) v = (i + 1 >= (int)c.cw) ? px(i, y) : (px(i, y) * 3 + px(i + 1, y) + 2) >> 2;
                else v = (i == 0) ? px(i, y) : (px(i, y) * 3 + px(i - 1, y) + 1) >> 2;
                out[(size_t)y * W + x] = (uint8_t)v;
            }
    } else if (hs == 2 && vs == 2) {
        for (uint32_t y = 0; y < H; ++y) {
            int iy = (int)(y >> 1);
            int ny = (y & 1) ? iy + 1 : iy - 1;          // next-nearest input row
            for (uint32_t x = 0; x < W; ++x) {
                int i = (int)(x >> 1);
                int cur = px(i, iy) * 3 + px(i, ny);
                int v;
                if (x & 1) {
                    if (i + 1 >= (int)c.cw) v = (cur * 4 + 7) >> 4;
                    else v = (cur * 3 + (px(i + 1, iy) * 3 + px(i + 1, ny)) + 7) >> 4;
                } else {
                    if (i == 0) v = (cur * 4 + 8) >> 4;
                    else v = (cur * 3 + (px(i - 1, iy) * 3 + px(i - 1, ny)) + 8) >> 4;
                }
                out[(size_t)y * W + x] = (uint8_t)v;
            }
        }
    } else {
        for (uint32_t y = 0; y < H; ++y)      // other ratios: replicate
            for (uint32_t x = 0; x < W; ++x) out[(size_t)y * W + x] = (uint8_t)px((int)(x / hs), (int)(y / vs));
    }
}

bool decode_jpeg(const std::vector<uint8_t> &d, Image &img, std::string &err)
{
    if (d.size() < 4 || d[0] != 0xFF || d[1] != 0xD8) { err = "not a JPEG"; return false; }
    uint16_t qt[4][64]; bool qt_ok[4] = {false, false, false, false};
    HuffTable hdc[4], hac[4];
    Component comp[3]; int ncomp = 0;
    uint32_t W = 0, H = 0; int hmax = 1, vmax = 1;
    int restart = 0; bool have_sof = false;
    int adobe_transform = -1;
    size_t p = 2;
    while (p + 4 <= d.size()) {
        if (d[p] != 0xFF) { ++p; continue; }
        int m = d[p + 1];
        if (m == 0xFF) { ++p; continue; }
        p += 2;
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (p + 2 > d.size()) break;
        size_t len = be16(&d[p]);
        if (len < 2 || p + len > d.size()) { err = "JPEG: truncated segment"; return false; }
        const uint8_t *s = &d[p + 2]; size_t n = len - 2;
        if (m == 0xDB) {
            size_t i = 0;
            while (i < n) {
                int pq = s[i] >> 4, tq = s[i] & 15; ++i;
                if (tq > 3 || i + (pq ? 128 : 64) > n) { err = "JPEG: bad DQT"; return false; }
                for (int k = 0; k < 64; ++k) {
                    qt[tq][kZigzag[k]] = pq ? (uint16_t)be16(&s[i + 2 * k]) : s[i + k];
                }
                i += pq ? 128 : 64; qt_ok[tq] = true;
            }
        } else if (m == 0xC4) {
            size_t i = 0;
            while (i + 17 <= n) {
                int tc = s[i] >> 4, th = s[i] & 15; ++i;
                if (th > 3 || tc > 1) { err = "JPEG: bad DHT"; return false; }
                HuffTable &t = tc ? hac[th] : hdc[th];
                int total = 0;
                for (int l = 1; l <= 16; ++l) { t.bits[l] = s[i + l - 1]; total += t.bits[l]; }
                i += 16;
                if (total > 256 || i + total > n) { err = "JPEG: bad DHT"; return false; }
                memcpy(t.vals, &s[i], total); i += total;
                if (!t.build()) { t.present = false; err = "JPEG: bad DHT"; return false; }
                t.present = true;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (n < 6) { err = "JPEG: bad SOF"; return false; }
            if (s[0] != 8) { err = "JPEG: only 8-bit precision is supported"; return false; }
            H = be16(&s[1]); W = be16(&s[3]); ncomp = s[5];
            if ((ncomp != 1 && ncomp != 3) || n < 6 + 3 * (size_t)ncomp) { err = "JPEG: unsupported component count"; return false; }
            for (int c = 0; c < ncomp; ++c) {
                comp[c].id = s[6 + 3 * c]; comp[c].h = s[7 + 3 * c] >> 4; comp[c].v = s[7 + 3 * c] & 15; comp[c].tq = s[8 + 3 * c];
                if (comp[c].h < 1 || comp[c].h > 4 || comp[c].v < 1 || comp[c].v > 4 || comp[c].tq > 3) { err = "JPEG: bad SOF"; return false; }
                hmax = std::max(hmax, comp[c].h); vmax = std::max(vmax, comp[c].v);
            }
            have_sof = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            err = "JPEG: progressive / lossless / arithmetic JPEG is not supported"; return false;
        } else if (m == 0xDD) {
            if (n >= 2) restart = (int)be16(s);
        } else if (m == 0xEE) {
            if (n >= 12 && !memcmp(s, "Adobe", 5)) adobe_transform = s[11];
        } else if (m == 0xDA) {
            if (!have_sof || W == 0 || H == 0) { err = "JPEG: SOS before SOF"; return false; }
            if (n < 1) { err = "JPEG: bad SOS"; return false; }
            int ns = s[0];
            if (ns != ncomp || n < 1 + 2 * (size_t)ns + 3) { err = "JPEG: non-interleaved multi-scan files are not supported"; return false; }
            for (int k = 0; k < ns; ++k) {
                int cid = s[1 + 2 * k], tb = s[2 + 2 * k];
                int c = -1;
                for (int j = 0; j < ncomp; ++j) if (comp[j].id == cid) c = j;
                if (c < 0) { err = "JPEG: bad SOS"; return false; }
                comp[c].td = tb >> 4; comp[c].ta = tb & 15;
                if (comp[c].td > 3 || comp[c].ta > 3 || !hdc[comp[c].td].present || !hac[comp[c].ta].present || !qt_ok[comp[c].tq]) { err = "JPEG: missing table"; return false; }
            }
            // entropy-coded data
            uint32_t mcu_w = 8 * hmax, mcu_h = 8 * vmax;
            uint32_t mcux = (W + mcu_w - 1) / mcu_w, mcuy = (H + mcu_h - 1) / mcu_h;
            for (int c = 0; c < ncomp; ++c) {
                comp[c].stride = (size_t)mcux * comp[c].h * 8; comp[c].rows = (size_t)mcuy * comp[c].v * 8;
                comp[c].plane.assign(comp[c].stride * comp[c].rows, 0);
                comp[c].cw = (W * comp[c].h + hmax - 1) / hmax; comp[c].ch = (H * comp[c].v + vmax - 1) / vmax;
                comp[c].pred = 0;
            }
            BitReader br; br.p = &d[p + len]; br.end = d.data() + d.size();
            int until_restart = restart;
            int coef[64];
            for (uint32_t my = 0; my < mcuy; ++my)
                for (uint32_t mx = 0; mx < mcux; ++mx) {
                    if (restart && until_restart == 0) {
                        // drop the padding bits and step over the RSTn marker
                        br.reset();
                        while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) ++br.p;
                        if (br.p + 1 < br.end) br.p += 2;
                        for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
                        until_restart = restart;
                    }
                    for (int c = 0; c < ncomp; ++c) {
                        Component &cc = comp[c];
                        const HuffTable &dc = hdc[cc.td], &ac = hac[cc.ta];
                        const uint16_t *q = qt[cc.tq];
                        for (int by = 0; by < cc.v; ++by)
                            for (int bx = 0; bx < cc.h; ++bx) {
                                memset(coef, 0, sizeof coef);
                                int t = huff_decode(br, dc);
                                if (t < 0 || t > 11) { err = "JPEG: corrupt DC code"; return false; }
                                int diff = t ? extend(br.get(t), t) : 0;
                                cc.pred += diff;
                                coef[0] = cc.pred * q[0];
                                for (int k = 1; k < 64;) {
                                    int rs = huff_decode(br, ac);
                                    if (rs < 0) { err = "JPEG: corrupt AC code"; return false; }
                                    int r = rs >> 4, sz = rs & 15;
                                    if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                                    k += r;
                                    if (k > 63) { err = "JPEG: corrupt AC run"; return false; }
                                    coef[kZigzag[k]] = extend(br.get(sz), sz) * q[kZigzag[k]];
                                    ++k;
                                }
                                size_t ox = ((size_t)mx * cc.h + bx) * 8, oy = ((size_t)my * cc.v + by) * 8;
                                idct_block(coef, &cc.plane[oy * cc.stride + ox], cc.stride);
                            }
                    }
                    if (restart) --until_restart;
                }
            // colour conversion
            img.width = W; img.height = H; img.channels = 3;
            img.pixels.resize((size_t)W * H * 3);
            if (ncomp == 1) {
                err = "JPEG: grayscale textures are not supported (R8 format, graphics.rs:499)"; return false;
            }
            std::vector<uint8_t> full[3];
            for (int c = 0; c < 3; ++c) upsample(comp[c], hmax, vmax, W, H, full[c]);
            bool rgb = (adobe_transform == 0) || (comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B');
            for (size_t i = 0; i < (size_t)W * H; ++i) {
                uint8_t *o = &img.pixels[i * 3];
                if (rgb) { o[0] = full[0][i]; o[1] = full[1][i]; o[2] = full[2][i]; continue; }
                int y = full[0][i], cb = full[1][i] - 128, cr = full[2][i] - 128;
                // JFIF YCbCr -> RGB with libjpeg's 16-bit fixed-point tables
                int r = y + ((91881 * cr + 32768) >> 16);
                int g = y + ((-22554 * cb - 46802 * cr + 32768) >> 16);
                int b = y + ((116130 * cb + 32768) >> 16);
                o[0] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
                o[1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
                o[2] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
            }
            return true;
        }
        p += len;
    }
    err = "JPEG: no scan found";
    return false;
}

}  // namespace

bool decode_image_file(const std::string &path, Image &img, std::string &err)
{
    std::vector<uint8_t> d;
    if (!read_file(path, d, err)) return false;
    if (d.size() >= 8 && d[0] == 0x89 && d[1] == 'P') return decode_png(d, img, err);
    if (d.size() >= 2 && d[0] == 0xFF && d[1] == 0xD8) return decode_jpeg(d, img, err);
    err = "unsupported image container: " + path;
    return false;
}

bool encode_png_rgb8(const std::string &path, uint32_t w, uint32_t h, const uint8_t *rgb, std::string &err)
{
    std::vector<uint8_t> raw(((size_t)w * 3 + 1) * h);
    for (uint32_t y = 0; y < h; ++y) {
        raw[((size_t)w * 3 + 1) * y] = 0;
        memcpy(&raw[((size_t)w * 3 + 1) * y + 1], rgb + (size_t)w * 3 * y, (size_t)w * 3);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { err = "PNG: deflate failed"; return false; }
    comp.resize(clen);
    std::vector<uint8_t> out;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    out.insert(out.end(), sig, sig + 8);
    auto chunk = [&](const char *type, const uint8_t *data, size_t n) {
        uint8_t hdr[8] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n,
                          (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
        out.insert(out.end(), hdr, hdr + 8);
        out.insert(out.end(), data, data + n);
        uLong crc = crc32(0L, hdr + 4, 4);
        if (n) crc = crc32(crc, data, (uInt)n);
        uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
        out.insert(out.end(), c, c + 4);
    };
    uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                        (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 2, 0, 0, 0};
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), comp.size());
    chunk("IEND", nullptr, 0);
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create " + path; return false; }
    size_t wr = fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    if (wr != out.size()) { err = "short write to " + path; return false; }
    return true;
}

}  // namespace pth

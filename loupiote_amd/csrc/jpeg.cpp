// jpeg.cpp — dependency-free baseline JPEG decoder for glTF images (DamagedHelmet / Sponza carry JPEG
// textures; the reference decodes them through the `gltf` crate's `image` import,
// crates/lib/src/loaders/gltf.rs:12-44,150-153).  Sequential DCT, Huffman, 8-bit, 1 or 3 components,
// sampling factors up to 2x2, restart intervals.  Progressive / arithmetic / 12-bit streams are rejected
// (the loader then reports Error::FileNotFound like any other undecodable image).
// Output: RGBA8 with alpha 0 for the missing channel, exactly like the RGB -> RGBA expansion of gltf.rs:26-38.
// Chroma is upsampled by replication and the IDCT is a separable float transform, so pixels can differ
// from libjpeg's (fancy upsampling, integer IDCT) by a few code values; tests bound the difference.
#include <cmath>

#include "common.h"

namespace lpt {
namespace {

struct JHuff {
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    // a scan may name a table no DHT segment defined: such a table decodes nothing (every code length is "absent")
    JHuff() { for (int l = 0; l < 17; ++l) { mincode[l] = 0; maxcode[l] = -1; valptr[l] = 0; } maxcode[17] = 0x7FFFFFFF; }
    void prepare() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7FFFFFFF;
    }
};

struct Comp { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; int bw = 0, bh = 0; std::vector<uint8_t> plane; };

struct Reader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int cnt = 0;
    bool marker = false, ok = true;
    int bit() {
        if (cnt == 0) {
            if (p >= end) { ok = false; return 0; }
            uint8_t b = *p++;
            if (b == 0xFF) {
                if (p < end && *p == 0x00) ++p;
                else { marker = true; --p; return 0; }  // a marker inside entropy data: feed zeros
            }
            acc = b;
            cnt = 8;
        }
        return (int)((acc >> --cnt) & 1u);
    }
    int bits(int n) { int v = 0; while (n--) v = (v << 1) | bit(); return v; }
    void reset() { cnt = 0; marker = false; }
};

int decode_sym(Reader &r, const JHuff &h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | r.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    r.ok = false;
    return 0;
}
int extend(int v, int n) { return n && v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

const uint8_t kZig[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

void idct8x8(const float *in, uint8_t *out, int stride) {
    static float c[8][8];
    static bool init = false;
    if (!init) {
        for (int x = 0; x < 8; ++x)
            for (int u = 0; u < 8; ++u) c[x][u] = (u == 0 ? 0.35355339059f : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
        init = true;
    }
    float tmp[64];
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) {
            float s = 0.f;
            for (int u = 0; u < 8; ++u) s += c[x][u] * in[y * 8 + u];
            tmp[y * 8 + x] = s;
        }
    for (int x = 0; x < 8; ++x)
        for (int y = 0; y < 8; ++y) {
            float s = 0.f;
            for (int v = 0; v < 8; ++v) s += c[y][v] * tmp[v * 8 + x];
            int q = (int)std::floor(s + 128.5f);
            out[y * stride + x] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
        }
}

uint16_t be16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }

}  // namespace

bool decode_jpeg(const uint8_t *data, size_t size, Image &out) {
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return false;
    uint16_t qt[4][64] = {{0}};
    JHuff hdc[4], hac[4];
    Comp comp[3];
    int ncomp = 0, W = 0, H = 0, hmax = 1, vmax = 1, restart = 0;
    size_t off = 2;
    bool have_frame = false;
    while (off + 4 <= size) {
        if (data[off] != 0xFF) { ++off; continue; }
        const uint8_t m = data[off + 1];
        off += 2;
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0xFF) { if (m == 0xFF) --off; continue; }
        if (m == 0xD9) break;
        if (off + 2 > size) return false;
        const size_t len = be16(data + off);
        if (len < 2 || off + len > size) return false;
        const uint8_t *seg = data + off + 2;
        const size_t n = len - 2;
        if (m == 0xDB) {  // DQT
            size_t i = 0;
            while (i < n) {
                const int pq = seg[i] >> 4, tq = seg[i] & 15;
                ++i;
                if (tq > 3 || i + (pq ? 128u : 64u) > n) return false;
                for (int k = 0; k < 64; ++k) { qt[tq][kZig[k]] = pq ? be16(seg + i + 2 * k) : seg[i + k]; }
                i += pq ? 128 : 64;
            }
        } else if (m == 0xC4) {  // DHT
            size_t i = 0;
            while (i + 17 <= n) {
                const int tc = seg[i] >> 4, th = seg[i] & 15;
                if (th > 3 || tc > 1) return false;
                JHuff &h = tc ? hac[th] : hdc[th];
                int total = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = seg[i + l]; total += h.bits[l]; }
                i += 17;
                if (total > 256 || i + (size_t)total > n) return false;
                memcpy(h.vals, seg + i, (size_t)total);
                i += (size_t)total;
                h.prepare();
            }
        } else if (m == 0xC0 || m == 0xC1) {  // baseline / extended sequential, Huffman
            if (n < 6 || seg[0] != 8) return false;
            H = be16(seg + 1); W = be16(seg + 3); ncomp = seg[5];
            if ((ncomp != 1 && ncomp != 3) || !W || !H || n < 6u + 3u * (size_t)ncomp) return false;
            for (int c = 0; c < ncomp; ++c) {
                comp[c].id = seg[6 + 3 * c]; comp[c].h = seg[7 + 3 * c] >> 4; comp[c].v = seg[7 + 3 * c] & 15; comp[c].tq = seg[8 + 3 * c];
                if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3) return false;
                hmax = std::max(hmax, comp[c].h); vmax = std::max(vmax, comp[c].v);
            }
            have_frame = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return false;  // progressive, lossless, arithmetic: not supported
        } else if (m == 0xDD) {
            if (n < 2) return false;
            restart = be16(seg);
        } else if (m == 0xDA) {  // SOS: the one scan of a sequential image
            if (!have_frame || n < 1 || seg[0] != ncomp || n < 1u + 2u * (size_t)ncomp + 3u) return false;
            for (int k = 0; k < ncomp; ++k) {
                const int id = seg[1 + 2 * k];
                int c = -1;
                for (int j = 0; j < ncomp; ++j) if (comp[j].id == id) c = j;
                if (c < 0) return false;
                comp[c].td = seg[2 + 2 * k] >> 4; comp[c].ta = seg[2 + 2 * k] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3) return false;
            }
            const int mcuw = 8 * hmax, mcuh = 8 * vmax;
            const int mx = (W + mcuw - 1) / mcuw, my = (H + mcuh - 1) / mcuh;
            // untrusted header: every coded 8x8 block takes at least two bits of the scan (a DC and an EOB code), so a
            // frame with more blocks than that cannot be in this file
            if ((size_t)mx * (size_t)my > 4u * (size - (off + len)) + 16u) return false;
            for (int c = 0; c < ncomp; ++c) {
                comp[c].bw = mx * comp[c].h * 8; comp[c].bh = my * comp[c].v * 8;
                comp[c].plane.assign((size_t)comp[c].bw * comp[c].bh, 0);
                comp[c].pred = 0;
            }
            Reader r{data + off + len, data + size};
            int count = 0;
            for (int my_ = 0; my_ < my; ++my_)
                for (int mx_ = 0; mx_ < mx; ++mx_) {
                    if (restart && count && count % restart == 0) {
                        r.reset();  // skip to past the RSTn marker
                        while (r.p + 1 < r.end && !(r.p[0] == 0xFF && r.p[1] >= 0xD0 && r.p[1] <= 0xD7)) ++r.p;
                        if (r.p + 1 < r.end) r.p += 2;
                        for (int c = 0; c < ncomp; ++c) comp[c].pred = 0;
                    }
                    ++count;
                    for (int c = 0; c < ncomp; ++c)
                        for (int by = 0; by < comp[c].v; ++by)
                            for (int bx = 0; bx < comp[c].h; ++bx) {
                                float blk[64] = {0};
                                const int t = decode_sym(r, hdc[comp[c].td]);
                                if (t > 11) return false;
                                comp[c].pred += extend(r.bits(t), t);
                                blk[0] = (float)comp[c].pred * (float)qt[comp[c].tq][0];
                                for (int k = 1; k < 64;) {
                                    const int rs = decode_sym(r, hac[comp[c].ta]);
                                    const int run = rs >> 4, sz = rs & 15;
                                    if (sz == 0) { if (run == 15) { k += 16; continue; } break; }
                                    k += run;
                                    if (k > 63) return false;
                                    blk[kZig[k]] = (float)extend(r.bits(sz), sz) * (float)qt[comp[c].tq][kZig[k]];
                                    ++k;
                                }
                                if (!r.ok) return false;
                                const int px = (mx_ * comp[c].h + bx) * 8, py = (my_ * comp[c].v + by) * 8;
                                idct8x8(blk, &comp[c].plane[(size_t)py * comp[c].bw + px], comp[c].bw);
                            }
                }
            out.width = (uint32_t)W; out.height = (uint32_t)H;
            out.rgba8.assign((size_t)W * H * 4, 0);
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    uint8_t *dst = &out.rgba8[((size_t)y * W + x) * 4];
                    const float Y = comp[0].plane[(size_t)(y * comp[0].v / vmax) * comp[0].bw + (x * comp[0].h / hmax)];
                    if (ncomp == 1) { dst[0] = (uint8_t)Y; continue; }  // grey: one channel, the rest stay 0 (gltf.rs:26-38)
                    const float cb = comp[1].plane[(size_t)(y * comp[1].v / vmax) * comp[1].bw + (x * comp[1].h / hmax)] - 128.f;
                    const float cr = comp[2].plane[(size_t)(y * comp[2].v / vmax) * comp[2].bw + (x * comp[2].h / hmax)] - 128.f;
                    const float rgb[3] = {Y + 1.402f * cr, Y - 0.344136f * cb - 0.714136f * cr, Y + 1.772f * cb};
                    for (int k = 0; k < 3; ++k) { const int q = (int)std::floor(rgb[k] + 0.5f); dst[k] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q)); }
                }
            return true;
        }
        off += len;
    }
    return false;
}

}  // namespace lpt

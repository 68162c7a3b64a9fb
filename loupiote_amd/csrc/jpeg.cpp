// jpeg.cpp — dependency-free JPEG decoder for glTF images (DamagedHelmet / Sponza carry JPEG
// textures; the reference decodes them through the `gltf` crate's `image` import,
// crates/lib/src/loaders/gltf.rs:12-44,150-153).  Sequential (SOF0 / SOF1) and progressive (SOF2: spectral selection and
// successive approximation, coefficients kept for the whole frame and transformed after the last scan) DCT, Huffman, 8-bit,
// 1 or 3 components, sampling factors up to 2x2, restart intervals.  Arithmetic-coded / lossless / 12-bit streams are
// rejected (the loader then reports Error::FileNotFound like any other undecodable image).
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

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
    int bw = 0, bh = 0;              // plane size in pixels (whole MCUs)
    std::vector<uint8_t> plane;
    // progressive only: the quantised coefficients of every block (natural order), refined scan by scan
    std::vector<int16_t> coef;
    int nbx = 0, nby = 0;            // blocks a non-interleaved scan of this component covers
};

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

// ---- progressive mode (SOF2; ITU T.81 annex G): one block of one scan.  `c` = the block's 64 coefficients (natural order).
struct ProgScan { int ss, se, ah, al; int eobrun; };

bool prog_dc(Reader &r, const JHuff &h, Comp &cp, int16_t *c, const ProgScan &sc) {
    if (sc.ah == 0) {
        const int t = decode_sym(r, h);
        if (t > 11) return false;
        cp.pred += t ? extend(r.bits(t), t) : 0;
        c[0] = (int16_t)(cp.pred * (1 << sc.al));
    } else if (r.bit()) {
        c[0] = (int16_t)(c[0] | (1 << sc.al));
    }
    return r.ok;
}

bool prog_ac(Reader &r, const JHuff &h, int16_t *c, ProgScan &sc) {
    if (sc.ah == 0) {                                   // first pass over this band
        if (sc.eobrun) { --sc.eobrun; return true; }
        for (int k = sc.ss; k <= sc.se;) {
            const int rs = decode_sym(r, h), run = rs >> 4, sz = rs & 15;
            if (!r.ok) return false;
            if (sz == 0) {
                if (run < 15) { sc.eobrun = (1 << run) - 1; if (run) sc.eobrun += r.bits(run); break; }
                k += 16;
            } else {
                k += run;
                if (k > sc.se) return false;
                c[kZig[k]] = (int16_t)(extend(r.bits(sz), sz) * (1 << sc.al));
                ++k;
            }
        }
        return r.ok;
    }
    // refinement: one more bit for every coefficient that is already non-zero, new +-1 coefficients in between
    const int p1 = 1 << sc.al, m1 = -(1 << sc.al);
    int k = sc.ss;
    auto refine = [&](int16_t &v) {
        if (r.bit() && (v & p1) == 0) v = (int16_t)(v + (v >= 0 ? p1 : m1));
    };
    if (sc.eobrun == 0) {
        bool eob = false;
        while (k <= sc.se) {
            const int rs = decode_sym(r, h);
            if (!r.ok) return false;
            int run = rs >> 4;
            const int sz = rs & 15;
            int val = 0;
            if (sz == 0) {
                if (run < 15) { sc.eobrun = (1 << run) - 1; if (run) sc.eobrun += r.bits(run); eob = true; break; }   // end of band: the rest below
            } else {
                if (sz != 1) return false;
                val = r.bit() ? p1 : m1;
            }
            while (k <= sc.se) {
                int16_t &v = c[kZig[k]];
                ++k;
                if (v != 0) { refine(v); continue; }
                if (run == 0) { if (sz) v = (int16_t)val; break; }   // run == 0 with sz == 0 only happens for ZRL's 16th zero
                --run;
            }
        }
        if (!eob) return r.ok;
        // an EOB (run) starts in this block: its remaining non-zero coefficients are still refined
        for (; k <= sc.se; ++k) { int16_t &v = c[kZig[k]]; if (v != 0) refine(v); }
        return r.ok;   // eobrun counts the FOLLOWING blocks
    }
    for (; k <= sc.se; ++k) { int16_t &v = c[kZig[k]]; if (v != 0) refine(v); }
    --sc.eobrun;
    return r.ok;
}

// planes (one byte per sample, whole MCUs) -> RGBA8: replicated chroma, BT.601 full-range YCbCr (SPEC §14.5)
void planes_to_rgba(const Comp *comp, int ncomp, int W, int H, int hmax, int vmax, Image &out) {
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
}

void skip_to_after_rst(Reader &r) {
    r.reset();  // skip to past the RSTn marker
    while (r.p + 1 < r.end && !(r.p[0] == 0xFF && r.p[1] >= 0xD0 && r.p[1] <= 0xD7)) ++r.p;
    if (r.p + 1 < r.end) r.p += 2;
}

}  // namespace

bool decode_jpeg(const uint8_t *data, size_t size, Image &out) {
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return false;
    uint16_t qt[4][64] = {{0}};
    JHuff hdc[4], hac[4];
    Comp comp[3];
    int ncomp = 0, W = 0, H = 0, hmax = 1, vmax = 1, restart = 0;
    size_t off = 2;
    bool have_frame = false, progressive = false, have_scan = false;
    int mx = 0, my = 0;   // MCUs per row / column
    int n_scans = 0;
    // progressive: all scans have been read (EOI, or the data ended): dequantise, inverse transform, convert
    auto finish_progressive = [&]() -> bool {
        if (!progressive || !have_scan) return false;
        for (int c = 0; c < ncomp; ++c) {
            Comp &cp = comp[c];
            cp.plane.assign((size_t)cp.bw * cp.bh, 0);
            const int bpr = cp.bw / 8;
            for (int by = 0; by < cp.bh / 8; ++by)
                for (int bx = 0; bx < bpr; ++bx) {
                    const int16_t *cf = &cp.coef[((size_t)by * bpr + bx) * 64];
                    float blk[64];
                    for (int k = 0; k < 64; ++k) blk[k] = (float)cf[k] * (float)qt[cp.tq][k];
                    idct8x8(blk, &cp.plane[(size_t)by * 8 * cp.bw + (size_t)bx * 8], cp.bw);
                }
        }
        planes_to_rgba(comp, ncomp, W, H, hmax, vmax, out);
        return true;
    };
    while (off + 4 <= size) {
        if (data[off] != 0xFF) { ++off; continue; }
        const uint8_t m = data[off + 1];
        off += 2;
        if (m == 0xD8 || m == 0x01 || m == 0x00 || (m >= 0xD0 && m <= 0xD7) || m == 0xFF) { if (m == 0xFF) --off; continue; }   // 0x00: a stuffed byte of entropy data
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
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {  // baseline / extended sequential / progressive, Huffman
            if (have_frame || n < 6 || seg[0] != 8) return false;
            progressive = m == 0xC2;
            H = be16(seg + 1); W = be16(seg + 3); ncomp = seg[5];
            if ((ncomp != 1 && ncomp != 3) || !W || !H || n < 6u + 3u * (size_t)ncomp) return false;
            // untrusted header: an absolute bound on what a file may make the decoder allocate (the renderer's own limit is
            // 8192 x 8192): 67 M pixels = 268 MB of RGBA8, 200 MB of progressive coefficients
            if ((size_t)W * (size_t)H > (size_t)8192 * 8192) return false;
            for (int c = 0; c < ncomp; ++c) {
                comp[c].id = seg[6 + 3 * c]; comp[c].h = seg[7 + 3 * c] >> 4; comp[c].v = seg[7 + 3 * c] & 15; comp[c].tq = seg[8 + 3 * c];
                if (comp[c].h < 1 || comp[c].h > 2 || comp[c].v < 1 || comp[c].v > 2 || comp[c].tq > 3) return false;
                hmax = std::max(hmax, comp[c].h); vmax = std::max(vmax, comp[c].v);
            }
            have_frame = true;
            mx = (W + 8 * hmax - 1) / (8 * hmax); my = (H + 8 * vmax - 1) / (8 * vmax);
            if (progressive) {
                // untrusted header: the first DC scan spends at least one bit on every block of the frame
                if ((size_t)mx * (size_t)my > 8u * size + 16u) return false;
                for (int c = 0; c < ncomp; ++c) {
                    Comp &cp = comp[c];
                    cp.bw = mx * cp.h * 8; cp.bh = my * cp.v * 8;
                    cp.coef.assign((size_t)(cp.bw / 8) * (cp.bh / 8) * 64, 0);
                    cp.nbx = ((W * cp.h + hmax - 1) / hmax + 7) / 8;
                    cp.nby = ((H * cp.v + vmax - 1) / vmax + 7) / 8;
                }
            }
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return false;  // lossless, hierarchical, arithmetic: not supported
        } else if (m == 0xDD) {
            if (n < 2) return false;
            restart = be16(seg);
        } else if (m == 0xDA && progressive) {  // SOS of a progressive image: one of several scans
            if (!have_frame || n < 1 || ++n_scans > 256) return false;   // real files carry ~10 scans; each one costs a pass over the frame
            const int ns = seg[0];
            if (ns < 1 || ns > ncomp || n < 1u + 2u * (size_t)ns + 3u) return false;
            int which[3] = {0, 0, 0};
            for (int k = 0; k < ns; ++k) {
                const int id = seg[1 + 2 * k];
                int c = -1;
                for (int j = 0; j < ncomp; ++j) if (comp[j].id == id) c = j;
                if (c < 0) return false;
                for (int j = 0; j < k; ++j) if (which[j] == c) return false;
                which[k] = c;
                comp[c].td = seg[2 + 2 * k] >> 4; comp[c].ta = seg[2 + 2 * k] & 15;
                if (comp[c].td > 3 || comp[c].ta > 3) return false;
            }
            ProgScan sc{seg[1 + 2 * ns], seg[2 + 2 * ns], seg[3 + 2 * ns] >> 4, seg[3 + 2 * ns] & 15, 0};
            if (sc.ss > 63 || sc.se > 63 || sc.ss > sc.se || sc.al > 13 || sc.ah > 13) return false;
            if ((sc.ss == 0) != (sc.se == 0)) return false;          // a scan is either DC only or AC only
            if (sc.ss > 0 && ns != 1) return false;                  // AC scans carry one component
            Reader r{data + off + len, data + size};
            for (int k = 0; k < ns; ++k) comp[which[k]].pred = 0;
            int count = 0;
            // false: the entropy-coded data ended (a marker other than the expected RSTn, or the file) before this block —
            // a truncated scan must fail here instead of walking every remaining block on zero bits
            auto restart_here = [&]() -> bool {
                const bool at_restart = restart && count && count % restart == 0;
                if ((r.marker || !r.ok) && !at_restart) return false;
                if (at_restart) {
                    skip_to_after_rst(r);
                    for (int k = 0; k < ns; ++k) comp[which[k]].pred = 0;
                    sc.eobrun = 0;
                }
                ++count;
                return true;
            };
            if (ns == 1) {   // non-interleaved: the component's own blocks, row by row
                Comp &cp = comp[which[0]];
                const int bpr = cp.bw / 8;
                for (int by = 0; by < cp.nby; ++by)
                    for (int bx = 0; bx < cp.nbx; ++bx) {
                        if (!restart_here()) return false;
                        int16_t *cf = &cp.coef[((size_t)by * bpr + bx) * 64];
                        const bool ok = sc.ss == 0 ? prog_dc(r, hdc[cp.td], cp, cf, sc) : prog_ac(r, hac[cp.ta], cf, sc);
                        if (!ok) return false;
                    }
            } else {         // interleaved (DC scans): MCU by MCU
                for (int my_ = 0; my_ < my; ++my_)
                    for (int mx_ = 0; mx_ < mx; ++mx_) {
                        if (!restart_here()) return false;
                        for (int k = 0; k < ns; ++k) {
                            Comp &cp = comp[which[k]];
                            const int bpr = cp.bw / 8;
                            for (int by = 0; by < cp.v; ++by)
                                for (int bx = 0; bx < cp.h; ++bx) {
                                    int16_t *cf = &cp.coef[((size_t)(my_ * cp.v + by) * bpr + (mx_ * cp.h + bx)) * 64];
                                    if (!prog_dc(r, hdc[cp.td], cp, cf, sc)) return false;
                                }
                        }
                    }
            }
            have_scan = true;
            off = (size_t)(r.p - data);   // the marker loop resumes behind the entropy-coded data of this scan
            continue;
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
                    const bool at_restart = restart && count && count % restart == 0;
                    if ((r.marker || !r.ok) && !at_restart) return false;   // the scan ended before its last MCU
                    if (at_restart) {
                        skip_to_after_rst(r);
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
            planes_to_rgba(comp, ncomp, W, H, hmax, vmax, out);
            return true;
        }
        off += len;
    }
    return finish_progressive();   // EOI (or the end of the data) after the scans of a progressive image
}

}  // namespace lpt

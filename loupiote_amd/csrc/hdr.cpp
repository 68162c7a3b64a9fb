// hdr.cpp — Radiance RGBE (.hdr / .pic) decoder for environment probes.
// Replaces the `image::codecs::hdr::HdrDecoder` step of ApplicationContext::load_env
// (reference crates/standalone/src/app.rs:138-155): the file's pixels are handed to ProbeGPU::new as they
// are stored — 4 bytes per pixel, shared exponent — so this decoder only undoes the run-length coding.
// Supported: "#?RADIANCE" / "#?RGBE" headers, FORMAT=32-bit_rle_rgbe, the standard "-Y H +X W" orientation,
// flat scanlines, old-style RLE runs and the new per-channel RLE (scanline widths 8..32767).
#include <algorithm>
#include <cmath>

#include "common.h"

namespace lpt {
namespace {

bool read_line(const uint8_t *d, size_t n, size_t &off, std::string &line) {
    line.clear();
    while (off < n) {
        const char c = (char)d[off++];
        if (c == '\n') return true;
        line.push_back(c);
        if (line.size() > 4096) return false;
    }
    return false;
}

}  // namespace

// out may be empty on failure; width/height are set on success
bool decode_hdr(const uint8_t *d, size_t n, uint32_t &width, uint32_t &height, std::vector<uint8_t> &rgbe) {
    size_t off = 0;
    std::string line;
    if (!read_line(d, n, off, line) || line.size() < 2 || line[0] != '#' || line[1] != '?') return false;
    bool fmt_ok = true;  // a missing FORMAT line means 32-bit_rle_rgbe
    for (;;) {
        if (!read_line(d, n, off, line)) return false;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) break;
        if (line.rfind("FORMAT=", 0) == 0) fmt_ok = line == "FORMAT=32-bit_rle_rgbe";
    }
    if (!fmt_ok) return false;
    if (!read_line(d, n, off, line)) return false;
    int H = 0, W = 0;
    if (sscanf(line.c_str(), "-Y %d +X %d", &H, &W) != 2 || H <= 0 || W <= 0 || (size_t)W * (size_t)H > (1u << 28)) return false;
    // untrusted header: a run-length code covers at most 127 pixels of one channel in two bytes, so the pixels cannot
    // outnumber the remaining bytes by more than 127 / 8
    if ((size_t)W * (size_t)H > 16u * (n - off) + 64u) return false;
    rgbe.assign((size_t)W * H * 4, 0);
    for (int y = 0; y < H; ++y) {
        uint8_t *row = &rgbe[(size_t)y * W * 4];
        if (off + 4 > n) return false;
        if (W >= 8 && W < 32768 && d[off] == 2 && d[off + 1] == 2 && ((d[off + 2] << 8) | d[off + 3]) == W) {
            off += 4;  // new RLE: the four channels one after the other
            for (int ch = 0; ch < 4; ++ch) {
                int x = 0;
                while (x < W) {
                    if (off >= n) return false;
                    int count = d[off++];
                    if (count > 128) {
                        count -= 128;
                        if (count == 0 || x + count > W || off >= n) return false;
                        const uint8_t v = d[off++];
                        for (int k = 0; k < count; ++k) row[4 * (x++) + ch] = v;
                    } else {
                        if (count == 0 || x + count > W || off + (size_t)count > n) return false;
                        for (int k = 0; k < count; ++k) row[4 * (x++) + ch] = d[off++];
                    }
                }
            }
        } else {
            // flat pixels, with old-style runs: (1,1,1,n) repeats the previous pixel n << shift times
            int x = 0, shift = 0;
            while (x < W) {
                if (off + 4 > n) return false;
                const uint8_t *p = d + off;
                off += 4;
                if (p[0] == 1 && p[1] == 1 && p[2] == 1) {
                    if (x == 0) return false;
                    const long count = (long)p[3] << shift;
                    if (count <= 0 || x + count > W) return false;
                    for (long k = 0; k < count; ++k, ++x) memcpy(row + 4 * x, row + 4 * (x - 1), 4);
                    shift += 8;
                    if (shift > 16) return false;
                } else {
                    memcpy(row + 4 * x, p, 4);
                    ++x;
                    shift = 0;
                }
            }
        }
    }
    width = (uint32_t)W;
    height = (uint32_t)H;
    return true;
}

}  // namespace lpt

extern "C" int lpt_decode_hdr(const uint8_t *data, size_t size, uint8_t *rgbe8, size_t capacity, uint32_t *width, uint32_t *height) {
    if (!data || !width || !height) return lpt::fail(LPT_ERR_INVALID_ARG, "lpt_decode_hdr: null");
    std::vector<uint8_t> px;
    uint32_t w = 0, h = 0;
    if (!lpt::decode_hdr(data, size, w, h, px)) return lpt::fail(LPT_ERR_FILE_NOT_FOUND, "file not found: not a decodable Radiance RGBE image");
    *width = w;
    *height = h;
    if (rgbe8) {
        if (capacity < px.size()) return lpt::fail(LPT_ERR_INVALID_ARG, "lpt_decode_hdr: buffer of %zu bytes, %zu needed", capacity, px.size());
        memcpy(rgbe8, px.data(), px.size());
    }
    return LPT_OK;
}

// Radiance writer for linear radiance (the "EXR / HDR" half of SURVEY §8f-4; the reference only saves the tonemapped
// PNG, app.rs:172-187): RGB floats -> shared-exponent RGBE (mantissas truncated as in Ward's float2rgbe), flat scanlines.
extern "C" int lpt_write_hdr(const char *path, const float *rgba, uint32_t width, uint32_t height, size_t row_floats) {
    if (!path || !rgba || !width || !height || row_floats < (size_t)width * 4) return lpt::fail(LPT_ERR_INVALID_ARG, "lpt_write_hdr: bad arguments");
    FILE *f = fopen(path, "wb");
    if (!f) return lpt::fail(LPT_ERR_FILE_NOT_FOUND, "file not found: cannot open %s for writing", path);
    fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %u +X %u\n", height, width);
    std::vector<uint8_t> row((size_t)width * 4);
    bool ok = true;
    for (uint32_t y = 0; y < height && ok; ++y) {
        const float *src = rgba + (size_t)y * row_floats;
        for (uint32_t x = 0; x < width; ++x) {
            float r = src[4 * x], g = src[4 * x + 1], b = src[4 * x + 2];
            if (!(r > 0.f)) r = 0.f;   // negative and NaN -> 0
            if (!(g > 0.f)) g = 0.f;
            if (!(b > 0.f)) b = 0.f;
            const float m = std::max(r, std::max(g, b));
            uint8_t *p = &row[4 * (size_t)x];
            if (!(m >= 1e-32f) || !std::isfinite(m)) { p[0] = p[1] = p[2] = p[3] = 0; if (std::isfinite(m)) continue; }
            if (!std::isfinite(m)) { p[0] = p[1] = p[2] = 255; p[3] = 255; continue; }
            int e;
            const float scale = std::frexp(m, &e) * 256.0f / m;   // m = f * 2^e, f in [0.5, 1)
            p[0] = (uint8_t)(r * scale); p[1] = (uint8_t)(g * scale); p[2] = (uint8_t)(b * scale);
            p[3] = (uint8_t)(e + 128);
        }
        ok = fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    fclose(f);
    return ok ? LPT_OK : lpt::fail(LPT_ERR_FILE_NOT_FOUND, "file not found: short write to %s", path);
}

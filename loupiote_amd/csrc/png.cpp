// png.cpp — dependency-free PNG decoder (zlib inflate + unfilter) for glTF images.
// The reference decodes images through the `gltf` crate's `image` import and then
// expands to RGBA8 leaving missing channels at 0 (crates/lib/src/loaders/gltf.rs:12-44);
// the same expansion rule is applied here.  Non-interlaced, 8/16-bit, colour types 0/2/3/4/6.
#include <algorithm>
#include <cstdio>

#include "common.h"

namespace lpt {
namespace {

struct BitReader {
    const uint8_t *p, *end;
    uint32_t buf = 0;
    int cnt = 0;
    bool ok = true;
    uint32_t bits(int n) {
        while (cnt < n) {
            if (p >= end) { ok = false; return 0; }
            buf |= (uint32_t)(*p++) << cnt;
            cnt += 8;
        }
        uint32_t v = buf & ((n == 32) ? 0xFFFFFFFFu : ((1u << n) - 1u));
        buf >>= n;
        cnt -= n;
        return v;
    }
    void align() { buf = 0; cnt = 0; }
};

struct Huff {
    uint16_t count[16];
    uint16_t symbol[288];
    void build(const uint8_t *len, int n) {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; ++i) offs[i + 1] = (uint16_t)(offs[i] + count[i]);
        for (int i = 0; i < n; ++i)
            if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; ++len) {
            code |= (int)br.bits(1);
            if (!br.ok) return -1;
            int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

bool inflate(const uint8_t *src, size_t n, std::vector<uint8_t> &out) {
    if (n < 2) return false;
    BitReader br{src + 2, src + n};  // skip zlib header
    static const uint16_t lbase[] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    int last;
    do {
        last = (int)br.bits(1);
        int type = (int)br.bits(2);
        if (!br.ok) return false;
        if (type == 0) {
            br.align();
            if (br.end - br.p < 4) return false;
            uint32_t len = br.p[0] | (br.p[1] << 8);
            br.p += 4;
            if ((size_t)(br.end - br.p) < len) return false;
            out.insert(out.end(), br.p, br.p + len);
            br.p += len;
        } else if (type == 1 || type == 2) {
            Huff hl, hd;
            uint8_t lens[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; ++i) lens[i] = 8;
                for (; i < 256; ++i) lens[i] = 9;
                for (; i < 280; ++i) lens[i] = 7;
                for (; i < 288; ++i) lens[i] = 8;
                hl.build(lens, 288);
                for (i = 0; i < 30; ++i) lens[i] = 5;
                hd.build(lens, 30);
            } else {
                int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.bits(3);
                Huff hc;
                hc.build(cl, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = hc.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) lens[idx++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (!idx) return false; val = lens[idx - 1]; rep = 3 + (int)br.bits(2); }
                        else if (sym == 17) rep = 3 + (int)br.bits(3);
                        else rep = 11 + (int)br.bits(7);
                        if (idx + rep > nlen + ndist) return false;
                        while (rep--) lens[idx++] = (uint8_t)val;
                    }
                }
                hl.build(lens, nlen);
                hd.build(lens + nlen, ndist);
            }
            for (;;) {
                int sym = hl.decode(br);
                if (sym < 0 || !br.ok) return false;
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    sym -= 257;
                    if (sym >= 29) return false;
                    int len = lbase[sym] + (int)br.bits(lext[sym]);
                    int ds = hd.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    size_t dist = dbase[ds] + br.bits(dext[ds]);
                    if (dist > out.size()) return false;
                    size_t from = out.size() - dist;
                    for (int i = 0; i < len; ++i) out.push_back(out[from + i]);
                }
            }
        } else return false;
    } while (!last);
    return br.ok;
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

bool decode_png(const uint8_t *data, size_t size, Image &out) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (size < 8 || memcmp(data, sig, 8) != 0) return false;
    size_t off = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    while (off + 12 <= size) {
        uint32_t len = be32(data + off);
        const uint8_t *type = data + off + 4, *body = data + off + 8;
        if (off + 12 + (size_t)len > size) return false;
        if (!memcmp(type, "IHDR", 4) && len >= 13) { w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) break;
        off += 12 + (size_t)len;
    }
    if (!w || !h || interlace || (depth != 8 && depth != 16)) return false;
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch || (ctype == 3 && depth != 8)) return false;
    const size_t bpp = (size_t)ch * (depth / 8), stride = bpp * w;
    // untrusted header: deflate expands by at most 1032:1, so an image larger than that cannot be in this file
    if (w > (1u << 16) || h > (1u << 16) || (stride + 1) * (size_t)h > idat.size() * 1032u + 1024u) return false;
    std::vector<uint8_t> raw;
    raw.reserve((stride + 1) * h);
    if (!inflate(idat.data(), idat.size(), raw) || raw.size() < (stride + 1) * h) return false;
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t *in = &raw[(stride + 1) * y + 1];
        uint8_t *cur = &img[stride * y];
        const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; ++x) {
            int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = in[x];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: return false;
            }
            cur[x] = (uint8_t)v;
        }
    }
    out.width = w; out.height = h;
    out.rgba8.assign((size_t)w * h * 4, 0);  // gltf.rs:26-38: channels the source lacks stay 0
    const size_t step = depth / 8;
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const uint8_t *px = &img[i * bpp];
        uint8_t *dst = &out.rgba8[i * 4];
        if (ctype == 3) {
            size_t k = px[0];
            if (3 * k + 2 < plte.size()) { dst[0] = plte[3 * k]; dst[1] = plte[3 * k + 1]; dst[2] = plte[3 * k + 2]; }
        } else {
            for (int c = 0; c < ch; ++c) dst[c] = px[c * step];  // 16-bit: most significant byte
        }
    }
    return true;
}

}  // namespace lpt

// ---------------------------------------------------------------------------- PNG writer
// Headless stand-in for the app's `save_screenshot` (reference crates/standalone/src/app.rs:172-187:
// read_pixels -> image::ImageBuffer::save).  RGBA8, filter 0, zlib "stored" blocks (no compression).
namespace lpt {
namespace {
uint32_t crc_table[256];
bool crc_ready = false;
uint32_t crc32(uint32_t crc, const uint8_t *p, size_t n) {
    if (!crc_ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crc_table[i] = c;
        }
        crc_ready = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = crc_table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}
void put32(std::vector<uint8_t> &v, uint32_t x) { v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); }
void chunk(std::vector<uint8_t> &out, const char *type, const std::vector<uint8_t> &body) {
    put32(out, (uint32_t)body.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body.begin(), body.end());
    put32(out, crc32(0, &out[start], out.size() - start));
}
}  // namespace

bool encode_png(const uint8_t *rgba8, uint32_t w, uint32_t h, size_t row_bytes, std::vector<uint8_t> &out) {
    if (!rgba8 || !w || !h || row_bytes < (size_t)w * 4) return false;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    out.assign(sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put32(ihdr, w); put32(ihdr, h);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)w * 4 + 1) * h);
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgba8 + (size_t)y * row_bytes, rgba8 + (size_t)y * row_bytes + (size_t)w * 4);
    }
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (uint8_t c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
    size_t off = 0;
    while (off < raw.size()) {
        const size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + n));
        off += n;
    }
    put32(z, (b << 16) | a);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    return true;
}
}  // namespace lpt

extern "C" int lpt_write_png(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height, size_t row_bytes) {
    if (!path) return lpt::fail(LPT_ERR_INVALID_ARG, "lpt_write_png: null path");
    std::vector<uint8_t> png;
    if (!lpt::encode_png(rgba8, width, height, row_bytes, png)) return lpt::fail(LPT_ERR_INVALID_ARG, "lpt_write_png: bad image");
    FILE *f = fopen(path, "wb");
    if (!f) return lpt::fail(LPT_ERR_FILE_NOT_FOUND, "file not found: cannot open %s for writing", path);
    const bool ok = fwrite(png.data(), 1, png.size(), f) == png.size();
    fclose(f);
    return ok ? LPT_OK : lpt::fail(LPT_ERR_FILE_NOT_FOUND, "file not found: short write to %s", path);
}

// replaces: the `image::io::Reader::open(path).decode()` half of ApplicationContext::load_blue_noise
// (crates/standalone/src/app.rs:116-132): PNG / JPEG bytes -> RGBA8 pixels for lpt_renderer_upload_noise (or any other use).
extern "C" int lpt_decode_image(const uint8_t *data, size_t size, uint8_t *rgba8, size_t capacity, uint32_t *width, uint32_t *height) {
    if (!data || !width || !height) return lpt::fail(LPT_ERR_INVALID_ARG, "lpt_decode_image: null");
    lpt::Image im;
    bool ok = false;
    try {
        ok = lpt::decode_png(data, size, im) || lpt::decode_jpeg(data, size, im);
    } catch (const std::exception &e) {
        return lpt::fail(LPT_ERR_FILE_NOT_FOUND, "file not found: %s", e.what());
    }
    if (!ok) return lpt::fail(LPT_ERR_FILE_NOT_FOUND, "file not found: neither a decodable PNG nor a Huffman-coded JPEG");
    *width = im.width; *height = im.height;
    if (rgba8) {
        if (capacity < im.rgba8.size()) return lpt::fail(LPT_ERR_INVALID_ARG, "lpt_decode_image: buffer of %zu bytes, %zu needed", capacity, im.rgba8.size());
        memcpy(rgba8, im.rgba8.data(), im.rgba8.size());
    }
    return LPT_OK;
}


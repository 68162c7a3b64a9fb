// fuzz_loaders.cpp — development tool (CPU, AddressSanitizer + UBSan): mutates seed files and feeds them to the host-side
// loaders of libloupiote_hip.so (glTF / GLB, PNG, baseline JPEG, Radiance HDR).  The loaders must reject or accept every
// input without reading or writing out of bounds.  Built and run by tests/test_loader_fuzz.py:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I include fuzz_loaders.cpp
//       loupiote_amd/csrc/{scene,gltf,png,jpeg,hdr}.cpp -o fuzz_loaders;  ./fuzz_loaders <iterations>[:<rng seed>] <seed files...>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../loupiote_amd/csrc/common.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17; return (uint32_t)(g_state >> 16); }

static void mutate(std::vector<uint8_t> &b) {
    if (b.empty()) return;
    const uint32_t n = 1 + rnd() % 6;
    for (uint32_t k = 0; k < n; ++k) {
        const size_t i = rnd() % b.size();
        switch (rnd() % 7) {
            case 0: b[i] = (uint8_t)rnd(); break;
            case 1: b[i] ^= (uint8_t)(1u << (rnd() % 8)); break;
            case 2: b[i] = (rnd() & 1) ? 0xFF : 0x00; break;
            case 3: if (b.size() > 8) b.resize(b.size() - 1 - rnd() % (b.size() / 4 + 1)); break;          // truncate
            case 4: { const size_t j = rnd() % b.size(); std::swap(b[i], b[j]); break; }
            case 5: {                                                                                     // splice a run of digits / a big number into text
                static const char *tok[] = {"-1", "4294967295", "18446744073709551616", "1e30", "-0.5", "999999999", "0"};
                const char *t = tok[rnd() % 7];
                for (size_t q = 0; t[q] && i + q < b.size(); ++q) b[i + q] = (uint8_t)t[q];
                break;
            }
            default: if (i + 4 <= b.size()) { const uint32_t v = (rnd() & 1) ? 0xFFFFFFFFu : rnd(); memcpy(&b[i], &v, 4); } break;   // 32-bit length fields
        }
    }
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: fuzz_loaders <iterations>[:<rng seed>] <seed files...>\n"); return 2; }
    const long iters = atol(argv[1]);
    if (const char *c = strchr(argv[1], ':')) g_state ^= (uint64_t)atoll(c + 1) * 0x100000001B3ull;
    long accepted = 0, rejected = 0;
    for (int f = 2; f < argc; ++f) {
        std::ifstream in(argv[f], std::ios::binary);
        const std::vector<uint8_t> seed((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        const std::string name = argv[f];
        for (long it = 0; it < iters; ++it) {
            std::vector<uint8_t> b = seed;
            if (it) mutate(b);
            // heap copy of the exact size: ASan sees any read past the end
            uint8_t *p = (uint8_t *)malloc(b.size() ? b.size() : 1);
            if (!b.empty()) memcpy(p, b.data(), b.size());
            bool ok = false;
            if (name.find(".hdr") != std::string::npos) {
                uint32_t w = 0, h = 0;
                if (lpt_decode_hdr(p, b.size(), nullptr, 0, &w, &h) == LPT_OK && (size_t)w * h <= (1u << 24)) {
                    std::vector<uint8_t> out((size_t)w * h * 4);
                    ok = lpt_decode_hdr(p, b.size(), out.data(), out.size(), &w, &h) == LPT_OK;
                }
            } else if (name.find(".png") != std::string::npos) {
                lpt::Image im;
                ok = lpt::decode_png(p, b.size(), im);
            } else if (name.find(".jpg") != std::string::npos) {
                lpt::Image im;
                ok = lpt::decode_jpeg(p, b.size(), im);
            } else {
                lpt_scene *s = nullptr;
                lpt_scene_create(&s);
                ok = lpt_load_gltf(s, p, b.size()) == LPT_OK;
                lpt_scene_destroy(s);
            }
            free(p);
            (ok ? accepted : rejected)++;
        }
    }
    printf("FUZZ_OK accepted %ld rejected %ld\n", accepted, rejected);
    return 0;
}

// node_fetch.hip — round 6: what the LAST of the five 16-byte fetches of an 80-byte node costs, and what it would cost as 12 or 8 bytes (same 80-byte stride, the
// record's tail unused), beside a 64-byte record — per-lane divergent fetches as in k_trace's node visit (tools/microbench/gather_rate.hip is the general tool).
// Every lane chases a pseudo-random sequence of records in a 2 MiB buffer (the bench scene's tree); LANES of 64 are active.
// build: hipcc --offload-arch=gfx950 -O3 -o node_fetch node_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int STRIDE, int FULL, int TAIL, int LANES>   // FULL 16-byte loads, then one of TAIL bytes (0, 4, 8, 12)
__global__ __launch_bounds__(64) void k(const unsigned char *buf, uint32_t mask, int iters, uint32_t *out) {
    if ((int)threadIdx.x >= LANES) return;
    uint32_t idx = (blockIdx.x * 64 + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t rec = (idx >> 7) & mask;
        const unsigned char *p = buf + (size_t)rec * STRIDE;
        uint4 v[FULL];
#pragma unroll
        for (int l = 0; l < FULL; ++l) v[l] = reinterpret_cast<const uint4 *>(p)[l];
        uint32_t t = 0;
        if (TAIL == 12) { const uint3 w = *reinterpret_cast<const uint3 *>(p + 16 * FULL); t = w.x ^ w.y ^ w.z; }
        if (TAIL == 8) { const uint2 w = *reinterpret_cast<const uint2 *>(p + 16 * FULL); t = w.x ^ w.y; }
        if (TAIL == 4) { t = *reinterpret_cast<const uint32_t *>(p + 16 * FULL); }
#pragma unroll
        for (int l = 0; l < FULL; ++l) acc += v[l].x ^ v[l].y ^ v[l].z ^ v[l].w;
        acc += t;
        idx = idx * 1664525u + 1013904223u + (acc & 1u);
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int STRIDE, int FULL, int TAIL, int LANES>
void run(const unsigned char *buf, uint32_t *out, int waves_per_simd) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 4 * waves_per_simd;
    const size_t bytes = 2u << 20;
    const uint32_t recs = (uint32_t)(bytes / STRIDE);
    uint32_t mask = 1;
    while (mask * 2 <= recs) mask *= 2;
    mask -= 1;
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<STRIDE, FULL, TAIL, LANES><<<blocks, 64>>>(buf, mask, 200, out);
    (void)hipEventRecord(e0);
    k<STRIDE, FULL, TAIL, LANES><<<blocks, 64>>>(buf, mask, iters, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("stride %3d: %d x 16 B + %2d B = %3d B per record, %2d lanes, %d waves/SIMD: %.3f ms | %.1f ns per wave-visit per CU\n", STRIDE, FULL, TAIL, 16 * FULL + TAIL, LANES,
           waves_per_simd, ms, ms * 1e6 / ((double)iters * 4 * waves_per_simd));
}

int main() {
    unsigned char *buf;
    uint32_t *out;
    (void)hipMalloc(&buf, 4u << 20);
    (void)hipMemset(buf, 1, 4u << 20);
    (void)hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<80, 5, 0, 64>(buf, out, 6);
        run<80, 4, 12, 64>(buf, out, 6);
        run<80, 4, 8, 64>(buf, out, 6);
        run<80, 4, 4, 64>(buf, out, 6);
        run<80, 4, 0, 64>(buf, out, 6);
        run<64, 4, 0, 64>(buf, out, 6);
        run<72, 4, 8, 64>(buf, out, 6);    // 16-byte loads at 8-byte alignment
        run<80, 5, 0, 40>(buf, out, 6);
        run<80, 4, 12, 40>(buf, out, 6);
        run<80, 4, 8, 40>(buf, out, 6);
        run<80, 4, 0, 40>(buf, out, 6);
        run<64, 4, 0, 40>(buf, out, 6);
        run<72, 4, 8, 40>(buf, out, 6);
    }
    return 0;
}

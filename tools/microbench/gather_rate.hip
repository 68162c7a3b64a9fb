// gather_rate.hip — throughput of per-lane divergent loads on gfx950 (development tool).
// Every lane chases a pseudo-random sequence of 128-byte records in a buffer of `mb` MiB and issues
// LOADS loads of WIDTH bytes from each record; prints record visits and load instructions per cycle per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int LOADS, int STRIDE, int WIDTH, int LANES>
__global__ __launch_bounds__(64) void k(const uint4 *buf, uint32_t mask, int iters, uint32_t *out) {
    if ((int)threadIdx.x >= LANES) return;
    uint32_t idx = (blockIdx.x * 64 + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t rec = (idx >> 7) & mask;                  // record index
        const uint4 *p = buf + (size_t)rec * (STRIDE / 16);
        if (WIDTH == 16) {
            uint4 v[LOADS];
#pragma unroll
            for (int l = 0; l < LOADS; ++l) v[l] = p[l];
#pragma unroll
            for (int l = 0; l < LOADS; ++l) acc += v[l].x ^ v[l].y ^ v[l].z ^ v[l].w;
        } else if (WIDTH == 8) {
            uint2 v[LOADS];
#pragma unroll
            for (int l = 0; l < LOADS; ++l) v[l] = reinterpret_cast<const uint2 *>(p)[l];
#pragma unroll
            for (int l = 0; l < LOADS; ++l) acc += v[l].x ^ v[l].y;
        } else {
            uint32_t v[LOADS];
#pragma unroll
            for (int l = 0; l < LOADS; ++l) v[l] = reinterpret_cast<const uint32_t *>(p)[l];
#pragma unroll
            for (int l = 0; l < LOADS; ++l) acc += v[l];
        }
        idx = idx * 1664525u + 1013904223u + (acc & 1u);          // next record depends on the data (like a tree walk)
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

// Cooperative form of the 80-byte record fetch: the 64 lanes of a wave fetch the 64 records their lanes want as 5 load
// instructions in which lane L of instruction r reads 16-byte chunk (64 r + L) % 5 of the record of owner lane (64 r + L) / 5 —
// five neighbouring lanes read one record, so an instruction touches ~13 records (13-26 cache lines) instead of 64 — and hands
// the data to the owners through LDS (linear 16-byte stores, then five 16-byte loads at an 80-byte lane stride: conflict-free).
template <int MODE>
__global__ __launch_bounds__(64) void k_coop(const uint4 *buf, uint32_t mask, int iters, uint32_t *out) {
    __shared__ uint32_t s_idx[64];
    __shared__ uint4 s_data[64 * 5];
    const uint32_t lane = threadIdx.x;
    uint32_t idx = (blockIdx.x * 64 + threadIdx.x) * 2654435761u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t rec = (idx >> 7) & mask;
        s_idx[lane] = rec;
        uint4 v[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const uint32_t g = 64u * r + lane, owner = g / 5u, chunk = g - owner * 5u;
            const uint32_t orec = s_idx[owner];
            v[r] = buf[(size_t)orec * 5u + chunk];
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) s_data[64 * r + lane] = v[r];
        uint4 n[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) n[c] = s_data[lane * 5 + c];
#pragma unroll
        for (int c = 0; c < 5; ++c) acc += n[c].x ^ n[c].y ^ n[c].z ^ n[c].w;
        idx = idx * 1664525u + 1013904223u + (acc & 1u);
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

void run_coop(const uint4 *buf, size_t bytes, uint32_t *out, int waves_per_simd) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 4 * waves_per_simd;
    const uint32_t recs = (uint32_t)(bytes / 80);
    uint32_t mask = 1;
    while (mask * 2 <= recs) mask *= 2;
    mask -= 1;
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_coop<0><<<blocks, 64>>>(buf, mask, 200, out);
    hipEventRecord(e0);
    k_coop<0><<<blocks, 64>>>(buf, mask, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double cycles = ms * 1e-3 * 2.4e9;
    const double visits_per_cu = (double)iters * 4 * waves_per_simd;
    printf("COOPERATIVE buffer %8.3f MiB stride 80 (5 lanes per record, LDS hand-over) waves/SIMD %d: %.3f ms | %.1f cycles per wave-visit per CU | %.2f TB/s\n",
           bytes / 1048576.0, waves_per_simd, ms, cycles / visits_per_cu, (double)blocks * 64 * iters * 80 / (ms * 1e-3) / 1e12);
}

template <int LOADS, int STRIDE, int WIDTH = 16, int LANES = 64>
void run(const uint4 *buf, size_t bytes, uint32_t *out, int waves_per_simd) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 4 * waves_per_simd;
    const uint32_t recs = (uint32_t)(bytes / STRIDE);
    uint32_t mask = 1;
    while (mask * 2 <= recs) mask *= 2;
    mask -= 1;
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<LOADS, STRIDE, WIDTH, LANES><<<blocks, 64>>>(buf, mask, 200, out);
    hipEventRecord(e0);
    k<LOADS, STRIDE, WIDTH, LANES><<<blocks, 64>>>(buf, mask, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double cycles = ms * 1e-3 * 2.4e9;
    const double visits_per_cu = (double)iters * 4 * waves_per_simd;  // wave-visits per CU
    printf("buffer %8.3f MiB stride %3d width %2d lanes %2d loads/visit %d waves/SIMD %d: %.3f ms | %.1f cycles per wave-visit per CU | %.1f cycles per load instr per CU | %.2f TB/s\n",
           bytes / 1048576.0, STRIDE, WIDTH, LANES, LOADS, waves_per_simd, ms, cycles / visits_per_cu, cycles / visits_per_cu / LOADS,
           (double)blocks * LANES * iters * LOADS * WIDTH / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t max_bytes = 64u << 20;
    uint4 *buf;
    uint32_t *out;
    hipMalloc(&buf, max_bytes);
    hipMemset(buf, 1, max_bytes);
    hipMalloc(&out, 256 * 4 * 8 * 64 * 4);
    const size_t b2 = 2u << 20;
    run<4, 128, 16, 64>(buf, b2, out, 8);
    run<4, 128, 8, 64>(buf, b2, out, 8);
    run<4, 128, 4, 64>(buf, b2, out, 8);
    run<8, 128, 8, 64>(buf, b2, out, 8);
    run<8, 128, 4, 64>(buf, b2, out, 8);
    run<4, 128, 16, 32>(buf, b2, out, 8);
    run<4, 128, 16, 16>(buf, b2, out, 8);
    run<4, 128, 16, 8>(buf, b2, out, 8);
    run<5, 80, 16, 64>(buf, b2, out, 8);
    run<5, 80, 16, 64>(buf, 16u << 10, out, 8);
    run<4, 128, 16, 64>(buf, 16u << 10, out, 8);
    run<5, 80, 16, 64>(buf, 256u << 10, out, 8);
    run<5, 80, 16, 64>(buf, 1u << 20, out, 8);
    run<5, 80, 16, 64>(buf, 4u << 20, out, 8);
    run<5, 80, 16, 64>(buf, 8u << 20, out, 8);
    run<5, 80, 16, 64>(buf, 64u << 20, out, 8);
    run<5, 80, 16, 64>(buf, b2, out, 4);
    run<5, 80, 16, 64>(buf, b2, out, 2);
    for (int w : {8, 6, 4, 2}) run_coop(buf, b2, out, w);
    run_coop(buf, 16u << 20, out, 8);
    run_coop(buf, 16u << 20, out, 4);
    run<5, 80, 16, 64>(buf, 16u << 20, out, 8);
    return 0;
}

// valu_rate.hip — instruction issue-rate microbenchmark for gfx950 (development tool, not part of the library).
// Each kernel runs ITER iterations of 16 independent register chains of one instruction; the printed figure is
// wave-instructions per cycle per SIMD (1/4 = one wave64 op every 4 cycles = full rate for a 16-lane SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
constexpr int ITER = 4096;

// cyc[block] = s_memtime ticks (shader cycles: the clock the SIMD really ran at, DVFS included) the wave spent in the timed loop
template <int OP>
__global__ __launch_bounds__(64) void k(float *out, float a, float b, unsigned qa, unsigned long long *cyc) {
    float v[16];
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f w[8];
    for (int i = 0; i < 16; ++i) v[i] = (float)threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; ++i) w[i] = v2f{v[2 * i], v[2 * i + 1]};
    unsigned q = qa + threadIdx.x;
    unsigned u[16];
    for (int i = 0; i < 16; ++i) u[i] = q * (i + 1);
    asm volatile("v_cmp_gt_u32_e32 vcc, 7, %0\n v_cmp_gt_u32_e64 s[20:21], 9, %0" : : "v"(q) : "vcc", "s20", "s21");
    v2f ab = {a, b};
    const unsigned long long c_begin = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        }
        else if (OP == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(w[i & 7]) : "v"(ab));
            REP16(X)
#undef X
        }
        else if (OP == 2) {
#define X(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(q), "v"(a));
            REP16(X)
#undef X
        }
        else if (OP == 3) {
#define X(i) asm volatile("v_cvt_f32_ubyte1_e32 %0, %1" : "=v"(v[i]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 4) {
#define X(i) asm volatile("v_cvt_f32_ubyte0_e32 %0, %1" : "=v"(v[i]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 5) {
#define X(i) asm volatile("v_cvt_f32_u32_e32 %0, %1" : "=v"(v[i]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 6) {
#define X(i) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(v[i]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 7) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        }
        else if (OP == 8) {
#define X(i) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            REP16(X)
#undef X
        }
        else if (OP == 9) {
#define X(i) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            REP16(X)
#undef X
        }
        else if (OP == 10) {
#define X(i) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            REP16(X)
#undef X
        }
        else if (OP == 11) {
#define X(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        }
        else if (OP == 12) {
#define X(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a));
            REP16(X)
#undef X
        }
        else if (OP == 13) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[i]) : "v"(a) : "s20", "s21");
            REP16(X)
#undef X
        }
        else if (OP == 14) {
#define X(i) asm volatile("v_cmp_le_f32_e32 vcc, %0, %1" : : "v"(v[i]), "v"(a) : "vcc");
            REP16(X)
#undef X
        }
        else if (OP == 15) {
#define X(i) asm volatile("v_cmp_le_f32_e64 s[20:21], %0, %1" : : "v"(v[i]), "v"(a) : "s20", "s21");
            REP16(X)
#undef X
        }
        else if (OP == 16) {
#define X(i) asm volatile("v_bfe_u32 %0, %0, 5, 3" : "+v"(u[i]));
            REP16(X)
#undef X
        }
        else if (OP == 17) {
#define X(i) asm volatile("v_lshlrev_b32_e32 %0, %1, %0" : "+v"(u[i]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 18) {
#define X(i) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(u[i]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 19) {
#define X(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(q), "v"(qa));
            REP16(X)
#undef X
        }
        else if (OP == 20) {
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(u[i]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 21) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(q), "v"(qa));
            REP16(X)
#undef X
        }
        else if (OP == 22) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(q), "v"(qa));
            REP16(X)
#undef X
        }
        else if (OP == 23) {
#define X(i) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(u[i]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 24) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u[i]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 25) {
#define X(i) asm volatile("v_sub_f32_e32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            REP16(X)
#undef X
        }
        else if (OP == 26) {
#define X(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        }
        else if (OP == 27) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        }
        else if (OP == 28) {
#define X(i) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(v[i]) : "v"(a));
            REP16(X)
#undef X
        }
        else if (OP == 29) {
#define X(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(u[i]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 30) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "s"(a), "v"(b));
            REP16(X)
#undef X
        }
        else if (OP == 31) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[i & 7]) : "v"(ab));
            REP16(X)
#undef X
        }
        else if (OP == 32) {
#define X(i) asm volatile("v_cvt_pk_f32_fp8_e32 %0, %1" : "=v"(w[i & 7]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 33) {
#define X(i) asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(u[i]) : "v"(q), "v"(qa));
            REP16(X)
#undef X
        }
        else if (OP == 34) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a));
            REP16(X)
#undef X
        }
        else if (OP == 35) {
#define X(i) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(v[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        }
        else if (OP == 36) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(v[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        }
        else if (OP == 37) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, 0, %0, s[20:21]" : "+v"(v[i]));
            REP16(X)
#undef X
        }
        else if (OP == 38) {
#define X(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_fma_f32 %2, %2, %1, %1\n v_fma_f32 %3, %3, %1, %1\n v_fma_f32 %4, %4, %1, %1" : "+v"(v[i]), "+v"(w[0].x), "+v"(w[1].x), "+v"(w[2].x) : "v"(a));
            REP16(X)
#undef X
        }
        else if (OP == 39) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_fma_f32 %2, %2, %1, %1\n v_fma_f32 %3, %3, %1, %1\n v_fma_f32 %4, %4, %1, %1" : "+v"(v[i]), "+v"(w[0].x), "+v"(w[1].x), "+v"(w[2].x) : "v"(a));
            REP16(X)
#undef X
        }
        else if (OP == 40) {
#define X(i) asm volatile("v_cmp_le_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : "vcc");
            REP16(X)
#undef X
        }
        else if (OP == 41) {
#define X(i) asm volatile("v_cmp_le_f32_e64 vcc, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(a) : "vcc");
            REP16(X)
#undef X
        }
        else if (OP == 42) {
#define X(i) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(u[i]) : "v"(q) : "vcc");
            REP16(X)
#undef X
        }
        else if (OP == 43) {
#define X(i) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(u[i]) : "s20");
            REP16(X)
#undef X
        }
        else if (OP == 44) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(v[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        }
        else if (OP == 45) {
#define X(i) asm volatile("v_fma_f32 %0, |%0|, %1, |%2|" : "+v"(v[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        }
        else if (OP == 46) {
#define X(i) asm volatile("v_add_f32_e64 %0, %0, %1 clamp" : "+v"(v[i]) : "v"(a));
            REP16(X)
#undef X
        }
        else if (OP == 47) {
#define X(i) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(v[i]) : "v"(a));
            REP16(X)
#undef X
        }
        else if (OP == 48) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2 mul:2" : "+v"(v[i]) : "v"(a), "v"(b));
            REP16(X)
#undef X
        }
        else if (OP == 50) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 51) {
#define X(i) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(u[i]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 52) {
#define X(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(u[i]) : "v"(q), "v"(qa));
            REP16(X)
#undef X
        }
        else if (OP == 53) {
#define X(i) asm volatile("v_ffbh_u32_e32 %0, %1" : "=v"(u[i]) : "v"(q));
            REP16(X)
#undef X
        }
        else if (OP == 54) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long *)&w[i & 7]) : "v"(q), "v"(qa) : "vcc");
            REP16(X)
#undef X
        }
        else if (OP == 49) {
#define X(i) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %2, 0, %3, vcc" : : "v"(v[i]), "v"(a), "v"(u[i]), "v"(q) : "vcc");
            REP16(X)
#undef X
        }
    }
    const unsigned long long c_end = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = c_end - c_begin;
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += w[i].x + w[i].y;
    for (int i = 0; i < 16; ++i) s += (float)u[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

__global__ void k_clock(unsigned long long *out) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x;
    for (int i = 0; i < 2000000; ++i) asm volatile("v_add_f32_e32 %0, %0, %0" : "+v"(x));
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (unsigned long long)x; }
}

static unsigned long long *g_cyc = nullptr;
template <int OP>
void run(const char *name, int per_iter, float *d, int waves_per_simd) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 4 * waves_per_simd;
    if (!g_cyc) hipMalloc(&g_cyc, sizeof(unsigned long long) * 256 * 4 * 8 * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 64>>>(d, 1.0001f, 0.5f, 0x3c003c00u, g_cyc);
    hipEventRecord(e0);
    k<OP><<<blocks, 64>>>(d, 1.0001f, 0.5f, 0x3c003c00u, g_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), g_cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mean_cyc = 0;
    for (auto c : h) mean_cyc += (double)c;
    mean_cyc /= blocks;
    const double clk = (double)p.clockRate * 1e3;  // Hz, nominal
    const double insts_per_simd = (double)ITER * per_iter * waves_per_simd;
    // two denominators: wall time x the NOMINAL clock (what round 2 quoted), and the shader cycles the waves really saw
    // (s_memtime): the second is the issue rate, the ratio of the two is the clock the SIMDs ran at under this instruction
    printf("%-18s waves/SIMD %d : %.3f ms, %.3f wave-inst/cycle/SIMD at the nominal %.0f MHz | %.3f per REAL shader cycle (s_memtime: %.0f cycles per wave => %.0f MHz effective)\n",
           name, waves_per_simd, ms, insts_per_simd / (ms * 1e-3 * clk), clk / 1e6, insts_per_simd / mean_cyc, mean_cyc, mean_cyc / (ms * 1e-3) / 1e6);
}

int main() {
    float *d;
    hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(float) * 2);
    {
        unsigned long long *dc, h[3];
        hipMalloc(&dc, 24);
        k_clock<<<1, 64>>>(dc);
        hipMemcpy(h, dc, 24, hipMemcpyDeviceToHost);
        int wc = 0;
        hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
        printf("clock64 ticks %llu, wall_clock64 ticks %llu (wall clock rate %d kHz) -> clock64 runs at %.1f MHz\n", h[0], h[1], wc, (double)h[0] / ((double)h[1] / (wc * 1e3)) / 1e6);
    }
    for (int w : {2, 8}) {
        run<0>("v_fma_f32", 16, d, w);
        run<1>("v_pk_fma_f32", 16, d, w);
        run<2>("v_fma_mix_f32", 16, d, w);
        run<3>("v_cvt_f32_ubyte1", 16, d, w);
        run<4>("v_cvt_f32_ubyte0", 16, d, w);
        run<5>("v_cvt_f32_u32", 16, d, w);
        run<6>("v_cvt_f32_f16", 16, d, w);
        run<7>("v_max3_f32", 16, d, w);
        run<8>("v_max_f32", 16, d, w);
        run<9>("v_mul_f32", 16, d, w);
        run<10>("v_add_f32", 16, d, w);
        run<11>("v_fmac_f32", 16, d, w);
        run<12>("v_cndmask_vcc", 16, d, w);
        run<13>("v_cndmask_sgpr", 16, d, w);
        run<14>("v_cmp_le_f32_vcc", 16, d, w);
        run<15>("v_cmp_le_f32_sgpr", 16, d, w);
        run<16>("v_bfe_u32", 16, d, w);
        run<17>("v_lshlrev_b32", 16, d, w);
        run<18>("v_and_b32", 16, d, w);
        run<19>("v_or3_b32", 16, d, w);
        run<20>("v_alignbit_b32", 16, d, w);
        run<21>("v_perm_b32", 16, d, w);
        run<22>("v_mad_u32_u24", 16, d, w);
        run<23>("v_add_u32", 16, d, w);
        run<24>("v_lshl_add_u32", 16, d, w);
        run<25>("v_sub_f32", 16, d, w);
        run<26>("v_min3_f32", 16, d, w);
        run<27>("v_med3_f32", 16, d, w);
        run<28>("v_mov_b32", 16, d, w);
        run<29>("v_bcnt_u32_b32", 16, d, w);
        run<30>("v_fma_f32 (sgpr)", 16, d, w);
        run<31>("v_pk_add_f32", 16, d, w);
        run<32>("v_cvt_pk_f32_fp8", 16, d, w);
        run<33>("v_dot4_u32_u8", 16, d, w);
        run<34>("v_cndmask_e64_vcc", 16, d, w);
        run<35>("v_cndmask_e32_dst!=src", 16, d, w);
        run<36>("v_cndmask_sgpr_dst!=src", 16, d, w);
        run<37>("v_cndmask_sgpr_0", 16, d, w);
        run<38>("e32cnd+3fma", 64, d, w);
        run<39>("e64cnd+3fma", 64, d, w);
        run<40>("cmp_e32+cnd_e32", 32, d, w);
        run<41>("cmp_e64vcc+cnd_e64vcc", 32, d, w);
        run<42>("v_addc_co_u32_e32", 16, d, w);
        run<43>("v_readfirstlane", 16, d, w);
        run<44>("v_fma_f32 clamp", 16, d, w);
        run<45>("v_fma_f32 |abs|", 16, d, w);
        run<46>("v_add_f32_e64 clamp", 16, d, w);
        run<47>("v_mul_f32_e64", 16, d, w);
        run<48>("v_fma_f32 mul:2", 16, d, w);
        run<50>("v_mul_lo_u32", 16, d, w);
        run<51>("v_mul_u32_u24", 16, d, w);
        run<52>("v_bitop3_b32", 16, d, w);
        run<53>("v_ffbh_u32", 16, d, w);
        run<54>("v_mad_u64_u32", 16, d, w);
    }
    return 0;
}

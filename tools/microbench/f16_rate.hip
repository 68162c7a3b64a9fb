// f16_rate.hip — round 6 (VERDICT r05 #1 i): issue rates of the packed-f16 / byte-shuffle instructions a two-children-per-instruction
// box test would be made of, on gfx950, beside v_fma_f32 / v_max_f32 / v_cvt_f32_ubyte as the known fast / slow classes
// (profiles/r03_valu_rate_real_clock.txt).  Same method as valu_rate.hip: 16 independent chains per wave, wave-instructions per REAL shader
// cycle (s_memtime) per SIMD.  Then a functional part: what v_perm_b32 selects, whether v_pk_fma_f16 honours f16 denormal inputs, what `clamp`
// and v_pk_maximum3_f16 do with the values the box test feeds them.
// build: hipcc --offload-arch=gfx950 -O3 -o f16_rate f16_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
constexpr int ITER = 4096;

#define OPS(F) \
    F(0, "v_fma_f32", "v_fma_f32 %0, %0, %2, %3") \
    F(1, "v_max_f32", "v_max_f32_e32 %0, %0, %2") \
    F(2, "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte1_e32 %0, %2") \
    F(3, "v_pk_fma_f16", "v_pk_fma_f16 %0, %0, %2, %3") \
    F(4, "v_pk_fma_f16 clamp", "v_pk_fma_f16 %0, %0, %2, %3 clamp") \
    F(5, "v_pk_mul_f16", "v_pk_mul_f16 %0, %0, %2") \
    F(6, "v_pk_add_f16", "v_pk_add_f16 %0, %0, %2") \
    F(7, "v_pk_add_f16 neg", "v_pk_add_f16 %0, %0, %2 neg_lo:[0,1] neg_hi:[0,1]") \
    F(8, "v_pk_min_f16", "v_pk_min_f16 %0, %0, %2") \
    F(9, "v_pk_max_f16", "v_pk_max_f16 %0, %0, %2") \
    F(10, "v_pk_maximum3_f16", "v_pk_maximum3_f16 %0, %0, %2, %3") \
    F(11, "v_pk_minimum3_f16", "v_pk_minimum3_f16 %0, %0, %2, %3") \
    F(12, "v_maximum3_f32", "v_maximum3_f32 %0, %0, %2, %3") \
    F(13, "v_perm_b32", "v_perm_b32 %0, %0, %2, %3") \
    F(14, "v_perm_b32 (sgpr src1)", "v_perm_b32 %0, %0, s20, %3") \
    F(15, "v_pk_lshrrev_b16", "v_pk_lshrrev_b16 %0, 8, %0") \
    F(16, "v_pk_ashrrev_i16", "v_pk_ashrrev_i16 %0, 15, %0") \
    F(17, "v_cmp_le_f16_sdwa hi", "v_cmp_le_f16_sdwa s[22:23], %0, %2 src0_sel:WORD_1 src1_sel:WORD_1") \
    F(18, "v_cmp_le_f16_e32", "v_cmp_le_f16_e32 vcc, %0, %2") \
    F(19, "v_and_or_b32", "v_and_or_b32 %0, %0, %2, %3") \
    F(20, "v_bfi_b32", "v_bfi_b32 %0, %0, %2, %3") \
    F(21, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 3, %2") \
    F(22, "v_or_b32", "v_or_b32_e32 %0, %2, %0") \
    F(23, "v_xor_b32", "v_xor_b32_e32 %0, %2, %0") \
    F(24, "v_lshrrev_b32", "v_lshrrev_b32_e32 %0, 8, %0") \
    F(25, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %0, %2, %3") \
    F(26, "v_rcp_f32", "v_rcp_f32_e32 %0, %0") \
    F(27, "v_pk_mad_u16", "v_pk_mad_u16 %0, %0, %2, %3") \
    F(28, "v_pk_min_u16", "v_pk_min_u16 %0, %0, %2") \
    F(29, "v_pk_sub_i16", "v_pk_sub_i16 %0, %0, %2") \
    F(30, "v_dot4_u32_u8", "v_dot4_u32_u8 %0, %2, %3, %0") \
    F(31, "v_sad_u8", "v_sad_u8 %0, %2, %3, %0") \
    F(32, "v_cvt_f16_u16_sdwa byte", "v_cvt_f16_u16_sdwa %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2") \
    F(33, "v_ldexp_f32", "v_ldexp_f32 %0, %0, %2") \
    F(34, "v_frexp_exp_i32_f32", "v_frexp_exp_i32_f32_e32 %0, %2") \
    F(35, "v_bitop3_b32", "v_bitop3_b32 %0, %0, %2, %3 bitop3:0x6c") \
    F(36, "v_and_b32", "v_and_b32_e32 %0, %2, %0") \
    F(37, "v_add_u32", "v_add_u32_e32 %0, %2, %0") \
    F(38, "v_sub_u32", "v_sub_u32_e32 %0, %2, %0") \
    F(39, "v_not_b32", "v_not_b32_e32 %0, %0") \
    F(40, "v_min_u32", "v_min_u32_e32 %0, %2, %0") \
    F(41, "v_mul_f16", "v_mul_f16_e32 %0, %2, %0") \
    F(42, "v_fma_f16", "v_fma_f16 %0, %0, %2, %3") \
    F(43, "v_max_f16_sdwa hi", "v_max_f16_sdwa %0, %0, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1") \
    F(44, "v_pk_fma_f32", "v_pk_fma_f32 %1, %1, %4, %4") \
    F(45, "v_lshlrev_b32 const", "v_lshlrev_b32_e32 %0, 4, %0") \
    F(46, "v_bfe_u32", "v_bfe_u32 %0, %0, 5, 3") \
    F(47, "v_cndmask_b32 sgpr", "v_cndmask_b32_e64 %0, %0, %2, s[22:23]") \
    F(48, "v_mov_b32", "v_mov_b32_e32 %0, %2") \
    F(49, "v_pk_add_u16", "v_pk_add_u16 %0, %0, %2") \
    F(50, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %2") \
    F(51, "v_mul_u32_u24", "v_mul_u32_u24_e32 %0, %2, %0") \
    F(52, "v_fma_mix_f32 (f16 b)", "v_fma_mix_f32 %0, %0, %2, %3 op_sel_hi:[0,1,0]") \
    F(53, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %2, %3") \
    F(54, "v_mov_b32_sdwa byte->byte1 preserve", "v_mov_b32_sdwa %0, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2") \
    F(55, "v_or_b32_sdwa src byte", "v_or_b32_sdwa %0, %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2") \
    F(56, "v_mov_b32_sdwa byte->dword", "v_mov_b32_sdwa %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2") \
    F(57, "v_addc_co_u32", "v_addc_co_u32_e32 %0, vcc, %0, %0, vcc") \
    F(58, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %2, 31") \
    F(59, "v_cmp_le_f32 vcc", "v_cmp_le_f32_e32 vcc, %0, %2") \
    F(60, "v_min3_f32", "v_min3_f32 %0, %0, %2, %3") \
    F(61, "v_add_f32", "v_add_f32_e32 %0, %2, %0") \
    F(62, "v_mul_f32", "v_mul_f32_e32 %0, %2, %0") \
    F(63, "v_fmac_f32 (VOP2)", "v_fmac_f32_e32 %0, %2, %3") \
    F(64, "v_fma_f32 clamp", "v_fma_f32 %0, %0, %2, %3 clamp") \
    F(65, "v_max3_f32", "v_max3_f32 %0, %0, %2, %3") \
    F(66, "v_med3_f32", "v_med3_f32 %0, %0, %2, %3") \
    F(67, "v_cmp_lt_f32 vcc", "v_cmp_lt_f32_e32 vcc, %0, %2")

template <int OP>
__global__ __launch_bounds__(64) void k(uint32_t *out, uint32_t a, uint32_t b, unsigned long long *cyc) {
    uint32_t u[16];
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f w[8];
    for (int i = 0; i < 16; ++i) u[i] = 0x3c003c00u + threadIdx.x * 0x00010001u + i;     // two f16 near 1.0
    for (int i = 0; i < 8; ++i) w[i] = v2f{1.0f + i, 2.0f + i};
    v2f ab = {1.0001f, 0.5f};
    uint32_t va = a + (threadIdx.x & 1), vb = b + (threadIdx.x & 3);
    asm volatile("v_cmp_gt_u32_e64 s[22:23], 9, %0\n s_mov_b32 s20, 0x64646464" : : "v"(va) : "s22", "s23", "s20");
    const unsigned long long c_begin = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#define MK(id, text, i) asm volatile(text : "+v"(u[i]), "+v"(w[i & 7]) : "v"(va), "v"(vb), "v"(ab) : "vcc");
        switch (OP) {
#define CASE(id, name, text) case id: { MK(id, text, 0) MK(id, text, 1) MK(id, text, 2) MK(id, text, 3) MK(id, text, 4) MK(id, text, 5) MK(id, text, 6) MK(id, text, 7) \
                                        MK(id, text, 8) MK(id, text, 9) MK(id, text, 10) MK(id, text, 11) MK(id, text, 12) MK(id, text, 13) MK(id, text, 14) MK(id, text, 15) } break;
            OPS(CASE)
#undef CASE
        }
    }
    const unsigned long long c_end = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = c_end - c_begin;
    uint32_t s = 0;
    for (int i = 0; i < 16; ++i) s += u[i];
    for (int i = 0; i < 8; ++i) s += (uint32_t)(w[i].x + w[i].y);
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

static unsigned long long *g_cyc = nullptr;
template <int OP>
void run(const char *name, uint32_t *d, int waves_per_simd) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 4 * waves_per_simd;
    if (!g_cyc) hipMalloc(&g_cyc, sizeof(unsigned long long) * 256 * 4 * 8 * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 64>>>(d, 0x3c013c01u, 0x38003800u, g_cyc);
    hipEventRecord(e0);
    k<OP><<<blocks, 64>>>(d, 0x3c013c01u, 0x38003800u, g_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), g_cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mean_cyc = 0;
    for (auto c : h) mean_cyc += (double)c;
    mean_cyc /= blocks;
    const double insts_per_simd = (double)ITER * 16 * waves_per_simd;
    printf("%-28s waves/SIMD %d : %.3f ms | %.3f wave-inst per REAL shader cycle per SIMD (%.2f cycles each)\n", name, waves_per_simd, ms, insts_per_simd / mean_cyc,
           mean_cyc / insts_per_simd);
}

// ---- functional part -------------------------------------------------------------------------------------------------------------------------
__global__ void k_func(uint32_t *out) {
    const uint32_t q = 0xC8FF0103u;                   // bytes (b3..b0) = 200, 255, 1, 3
    uint32_t r;
    // v_perm_b32 D, S0, S1, sel: byte i of D = byte sel[i] of {S0 (bytes 7..4), S1 (bytes 3..0)}; 0x0c = 0x00
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(0x64646464u), "v"(q), "v"(0x04010400u)); out[0] = r;   // expect 0x64 01 64 03
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(0x64646464u), "v"(q), "v"(0x04030402u)); out[1] = r;   // expect 0x64 c8 64 ff
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r) : "v"(0x64646464u), "v"(q), "v"(0x0c010c00u)); out[2] = r;   // expect 0x00 01 00 03
    // f16 denormal inputs: (3 * 2^-24) * 2^14 + 0 = 3 * 2^-10
    uint32_t qd = 0x00010003u, A = 0x74007400u /* 2^14 */, B = 0u;
    asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(r) : "v"(qd), "v"(A), "v"(B)); out[3] = r;                    // expect hi 2^-10 = 0x1400, lo 3*2^-10 = 0x1600
    // magic exponent: (1024 + q) * a + (b - 1024 a)
    uint32_t qm = 0x64016403u;   // 1025, 1027
    asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(r) : "v"(qm), "v"(0x38003800u) /* 0.5 */, "v"(0xE000E000u) /* -512 */); out[4] = r;   // 0.5, 1.5 -> 0x3800, 0x3e00
    // clamp: -3 -> 0, 7 -> 1, 0.25 stays
    asm volatile("v_pk_fma_f16 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(0x3c003c00u), "v"(0xC2004700u) /* hi -3, lo 7 */, "v"(0u)); out[5] = r;           // hi 0, lo 0x3c00
    asm volatile("v_pk_fma_f16 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(0x3c003c00u), "v"(0x34007e00u) /* hi .25, lo NaN */, "v"(0u)); out[6] = r;        // hi 0x3400, lo: NaN -> 0 with DX10_CLAMP
    // inf * 0 + x = NaN; maximum3 propagates NaN; v_pk_max_f16 does not
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(0x3c007e00u), "v"(0x40003800u), "v"(0x42003400u)); out[7] = r;   // hi max(1,2,3) = 0x4200, lo NaN
    asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(0x3c007e00u), "v"(0x40003800u)); out[8] = r;                              // hi 2 = 0x4000, lo 0.5 = 0x3800
    asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(0x3c00fc00u), "v"(0x40003800u), "v"(0x42003400u)); out[9] = r;   // hi 1 = 0x3c00, lo -inf = 0xfc00
    // sign of a packed difference: tf - tn
    asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(0x38003800u), "v"(0x38003a00u)); out[10] = r;   // hi 0.5-0.5 = +0 (0x0000), lo 0.5-0.75 = -0.25 (0xb400)
    // v_dot4_u32_u8 of the sign bytes with the bit weights
    asm volatile("v_dot4_u32_u8 %0, %1, %2, %3" : "=v"(r) : "v"(0x80000080u), "v"(0x08040201u), "v"(0u)); out[11] = r;              // 128 * (8 + 1) = 1152
    // overflow of f16: 300 * 300 -> inf; inf - inf -> NaN
    asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(r) : "v"(0x5cb05cb0u), "v"(0x5cb05cb0u), "v"(0u)); out[12] = r;               // 0x7c00 x2
    asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r) : "v"(1.0e9f), "v"(-1.0e-9f)); out[13] = r;                             // rtz: lo 65504 (0x7bff), hi -0 (0x8000)
    uint32_t mode;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE, 0, 12)" : "=s"(mode)); out[14] = mode;                                         // [3:0] round, [7:4] denorm, [8] dx10_clamp, [9] ieee
}

int main() {
    uint32_t *d;
    hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(uint32_t) * 2);
    for (int w : {2, 6}) {
#define RUN(id, name, text) run<id>(name, d, w);
        OPS(RUN)
#undef RUN
    }
    uint32_t *f, h[16];
    hipMalloc(&f, 64);
    k_func<<<1, 1>>>(f);
    hipMemcpy(h, f, 60, hipMemcpyDeviceToHost);
    const char *what[15] = {"perm magic lo pair (expect 64016403)", "perm magic hi pair (expect 64c864ff)", "perm zero-extend (expect 00010003)", "pk_fma denormal q (expect 14001600)",
                            "pk_fma magic exponent (expect 38003e00)", "pk_fma clamp -3 / 7 (expect 00003c00)", "pk_fma clamp .25 / NaN (expect 3400 0000?)", "pk_maximum3 (expect 4200 7e00)",
                            "pk_max_f16 with NaN (expect 40003800)", "pk_minimum3 (expect 3c00fc00)", "pk_add neg (expect 0000b400)", "dot4 sign bytes (expect 00000480)",
                            "pk_fma overflow (expect 7c007c00)", "cvt_pkrtz 1e9 / -1e-9 (expect 80007bff)", "MODE[11:0]"};
    for (int i = 0; i < 15; ++i) printf("func %-45s : %08x\n", what[i], h[i]);
    return 0;
}

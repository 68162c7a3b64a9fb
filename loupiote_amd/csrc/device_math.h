// device_math.h — gfx950 device-side arithmetic of the integrator (SPEC.md §3-§12).
// Every expression is written with the parenthesisation SPEC.md states; the TU is
// compiled with -ffp-contract=off so nothing is fused except the explicit fmaf()s.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lptd {

#define LPT_PI 3.14159265358979323846f
#define LPT_INV_PI 0.31830988618379067154f
#define LPT_INV_2PI 0.15915494309189533577f
#define LPT_HALF_PI 1.57079632679489661923f
#define LPT_T_INF 1.0e30f
#define LPT_TAG_RAYGEN 0x52415947u
#define LPT_TAG_SHADE 0u
#define LPT_MIN_ROUGHNESS 0.045f
#define LPT_MIN_NOV 1.0e-4f
#define LPT_LIGHT_BIT 0x80000000u

struct f3 { float x, y, z; };
__host__ __device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 neg(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float max2(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float min2(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return min2(max2(x, lo), hi); }
__device__ __forceinline__ f3 normalize(f3 a) {
    float l2 = dot(a, a);
    if (!(l2 > 0.0f)) return mk3(0.0f, 0.0f, 0.0f);
    float inv = 1.0f / sqrtf(l2);
    return a * inv;
}

// ---- SPEC §4: counter-based RNG -------------------------------------------------
__device__ __forceinline__ uint32_t pcg_hash(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
struct Rng { uint32_t state; };
__device__ __forceinline__ uint32_t stage_seed(uint32_t user_seed, uint32_t seed_counter) { return user_seed * 0x9E3779B9u + seed_counter; }
__device__ __forceinline__ Rng rng_init(uint32_t pixel, uint32_t sseed, uint32_t tag) {
    Rng r;
    r.state = pcg_hash(pixel ^ pcg_hash(sseed ^ tag));
    return r;
}
__device__ __forceinline__ float rng_next(Rng &r) {
    r.state = r.state * 747796405u + 2891336453u;
    uint32_t s = r.state;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    w = (w >> 22u) ^ w;
    return (float)(w >> 8) * 5.9604644775390625e-8f;
}

// ---- SPEC §5: polynomial approximations ------------------------------------------
__device__ __forceinline__ void sincos2pi(float u, float &s, float &c) {
    float q = u * 4.0f;
    int k = (int)q;
    float f = q - (float)k;
    k &= 3;
    float x = f * LPT_HALF_PI;
    float x2 = x * x;
    float ps = fmaf(x2, 2.7557319223985893e-6f, -1.984126984126984e-4f);
    ps = fmaf(x2, ps, 8.333333333333333e-3f);
    ps = fmaf(x2, ps, -1.6666666666666666e-1f);
    ps = fmaf(x2, ps, 1.0f);
    float sn = x * ps;
    float pc = fmaf(x2, -2.755731922398589e-7f, 2.48015873015873e-5f);
    pc = fmaf(x2, pc, -1.3888888888888889e-3f);
    pc = fmaf(x2, pc, 4.1666666666666664e-2f);
    pc = fmaf(x2, pc, -0.5f);
    float cs = fmaf(x2, pc, 1.0f);
    if (k == 0) { s = sn; c = cs; }
    else if (k == 1) { s = cs; c = -sn; }
    else if (k == 2) { s = -sn; c = -cs; }
    else { s = -cs; c = sn; }
}
__device__ __forceinline__ float atan2_approx(float y, float x) {
    float ax = fabsf(x), ay = fabsf(y);
    float mx = max2(ax, ay), mn = min2(ax, ay);
    if (!(mx > 0.0f)) return 0.0f;
    float a = mn / mx;
    float s = a * a;
    float r = fmaf(s, -0.0464964749f, 0.15931422f);
    r = fmaf(s, r, -0.327622764f);
    r = fmaf(r * s, a, a);
    if (ay > ax) r = LPT_HALF_PI - r;
    if (x < 0.0f) r = LPT_PI - r;
    if (y < 0.0f) r = -r;
    return r;
}
__device__ __forceinline__ float acos_approx(float x) {
    float a = fabsf(x);
    if (a > 1.0f) a = 1.0f;
    float p = fmaf(a, -0.0187293f, 0.0742610f);
    p = fmaf(a, p, -0.2121144f);
    p = fmaf(a, p, 1.5707288f);
    float r = sqrtf(1.0f - a) * p;
    return x < 0.0f ? LPT_PI - r : r;
}
__device__ __forceinline__ void onb(f3 n, f3 &t, f3 &b) {
    float sign = copysignf(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float bb = n.x * n.y * a;
    t = mk3(1.0f + sign * n.x * n.x * a, sign * bb, -sign * n.x);
    b = mk3(bb, sign + n.y * n.y * a, -n.y);
}

// ---- SPEC §10: BSDF ---------------------------------------------------------------
struct Surface { f3 diff, f0; float alpha, a2; };
__device__ __forceinline__ Surface make_surface(f3 base, float roughness, float metallic) {
    Surface s;
    float r = clampf(roughness, LPT_MIN_ROUGHNESS, 1.0f);
    float m = clampf(metallic, 0.0f, 1.0f);
    s.alpha = r * r;
    s.a2 = s.alpha * s.alpha;
    float om = 1.0f - m;
    s.diff = mk3(base.x * om, base.y * om, base.z * om);
    s.f0 = mk3(0.04f * om + base.x * m, 0.04f * om + base.y * m, 0.04f * om + base.z * m);
    return s;
}
__device__ __forceinline__ float pow5(float m) { float m2 = m * m; return (m2 * m2) * m; }
__device__ __forceinline__ float lum(f3 c) { return (0.2126f * c.x + 0.7152f * c.y) + 0.0722f * c.z; }
__device__ __forceinline__ float spec_probability(const Surface &s, float NoV) {
    float fc = pow5(1.0f - NoV);
    f3 Fv = mk3(s.f0.x + (1.0f - s.f0.x) * fc, s.f0.y + (1.0f - s.f0.y) * fc, s.f0.z + (1.0f - s.f0.z) * fc);
    float ws = lum(Fv);
    float wd = lum(s.diff) * (1.0f - ws);
    if (!(wd > 0.0f)) return 1.0f;
    return clampf(ws / (ws + wd), 0.1f, 0.9f);
}
__device__ __forceinline__ void bsdf_eval(const Surface &s, f3 N, f3 Ng, f3 V, float NoV, float pspec, f3 L, f3 &f, float &pdf) {
    f = mk3(0.0f, 0.0f, 0.0f);
    pdf = 0.0f;
    float NoL = dot(N, L);
    if (!(NoL > 0.0f) || !(dot(Ng, L) > 0.0f)) return;
    f3 H = normalize(V + L);
    float NoH = max2(dot(N, H), 0.0f);
    float VoH = max2(dot(V, H), 0.0f);
    float dd = (NoH * NoH) * (s.a2 - 1.0f) + 1.0f;
    float D = s.a2 / (LPT_PI * (dd * dd));
    float k = s.alpha * 0.5f;
    float gl = NoL * (1.0f - k) + k, gv = NoV * (1.0f - k) + k;
    float vis = 1.0f / (4.0f * (gl * gv));
    float fc = pow5(1.0f - VoH);
    f3 F = mk3(s.f0.x + (1.0f - s.f0.x) * fc, s.f0.y + (1.0f - s.f0.y) * fc, s.f0.z + (1.0f - s.f0.z) * fc);
    float dv = D * vis;
    f.x = (s.diff.x * LPT_INV_PI) * (1.0f - F.x) + dv * F.x;
    f.y = (s.diff.y * LPT_INV_PI) * (1.0f - F.y) + dv * F.y;
    f.z = (s.diff.z * LPT_INV_PI) * (1.0f - F.z) + dv * F.z;
    float pdf_d = NoL * LPT_INV_PI;
    float pdf_s = VoH > 0.0f ? (D * NoH) / (4.0f * VoH) : 0.0f;
    pdf = pspec * pdf_s + (1.0f - pspec) * pdf_d;
}
__device__ __forceinline__ bool bsdf_sample(const Surface &s, f3 N, f3 Ng, f3 V, float NoV, float pspec, float r3, float r4, float r5,
                                            f3 &Lout, f3 &weight, float &pdf) {
    f3 T, B;
    onb(N, T, B);
    float sn, cs;
    sincos2pi(r5, sn, cs);
    f3 L;
    if (r3 < pspec) {
        float cos2 = (1.0f - r4) / (1.0f + (s.a2 - 1.0f) * r4);
        float ct = sqrtf(cos2);
        float st = sqrtf(max2(0.0f, 1.0f - cos2));
        float hx = st * cs, hy = st * sn;
        f3 H = ((T * hx) + (B * hy)) + (N * ct);
        float vh2 = 2.0f * dot(V, H);
        L = mk3(vh2 * H.x - V.x, vh2 * H.y - V.y, vh2 * H.z - V.z);
    } else {
        float r = sqrtf(r4);
        float lx = r * cs, ly = r * sn, lz = sqrtf(max2(0.0f, 1.0f - r4));
        L = ((T * lx) + (B * ly)) + (N * lz);
    }
    L = normalize(L);
    f3 f;
    float p;
    bsdf_eval(s, N, Ng, V, NoV, pspec, L, f, p);
    if (!(p > 0.0f)) return false;
    float NoL = dot(N, L);
    float w = NoL / p;
    weight = mk3(f.x * w, f.y * w, f.z * w);
    Lout = L;
    pdf = p;
    return true;
}

}  // namespace lptd

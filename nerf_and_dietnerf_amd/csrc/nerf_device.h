// nerf_device.h -- device helpers shared by the kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nerf {

// ------------------------------------------------------------------------------------------------
// sin(theta + h*pi/2) with 3-constant Cody-Waite reduction (FMA) and the classic single-precision
// minimax polynomials on [-pi/4, pi/4]; |theta| < ~1e5.  h = 0 -> sin, h = 1 -> cos.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sin_shifted(float th, int h) {
    const float n = rintf(th * 0.6366197466850281f);
    float r = fmaf(-n, 1.5707963705062866f, th);
    r = fmaf(-n, -4.371138828673793e-08f, r);
    r = fmaf(-n, -1.7151245100058819e-15f, r);
    const int q = (int)n + h;
    const float r2 = r * r;
    float sp = fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f);
    sp = fmaf(sp, r2, -1.6666654611e-1f);
    sp = fmaf(sp * r2, r, r);
    float cp = fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f);
    cp = fmaf(cp, r2, 4.166664568298827e-2f);
    cp = fmaf(cp * r2, r2, fmaf(-0.5f, r2, 1.0f));
    float v = (q & 1) ? cp : sp;
    return (q & 2) ? -v : v;
}

// out[k] = sin(2^k theta + h pi/2), k < L, for the SINGLE-PASS FP16 kernels only: one full evaluation of sin and cos at
// k = 0 (same reduction and polynomials as sin_shifted) and the angle-doubling recurrence above it,
// s' = 2 s c, c' = 1 - 2 s^2 -- 3 operations a level instead of ~22.  The error doubles per level (~2e-6 at k = 4),
// two orders below the fp16 rounding the encodings get in those kernels; the fp32-class modes keep sin_shifted.
template <int L>
__device__ __forceinline__ void sin_ladder_fp16_modes(float th, int h, float* out) {
    const float n = rintf(th * 0.6366197466850281f);
    float r = fmaf(-n, 1.5707963705062866f, th);
    r = fmaf(-n, -4.371138828673793e-08f, r);
    r = fmaf(-n, -1.7151245100058819e-15f, r);
    const int q = (int)n;
    const float r2 = r * r;
    float sp = fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f);
    sp = fmaf(sp, r2, -1.6666654611e-1f);
    sp = fmaf(sp * r2, r, r);
    float cp = fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f);
    cp = fmaf(cp, r2, 4.166664568298827e-2f);
    cp = fmaf(cp * r2, r2, fmaf(-0.5f, r2, 1.0f));
    float s = (q & 1) ? cp : sp;
    float c = (q & 1) ? -sp : cp;
    if (q & 2) { s = -s; c = -c; }
    out[0] = h ? c : s;
#pragma unroll
    for (int k = 1; k < L; ++k) {
        const float t = s + s;
        const float s2 = t * c;
        c = fmaf(-t, s, 1.0f);
        s = s2;
        out[k] = h ? c : s;
    }
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG: counter = (ray_lo, ray_hi, sample/4, stream), key = seed.
// Keyed by the GLOBAL ray index so draws do not depend on batching or on the GPU a ray lands on.
// stream 0 = stratified jitter (get_z_values, UtilsCV.py:580), 1 = inverse-CDF draws (:516).
// Mirrors oracle/nerf_oracle.py:philox_uniform bit for bit.
// ------------------------------------------------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ inline u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)c.x * 0xD2511F53u;
        const uint64_t p1 = (uint64_t)c.z * 0xCD9E8D57u;
        u32x4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// uint32 -> fp32 in [0,1) with 23 random mantissa bits
__device__ __forceinline__ float bits_to_uniform(uint32_t x) {
    return __uint_as_float((x >> 9) | 0x3F800000u) - 1.0f;
}

__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t ray, int sample, uint32_t stream) {
    u32x4 c;
    c.x = (uint32_t)ray; c.y = (uint32_t)(ray >> 32); c.z = (uint32_t)(sample >> 2); c.w = stream;
    const u32x4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const int e = sample & 3;
    return bits_to_uniform(e == 0 ? r.x : e == 1 ? r.y : e == 2 ? r.z : r.w);
}

}  // namespace nerf

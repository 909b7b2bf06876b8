// mlp_f16_frag.h -- pieces shared by the fused fp16-core kernels (mlp_f16x3.hip: forward; mlp_bwd_f16x3.hip: the
// training path's fused data-gradient chain): fragment types, the asm LDS fragment reads with their counted waits,
// the hi/lo split, and the geometry of the 32 KiB-chunk weight ring.  gfx950 only.
#pragma once
#include "mlp_common.h"

namespace nerf {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef uint32_t frag4 __attribute__((ext_vector_type(4)));   // one fp16 fragment = 4 dwords of 2 halfs

// asm LDS read of one fragment into an AGPR quad, and the counted wait that retires it (see layer_body_h)
template <int OFF>
__device__ __forceinline__ void lds_read_frag_asm(f32x4& dst, uint32_t base) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(base), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_frag_asm(f32x4& reg) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+a"(reg) : "n"(N) : "memory");
}

__device__ __forceinline__ uint32_t pack_h2(float a, float b) {   // RNE; v_cvt_pk_f16_f32
    const h2 t = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, t);
}

// ---- ring geometry of the fp16 streams (quads of 1 KiB = one fp16 A fragment: 64 lanes x 8 halfs) ----
constexpr int kHCQ = 32;          // quads per chunk (32 KiB)
constexpr int kHRing = 4;         // ring slots (128 KiB)
constexpr int kHChunkBytes = kHCQ * kQuadBytes;
static_assert(kHRing * kHChunkBytes == kRingBytes, "fp16 ring must fill the shared LDS carve");

// fp32 -> (hi, lo) with hi = the top 11 significand bits (exact in fp16) and lo = y - hi (exact in
// fp32, then rounded to fp16): |lo| <= 2^-10 |y|, total representation error <= 2^-21 |y|.  Costs two
// plain VALU ops (v_and, v_sub) instead of a v_cvt round trip: beside the fp16 MFMA, conversions and
// moves are "8-cycle" instructions, plain arithmetic is nearly free (tools/microbench/valu_cost_f16.hip).
__device__ __forceinline__ void split_trunc(float y, float& hi_f, float& lo_f) {
    hi_f = __uint_as_float(__float_as_uint(y) & 0xFFFFE000u);
    lo_f = y - hi_f;
}


// activation fragment order shared by forward and backward: element e of lane half h of k-step n (n = 2t + s) is
// feature 32t + 16s + 8(e>>2) + 4h + (e&3) -- the accumulator-as-operand order of the 32x32 C/D layout
__host__ __device__ inline int frag_feature(int n, int e, int h) {
    const int t = n >> 1, s = n & 1;
    return 32 * t + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3);
}

}  // namespace nerf

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
#ifdef NERF_DIAG_NO_LDSWAIT    // timing-only diagnostic (wrong results): the fragment is used without waiting for it
    asm volatile("" : "+a"(reg) : : "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+a"(reg) : "n"(N) : "memory");
#endif
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


// ================= geometry of the forward streams and constant blocks (mlp_f16x3.hip, mlp_f16_2t.hip) =================
// ---- stream geometry (quads of 1 KiB = one fp16 A fragment: 64 lanes x 8 halfs) ----
constexpr int kHStepsPE = 3;      // 33 inputs -> 48 slots
constexpr int kHStepsHid = 16;
constexpr int kHStepsDir = 2;     // 24 inputs -> 32 slots
constexpr int kHQpuPE = 2 * kHStepsPE;                       // hi + lo fragment per k-step
constexpr int kHQpuHid = 2 * kHStepsHid;
constexpr int kHQpuSkip = kHQpuPE + kHQpuHid;
constexpr int kHQpuLast = kHQpuHid + 2 * kHStepsDir;
constexpr int kHTilesLast = 5;    // 4 x 32 features of layer 8 + the sigma row
constexpr int kHChunksPE = (8 * kHQpuPE + kHCQ - 1) / kHCQ;          // 2
constexpr int kHChunksHid = (8 * kHQpuHid) / kHCQ;                   // 8
constexpr int kHChunksSkip = (8 * kHQpuSkip + kHCQ - 1) / kHCQ;      // 10
constexpr int kHChunksLast = (kHTilesLast * kHQpuLast + kHCQ - 1) / kHCQ;   // 6
constexpr int kHStreamChunks = kHChunksPE + 6 * kHChunksHid + kHChunksSkip + kHChunksLast;   // 66
// single-pass mode: its own stream with the hi fragments only (one quad per k-step)
constexpr int kFChunksPE = (8 * kHStepsPE + kHCQ - 1) / kHCQ;                          // 1
constexpr int kFChunksHid = (8 * kHStepsHid) / kHCQ;                                   // 4
constexpr int kFChunksSkip = (8 * (kHStepsPE + kHStepsHid) + kHCQ - 1) / kHCQ;         // 5
constexpr int kFChunksLast = (kHTilesLast * (kHStepsHid + kHStepsDir) + kHCQ - 1) / kHCQ;   // 3
constexpr int kFStreamChunks = kFChunksPE + 6 * kFChunksHid + kFChunksSkip + kFChunksLast;  // 33
static_assert(kFStreamChunks * (size_t)kHChunkBytes == kStreamBytesF16Hi, "hi-only stream size mismatch");
// constant region (floats)
constexpr int kHConstBias = 0;        // 8 x 256
constexpr int kHConstBias8 = 2048;    // 128
constexpr int kHConstBiasSig = 2176;  // 32: row 0 = sigma bias
constexpr int kHConstWrgb = 2208;     // [3][128]
constexpr int kHConstBHead = 2592;    // b_r, b_g, b_b, (unused)
constexpr int kHConstFloats = 2608;

static_assert(kHStreamChunks * (size_t)kHChunkBytes == kStreamBytesF16, "stream size mismatch");
// xyz-only network (n_angles_for_model = 0, src/NeRF.py:248-288): ... -> h8 -> [sigma | dense 256] -> dense 128 -> rgb.
// Its tail replaces BODY_LAST by two bodies: BODY_HIDSIG (the extra 256-wide layer with the sigma row as a LEADING 9th
// tile: sigma reads h8, this body's input) and BODY_LAST0 (256 -> 128, no direction k-steps, no sigma tile).
constexpr int kXChunksHidSig = (9 * kHQpuHid) / kHCQ;                   // 9
constexpr int kXChunksLast0 = (4 * kHQpuHid) / kHCQ;                    // 4
constexpr int kXStreamChunks = kHChunksPE + 6 * kHChunksHid + kHChunksSkip + kXChunksHidSig + kXChunksLast0;   // 73
constexpr int kXFChunksHidSig = (9 * kHStepsHid + kHCQ - 1) / kHCQ;     // 5
constexpr int kXFChunksLast0 = (4 * kHStepsHid) / kHCQ;                 // 2
constexpr int kXFStreamChunks = kFChunksPE + 6 * kFChunksHid + kFChunksSkip + kXFChunksHidSig + kXFChunksLast0;   // 37
static_assert(kXStreamChunks * (size_t)kHChunkBytes == kStreamBytesF16Xyz, "xyz-only stream size mismatch");
static_assert(kXFStreamChunks * (size_t)kHChunkBytes == kStreamBytesF16HiXyz, "xyz-only hi stream size mismatch");
// constant region of the xyz-only variant (floats): 8 x 256 biases as above, then
constexpr int kXConstBiasSig = 2048;  // 32: row 0 = sigma bias (the leading tile of BODY_HIDSIG)
constexpr int kXConstBias8 = 2080;    // 256: the extra hidden layer
constexpr int kXConstBias9 = 2336;    // 128
constexpr int kXConstWrgb = 2464;     // [3][128]
constexpr int kXConstBHead = 2848;    // b_r, b_g, b_b, (unused)
constexpr int kXConstWsig = 2864;     // [256] the sigma head's weights as floats (the trainer's backward adds its rank-1 term on the VALU)
constexpr int kXConstFloats = 3120;
static_assert(kXConstFloats <= kConstFloats, "xyz-only constants must fit the shared LDS carve");
enum { BODY_HIDSIG = 4, BODY_LAST0 = 5 };
static_assert(kHConstFloats <= kConstFloats, "f16x3 constants must fit the shared LDS carve");

// Stash / gradient-buffer stores of the fused trainer kernels: streamed out once, read back gigabytes later by the
// weight-gradient GEMMs -- non-temporal (`global_store ... nt`): the lines do not displace the weight stream in L2
// (A/B on one device: 4.60 -> 4.48 ms per mixed_float16 step, 9.10 -> 8.97 ms under the fp32 policy; -DNERF_STASH_NO_NT
// builds the plain stores back).
template <class T>
__device__ __forceinline__ void stream_store(T* p, const T& v) {
#ifndef NERF_STASH_NO_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void stream_store(uint2* p, const uint2& v) {     // (the builtin wants a clang vector type)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 w;
    w[0] = v.x; w[1] = v.y;
    stream_store(reinterpret_cast<u32x2*>(p), w);
}

// LeakyReLU' record shared by the stash forward (writer) and the fused backward (reader): per lane and layer four 32-bit
// words, word ut >> 1 for output tiles ut, ut + 1; the epilogue handles a tile's 16 accumulator registers as 8 pairs in
// order and pushes each pair's two fp16 SIGN bits (bits 15 and 31 of the packed pair) into its word from the top, so
// after a word's 16 pushes pair k (k = 8 (ut & 1) + r / 2) sits at bits k (even register) and 16 + k (odd register).
// A set bit = negative activation = LeakyReLU' is alpha.  (An activation of exactly +0 counts as positive; the reference's
// LeakyReLU gradient takes alpha there -- a measure-zero difference that only an all-zero weight row can produce.)
__device__ __forceinline__ uint32_t mask_push(uint32_t word, uint32_t packed_pair) {
    return (packed_pair & 0x80008000u) | (word >> 1);
}
__host__ __device__ constexpr int mask_bit(int ut, int r) { return 8 * (ut & 1) + (r >> 1) + 16 * (r & 1); }

// activation fragment order shared by forward and backward: element e of lane half h of k-step n (n = 2t + s) is
// feature 32t + 16s + 8(e>>2) + 4h + (e&3) -- the accumulator-as-operand order of the 32x32 C/D layout
__host__ __device__ inline int frag_feature(int n, int e, int h) {
    const int t = n >> 1, s = n & 1;
    return 32 * t + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3);
}

}  // namespace nerf

// mlp_bwd_f16x3.hip -- the training path's fused data-gradient chain: one kernel walks a 128-row tile of samples
// backwards through the Dense stack (layer 8 .. layer 1 [, the encoding rows of layers 4 and 0]) with the gradient
// kept on the lane from layer to layer, exactly as mlp_f16x3.hip walks it forwards.
//
// Reference: the backward of NeRF.train_step's tape (src/NeRF.py:136-167) through the Keras model of
// src/NeRF.py:316-339 -- for every sample row
//     G9  = (W9 . d_rgb) * LeakyReLU'(h9)                                       (rgb head, VALU)
//     D7  = (W8[hidden rows] . G9 + W10[hidden rows] * d_sigma) * LeakyReLU'(h8)
//     D_l-1 = (W_l . D_l) * LeakyReLU'(h_l)          l = 7 .. 1   (layer 4: its 256 hidden rows)
//     dXa = W4[xyz rows] . D4,  dXb = W0 . D0        (gradient w.r.t. the xyz encoding; only for the sampler term)
// where D_l is the gradient w.r.t. the pre-activation of layer l.  It replaces the layer-by-layer gemm_abt_h launches
// (train_kernels.hip) whose every operand made a round trip through HBM.
//
// Same skeleton as the forward kernel: one wave per SIMD, 32 samples per wave with the sample on the lane,
// D^T = W . D_next^T on v_mfma_f32_32x32x16_f16 in three passes over hi/lo split operands (fp32-class results), the
// accumulator -- after the LeakyReLU' mask -- re-packed as the next layer's B operand, weights (here the
// NON-transposed kernels, one 32-row tile of input features per accumulator) streamed L2 -> LDS ring by LDS-DMA.
// What is new:
//   * LeakyReLU' comes from a 1-bit-per-activation record written by the stash forward (mlp_f16x3.hip): the lane loads
//     one 16-byte word per layer whose bit 16 ut + r is the mask of its accumulator register r of tile ut;
//   * gradients are tiny and grow or shrink from layer to layer, fp16 has 5 exponent bits: every SAMPLE (= lane) carries
//     its own power-of-two scale, renewed per layer from the largest entry of that sample's operand, so that the packed
//     operand peaks between 2^5 and 2^12 for every row independently (the layer-wise GEMM could only scale whole
//     buffers); scaling a column of B scales that column of the product, so the epilogue undoes it exactly;
//   * every D_l is also written in fp32 (true scale) for the weight-gradient GEMMs, with max|D_l| for their scaling.
#include "mlp_f16_frag.h"

#include <string.h>

namespace nerf {

// ---- ring geometry.  This kernel (like the stash forward) also STORES 8.5 KB per sample row, and gfx9 counts loads and
// stores on one vmcnt, so the chunk synchronisation bounds how long a store may stay unacknowledged.  A finer ring
// (NERF_BWD_CQ=16: 16 KiB chunks in 8 slots, 20 operations of slack instead of 8) was built to widen that window and
// measured 1.7 % SLOWER per training step (12.05 vs 11.84 ms): the window is not what bounds the stores
// (L2 bandwidth shared with the weight stream is).  The render kernels' 32 KiB x 4 stays.
//
// The single-pass kernels (mixed_float16 policy) carry half the operand registers of the 3-pass ones and fit 256, so two
// workgroups per CU are possible, each on its own 64 KiB ring of 16 KiB chunks (NERF_BWD_FAST_OCC=2): built and measured
// 2 % SLOWER per mixed_float16 step (4.26 vs 4.16 ms, same device) -- a second wave per SIMD hides waits, and waits are not
// what this kernel is short of: its vector instructions (10 per MFMA) and its MFMAs do not overlap, and two waves share
// one VALU.  The default stays one workgroup per CU.
#ifndef NERF_BWD_CQ
#define NERF_BWD_CQ 32
#endif
#ifndef NERF_BWD_FAST_OCC
#define NERF_BWD_FAST_OCC 1
#endif
template <bool FAST>
struct BGeo {
    static constexpr int Occ = FAST ? NERF_BWD_FAST_OCC : 1;                 // workgroups per CU
    static constexpr int CQ = Occ == 2 ? 16 : NERF_BWD_CQ;                    // quads per chunk
    static constexpr int RingBytes = kRingBytes / Occ;
    static constexpr int ChunkBytes = CQ * kQuadBytes;
    static constexpr int Ring = RingBytes / ChunkBytes;
    static constexpr int LdsConst = kLdsRing + RingBytes;                     // the constant block follows the ring
    static constexpr int LdsGmax = LdsConst + (kConstFloats - 16) * 4;        // 9 (xyz-only network: 10) x uint32: max|D| bits per gradient buffer
    static constexpr int LdsTotal = RingBytes + kConstBytes;
    static constexpr int NAcc = Occ == 2 ? 2 : 4;                             // rotating accumulator tiles
    static constexpr int Pf = FAST ? (Occ == 2 ? 4 : 8) : 4;                  // LDS fragment reads in flight
    static_assert(Ring * ChunkBytes == RingBytes && (Ring & (Ring - 1)) == 0, "ring must fill its LDS carve");
    static_assert(Occ * LdsTotal <= 160 * 1024, "workgroups of one CU must share 160 KiB");
};

// ---- stream geometry: bodies in execution order, each padded to whole chunks ----
constexpr int kBStepsHead = 9;                 // 128 features of G9 (8 k-steps) + 1 k-step carrying d_sigma
constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
// chunks of the stream of one kernel variant (3-pass: hi + lo fragment per k-step; single-pass: hi only)
template <bool FAST>
constexpr int bwd_stream_chunks(bool dx, bool xyz) {
    constexpr int CQ = BGeo<FAST>::CQ, T = FAST ? 1 : 2;
    const int head = cdiv(8 * T * kBStepsHead, CQ), hid = cdiv(8 * T * 16, CQ);
    const int hidx = cdiv(10 * T * 16, CQ), enc = cdiv(2 * T * 16, CQ);   // layer 4 with its two encoding tiles; layer 0's encoding rows
    return head + (dx ? 6 * hid + hidx + enc : 7 * hid) + (xyz ? hid : 0);   // the xyz-only network has one more 256 x 256 body
}
static_assert((size_t)bwd_stream_chunks<false>(true, true) * BGeo<false>::ChunkBytes <= kBwdStreamBytes &&
              (size_t)bwd_stream_chunks<true>(true, true) * BGeo<true>::ChunkBytes <= kBwdStreamBytes,
              "backward stream does not fit its buffer");

constexpr int kBConstWrgb = 2208;     // the forward kernel's constant block is reused: [3][128] rgb head weights
constexpr int kBConstFloats = 2608;
static_assert(kBConstFloats + 16 <= kConstFloats && kXConstFloats + 16 <= kConstFloats, "gmax slots must fit the shared LDS carve");

enum { BW_HEAD = 0, BW_HID = 1, BW_XYZ = 2 };

// max|D| slot k of the workgroup: a plain ds_max_u32 from every lane (one address: the LDS takes the 64 lanes one after
// the other, in ITS cycles; the wave does not wait for a no-return atomic).  Written as atomicMax(), hipcc's atomic
// optimizer turns it into a scalar loop over the 64 lanes (s_ff1 / v_readlane / s_max, ~50 cycles a turn) in the middle of
// the MFMA chain, once per body: a fifth of the single-pass kernel's time.  (The dynamic LDS segment starts at address 0,
// as the asm fragment reads assume as well.)
template <bool FAST>
__device__ __forceinline__ void lds_gmax_update(int slot, uint32_t bits) {
    asm volatile("ds_max_u32 %0, %1" : : "v"((uint32_t)(BGeo<FAST>::LdsGmax + 4 * slot)), "v"(bits) : "memory");
}

struct BwdLane {          // per-lane (= per-sample) scale state
    float inv_sig;        // 1 / (scale of the current body's input operand)
    float rho;            // power of two applied to the current body's outputs when they are packed
    float inv_prev, rho_prev;   // the same two numbers of the previous body (for its pending last tile)
    float mrun;           // running max |packed value| of the operand being produced
};

__device__ __forceinline__ float pow2_to_peak(float m) {
    // power of two r with m * r in (2^5, 2^6]  (1 for m == 0): exponent arithmetic on the bits
    const uint32_t b = __float_as_uint(m);
    int e = 259 - (int)(b >> 23);
    e = e < 1 ? 1 : e > 254 ? 254 : e;
    return b == 0u ? 1.0f : __uint_as_float((uint32_t)e << 23);
}
__device__ __forceinline__ float pow2_inverse(float r) { return __uint_as_float(0x7F000000u - __float_as_uint(r)); }
__device__ __forceinline__ float max_with_other_half(float v) {   // max over the two lanes (j, 0) and (j, 1) of a sample
    return fmaxf(v, __shfl_xor(v, 32));
}
// `neg` or `pos` by bit BIT of the LeakyReLU' record w (set = negative activation): the bit sign-extended to all-ones /
// zero, then a bitfield insert -- two plain VALU ops, in asm because hipcc turns the C form into and + compare + select
template <int BIT>
__device__ __forceinline__ int mask_ones(uint32_t w) {
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(w), "n"(BIT));
    return m;
}
__device__ __forceinline__ float mask_select(int m, float neg, float pos) {
    float f;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(f) : "v"(m), "v"(neg), "v"(pos));
    return f;
}

// One transposed layer.  Tiles 0..NX-1 produce rows of the encoding gradient (true scale, no mask, not re-packed), the
// NH = 8 (0 for BW_XYZ) tiles after them the masked gradient of the 256 hidden features.  As in the forward kernel the
// epilogue of a tile is dealt out over the NEXT tile's k-steps and the last hidden tile is finished by the next body
// (PEND).  copy_tail: the previous body staged its outputs in nh/nl (an in-place BW_HID body), so fragments 12..13
// still have to move to xh/xl; after BW_HEAD (which writes xh/xl directly) they must not.
// FAST = the mixed_float16 policy's arithmetic: one fp16 MFMA pass per product (hi fragments only, their own stream),
// the masked gradient rounded (RNE) to fp16 as the next operand; the per-sample scales work as before.
// SIG / SIGP (the xyz-only network, src/NeRF.py:248-288: sigma = Dense(1)(h8) while the colour branch runs on through another
// 256-wide layer): this body's / the previous body's hidden tiles are dL/dh8 = W8 . D8b + w_sigma * d_sigma -- the rank-1
// term is added to the accumulator on the VALU (w_sigma from the constant block, d_sigma in the accumulator's scale).
template <int KIND, int NX, bool PEND, bool FAST, bool SIG = false, bool SIGP = false>
__device__ __forceinline__ void bwd_body(Pipe& p, uint32_t lane16, float alpha, BwdLane& L, bool copy_tail,
                                         float* d_prev, float* d_cur, float* dx_cur, const frag4& mk_prev,
                                         const frag4& mk_cur, int gslot_prev, f32x16 (&accs)[BGeo<FAST>::NAcc], frag4 (&xh)[16],
                                         frag4 (&xl)[16], frag4 (&nh)[14], frag4 (&nl)[14], float d_sigma = 0.f,
                                         uint32_t cb_h = 0u) {
#ifndef NERF_BWD_EXTRA_FAST
#define NERF_BWD_EXTRA_FAST 0
#endif
#ifndef NERF_BWD_EXTRA_3P
#define NERF_BWD_EXTRA_3P 0
#endif
    constexpr int kBwdExtra = FAST ? NERF_BWD_EXTRA_FAST : NERF_BWD_EXTRA_3P;
    constexpr int NH = KIND == BW_XYZ ? 0 : 8;
    constexpr int NU = NX + NH;
    constexpr int NSTEP = KIND == BW_HEAD ? kBStepsHead : 16;
    constexpr int TPS = FAST ? 1 : 2;
    constexpr int QPU = TPS * NSTEP;
    constexpr int NQ = NU * QPU;
    using G = BGeo<FAST>;
    constexpr int kBCQ = G::CQ, kBRing = G::Ring, kBChunkBytes = G::ChunkBytes, kPf = G::Pf, NA = G::NAcc;
    f32x4 pf[kPf];
    const int ck0 = p.ck;
    uint32_t rdbase[2];
    rdbase[0] = lane16 + (uint32_t)((ck0 + 0) & (kBRing - 1)) * kBChunkBytes;
    rdbase[1] = lane16 + (uint32_t)((ck0 + 1) & (kBRing - 1)) * kBChunkBytes;
    auto issue_read = [&](auto qc) {
        constexpr int Qa = decltype(qc)::value;
        lds_read_frag_asm<(Qa % kBCQ) * kQuadBytes>(pf[Qa % kPf], rdbase[(Qa / kBCQ) & 1]);
    };
    static_for<0, kPf>([&](auto ic) {
        if constexpr (decltype(ic)::value < NQ) issue_read(ic);
    });

    float q0 = 0.f, q1 = 0.f, q2 = 0.f;      // true-scale values waiting for the 4th of their float4
    float pc = 0.f;                          // packed-scale value waiting for its pair
    (void)q0; (void)q1; (void)q2; (void)pc;
    // true-scale store of registers r-3..r (r & 3 == 3) of a tile whose 32 rows start at column c0 of `base`
    // fragc: `base` is a fragment-major buffer (frag_layout.h::frag_index: a feature offset c, c % 8 == 0, is the
    // element offset 32 c from the lane's base) -- D and dx both are
    auto store4 = [&](float* base, int c0, int r, float v, auto fragc) {
        f32x4 o;
        o[0] = q0; o[1] = q1; o[2] = q2; o[3] = v;
        stream_store(reinterpret_cast<f32x4*>(base + (decltype(fragc)::value ? 32 : 1) * (c0 + 8 * (r >> 2))), o);
    };
    // register r of hidden tile ht: mask, write the true value, scale + split + pack into the next operand
    // alpha folded into the four scales once per body (pinned: left to itself hipcc re-multiplies per value rather than
    // hold the registers)
    auto pinned = [](float v) { asm volatile("" : "+v"(v)); return v; };
    const float ainv_prev = pinned(alpha * L.inv_prev), arho_prev = pinned(alpha * L.rho_prev);
    const float ainv_cur = pinned(alpha * L.inv_sig);
    float arho_cur = pinned(alpha * L.rho);           // renewed with L.rho (PEND bodies fix it at k-step 8 of tile 0)
    // d_sigma in the scale of this body's / the previous body's accumulators (operand scale = 1 / inv)
    const float dsig_cur = SIG ? d_sigma * pow2_inverse(L.inv_sig) : 0.f;
    const float dsig_prev = SIGP ? d_sigma * pow2_inverse(L.inv_prev) : 0.f;
    (void)dsig_cur; (void)dsig_prev;
    // single-pass: the stored D pair is the packed operand pair times inv / rho, a power of two -- one v_pk_mul_f16 per
    // PAIR instead of a second select + multiply per value and a second conversion per pair (the product is exact unless
    // it lands below 2^-14; a ratio beyond the fp16 range becomes Inf / 0: the former means the true D overflowed fp16
    // anyway -- the step is skipped and the loss scale halved -- the latter that |D| < 2^-18 for the whole sample)
    auto ratio_h2 = [](float inv, float rho) {
        const _Float16 r = (_Float16)(inv * pow2_inverse(rho));
        return h2{r, r};
    };
    const h2 rat_prev = ratio_h2(L.inv_prev, L.rho_prev);
    h2 rat_cur = ratio_h2(L.inv_sig, L.rho);          // renewed with L.rho
    (void)rat_prev; (void)rat_cur;
    auto hidden_reg = [&](auto htc, auto rc, float acc_in, const frag4& mk, float inv_s, float rho_s, float ainv_s,
                          float arho_s, h2 rat_s, float* dst, auto to_x, auto sig_sel) {
        constexpr int ht = decltype(htc)::value;
        constexpr int r = decltype(rc)::value;
#if defined(NERF_DIAG_BWD_EPI) && NERF_DIAG_BWD_EPI <= 2   // timing-only diagnostics (wrong results): 2 = the value is moved out of
        if constexpr (FAST) {                               // its accumulator and dropped, 1 = one move per 16 values
            if constexpr (NERF_DIAG_BWD_EPI == 2 || r == 15) asm volatile("" : : "v"(acc_in));
            return;
        }
#endif
        float acc_v = acc_in;
        if constexpr (decltype(sig_sel)::value != 0) {
            // feature 32 ht + 8 (r >> 2) + 4 h + (r & 3) of this lane (cb_h carries the 4 h)
            extern __shared__ __attribute__((aligned(16))) char smem_[];
            const float ws = *reinterpret_cast<const float*>(smem_ + cb_h + (kXConstWsig + 32 * ht + 8 * (r >> 2) + (r & 3)) * 4);
            acc_v = fmaf(ws, decltype(sig_sel)::value == 1 ? dsig_cur : dsig_prev, acc_in);
        }
        // t = acc * LeakyReLU' / scale-in (the true value), pk = acc * LeakyReLU' * scale-out (the next operand): LeakyReLU'
        // is folded into the two power-of-two scales (exact), one selected factor each -- bfe + 2 bfi + 2 mul where
        // and + compare + select + three multiplications were 6 VALU ops per value (single-pass: t comes from the packed
        // pair, see ratio_h2: bfe + bfi + mul per value)
        const int neg = mask_ones<mask_bit(ht, r)>(mk[ht >> 1]);
        float pk;
        if constexpr (!FAST && !kPair16) {
            const float t = acc_v * mask_select(neg, ainv_s, inv_s);
            if constexpr ((r & 3) == 0) q0 = t;
            else if constexpr ((r & 3) == 1) q1 = t;
            else if constexpr ((r & 3) == 2) q2 = t;
            else store4(dst, 32 * ht, r, t, std::true_type{});
        }
        // (pk = t * (rho / inv) would save the second select, but a sample whose gradient underflows has inv = 0 and
        // rho / inv = Inf: 0 * Inf poisoned the weight gradients -- measured, reverted)
        pk = acc_v * mask_select(neg, arho_s, rho_s);
#if !(defined(NERF_DIAG_BWD_EPI) && NERF_DIAG_BWD_EPI == 6)    // 6: timing-only, no running max of the packed operand
        L.mrun = fmaxf(L.mrun, fabsf(pk));
#endif
        if constexpr ((r & 1) == 0) pc = pk;
        else {
            constexpr int n = 2 * ht + (r >> 3), d = (r & 7) >> 1;
            if constexpr (FAST) {
                const uint32_t ph = pack_h2(pc, pk);
#if defined(NERF_DIAG_BWD_EPI) && NERF_DIAG_BWD_EPI == 5    // timing-only: the next operand is NOT rewritten (the value is only kept alive)
                asm volatile("" : : "v"(ph));
#else
                if constexpr (decltype(to_x)::value) xh[n][d] = ph;
                else nh[n][d] = ph;
#endif
                // mixed_float16 policy: D is stored in fp16 (dst points at fp16 rows).  The values carry the loss scale, as
                // the policy's activation gradients do: an overflow becomes Inf here, NaN in the weight gradient, and the
                // LossScaleOptimizer logic skips the step and halves the scale.
                const uint32_t tp = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, ph) * rat_s);
                if constexpr ((r & 3) == 1) q1 = __uint_as_float(tp);
#if defined(NERF_DIAG_BWD_EPI) && NERF_DIAG_BWD_EPI == 3     // timing-only: everything but the store instruction
                else asm volatile("" : : "v"(q1), "v"(tp), "v"(dst));
#else
                else stream_store(reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(dst) + 32 * (32 * ht + 8 * (r >> 2))),
                                  make_uint2(__float_as_uint(q1), tp));
#endif
            } else {
                float h0, l0, h1, l1;
                split_trunc(pc, h0, l0);
                split_trunc(pk, h1, l1);
                const uint32_t ph = pack_h2(h0, h1), pl = pack_h2(l0, l1);
                if constexpr (decltype(to_x)::value) { xh[n][d] = ph; xl[n][d] = pl; }
                else { nh[n][d] = ph; nl[n][d] = pl; }
                // pair16 gradient buffers (nerf_kernels.h::kPair16): D is stored as this very operand pair -- the row's power-of-
                // two scale stays on it, its inverse goes to rs_ptr once per row and buffer -- in the slot of its fp32 value:
                // {hi01, hi23, lo01, lo23} per four features.  No true-scale value is formed at all (a select and a multiply
                // per value less than the fp32 buffers needed).
                if constexpr (kPair16) {
                    if constexpr ((r & 3) == 1) { q0 = __uint_as_float(ph); q1 = __uint_as_float(pl); }
                    else {
                        frag4 o;
                        o[0] = __float_as_uint(q0); o[1] = ph; o[2] = __float_as_uint(q1); o[3] = pl;
                        stream_store(reinterpret_cast<frag4*>(dst + 32 * (32 * ht + 8 * (r >> 2))), o);
                    }
                }
            }
        }
    };
    auto xyz_reg = [&](auto xtc, auto rc, float acc_v) {
        constexpr int xt = decltype(xtc)::value;
        constexpr int r = decltype(rc)::value;
        const float t = acc_v * L.inv_sig;
        if constexpr ((r & 3) == 0) q0 = t;
        else if constexpr ((r & 3) == 1) q1 = t;
        else if constexpr ((r & 3) == 2) q2 = t;
        else store4(dx_cur, 32 * xt, r, t, std::true_type{});
    };

    static_for<0, NU>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        // accumulators rotate so that every body's LAST tile ends in accs[NA - 1]: that is where the next body's pending
        // epilogue (u == 0) looks for it, whatever the tile count
        constexpr int kOff = (NA - NU % NA) % NA;
        f32x16& acc = accs[(u + kOff) % NA];
#if defined(NERF_DIAG_BWD_EPI) && NERF_DIAG_BWD_EPI == 4    // timing-only: the epilogue reads a tile finished TWO tiles ago
        f32x16& prv = accs[(u + kOff + NA - 2) % NA];
#else
        f32x16& prv = accs[u == 0 ? NA - 1 : (u + kOff + NA - 1) % NA];
#endif
        static_for<0, NSTEP>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            f32x4 araw[2];
            static_for<0, TPS>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int Q = u * QPU + TPS * n + t;
                constexpr int qc = Q % kBCQ;
                if constexpr (qc == 0 && Q > 0) p.ck += 1;
                if constexpr (qc == kBCQ / 2) {
                    constexpr int room = (NQ - 1 - Q) / 2;
                    pipe_sync_c<kBCQ, kBRing, (room + 1 < kBCQ / 4 ? room + 1 : kBCQ / 4), kBwdExtra>(p);
                }
                if constexpr (qc > kBCQ / 2 && (qc - kBCQ / 2) % 2 == 0) {
                    constexpr int Qs = Q - (qc - kBCQ / 2);
                    constexpr bool tail = (NQ - 1 - Qs) / 2 + 1 < kBCQ / 4;
                    pipe_piece_c<(qc - kBCQ / 2) / 2, tail>(p);
                }
                lds_wait_frag_asm<(NQ - Q >= kPf ? kPf - 1 : NQ - Q - 1)>(pf[Q % kPf]);
                araw[t] = pf[Q % kPf];
                if constexpr (Q + kPf < NQ) {
                    constexpr int Qn = Q + kPf;
                    if constexpr (Qn % kBCQ == 0)
                        rdbase[(Qn / kBCQ) & 1] = lane16 + (uint32_t)((ck0 + Qn / kBCQ) & (kBRing - 1)) * kBChunkBytes;
                    issue_read(std::integral_constant<int, Qn>{});
                }
            });
            const h8 a_hi = __builtin_bit_cast(h8, araw[0]);
            const h8 a_lo = __builtin_bit_cast(h8, araw[FAST ? 0 : 1]);
            (void)a_lo;
            frag4 bh_, bl_;
            if constexpr (KIND == BW_HEAD) { bh_ = nh[n]; bl_ = nl[n]; }
            else { bh_ = xh[n]; bl_ = xl[n]; }
            const h8 b_hi = __builtin_bit_cast(h8, bh_), b_lo = __builtin_bit_cast(h8, bl_);
            (void)b_lo;
            f32x16 zero;
#pragma unroll
            for (int i = 0; i < 16; ++i) zero[i] = 0.f;
            if constexpr (FAST) {
                if constexpr (n == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, zero, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc, 0, 0, 0);
            } else {
                if constexpr (n == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, zero, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc, 0, 0, 0);
            }

            // ---- deferred epilogues ----
            if constexpr (u == 0 && PEND) {
                if constexpr (n < 8) {           // previous body's hidden tile 7, a register pair per k-step
                    hidden_reg(std::integral_constant<int, 7>{}, std::integral_constant<int, 2 * n>{}, prv[2 * n],
                               mk_prev, L.inv_prev, L.rho_prev, ainv_prev, arho_prev, rat_prev, d_prev, std::true_type{},
                               std::integral_constant<int, SIGP ? 2 : 0>{});
                    hidden_reg(std::integral_constant<int, 7>{}, std::integral_constant<int, 2 * n + 1>{},
                               prv[2 * n + 1], mk_prev, L.inv_prev, L.rho_prev, ainv_prev, arho_prev, rat_prev, d_prev, std::true_type{},
                               std::integral_constant<int, SIGP ? 2 : 0>{});
                }
                if constexpr (n == 8) {
                    // this body's operand is complete: its peak fixes the scale of this body's outputs, and (in true
                    // scale) is max|D| of the buffer the previous body wrote
                    const float m_in = max_with_other_half(L.mrun);
                    // (SIG: this body's outputs also carry w_sigma * d_sigma, whose size is unrelated to the operand's --
                    // with only the operand's peak a ray whose colour gradient vanishes overflowed the fp16 packing)
                    L.rho = pow2_to_peak(SIG ? fmaxf(m_in, fabsf(dsig_cur)) : m_in);
                    arho_cur = pinned(alpha * L.rho);
                    rat_cur = ratio_h2(L.inv_sig, L.rho);
                    L.mrun = 0.f;
                    lds_gmax_update<FAST>(gslot_prev, __float_as_uint(m_in * L.inv_sig));
                    if (copy_tail) { xh[12] = nh[12]; xl[12] = nl[12]; }
                }
                if constexpr (n == 9) { if (copy_tail) { xh[13] = nh[13]; xl[13] = nl[13]; } }
            }
            if constexpr (u > 0) {
                constexpr int pt = u - 1;        // the tile whose epilogue rides on this chain
                if constexpr (pt < NX) {
                    if constexpr (n < 16) xyz_reg(std::integral_constant<int, pt>{}, std::integral_constant<int, n>{}, prv[n]);
                } else {
                    constexpr int ht = pt - NX;
                    if constexpr (NSTEP >= 16) {
                        hidden_reg(std::integral_constant<int, ht>{}, std::integral_constant<int, n>{}, prv[n], mk_cur,
                                   L.inv_sig, L.rho, ainv_cur, arho_cur, rat_cur, d_cur, std::integral_constant<bool, KIND == BW_HEAD>{},
                                   std::integral_constant<int, SIG ? 1 : 0>{});
                    } else if constexpr (n < 8) {
                        hidden_reg(std::integral_constant<int, ht>{}, std::integral_constant<int, 2 * n>{}, prv[2 * n],
                                   mk_cur, L.inv_sig, L.rho, ainv_cur, arho_cur, rat_cur, d_cur, std::integral_constant<bool, KIND == BW_HEAD>{},
                                   std::integral_constant<int, SIG ? 1 : 0>{});
                        hidden_reg(std::integral_constant<int, ht>{}, std::integral_constant<int, 2 * n + 1>{},
                                   prv[2 * n + 1], mk_cur, L.inv_sig, L.rho, ainv_cur, arho_cur, rat_cur, d_cur,
                                   std::integral_constant<bool, KIND == BW_HEAD>{}, std::integral_constant<int, SIG ? 1 : 0>{});
                    }
                }
            }
            // last tile of an in-place body: input fragment m-1 died with k-step m-1, outputs 0..11 are complete
            if constexpr (KIND == BW_HID && u == NU - 1) {
                if constexpr (n >= 1 && n - 1 < 12) { xh[n - 1] = nh[n - 1]; xl[n - 1] = nl[n - 1]; }
            }
#ifdef NERF_BWD_KSTEP_FENCE
            // experiment, off: pin every k-step's epilogue work behind ITS MFMA (left alone, hipcc pairs the MFMAs of two
            // k-steps and runs both epilogues after them).  The fenced schedule alternates 1 MFMA : 4-7 vector instructions
            // as written -- and measured 1 % slower per mixed_float16 step: the order is not what keeps the epilogue from
            // overlapping the matrix pipe (profiles/r3_diagnostic_ab.txt)
            if constexpr (FAST) __builtin_amdgcn_sched_barrier(0);
#endif
        });
    });
    if constexpr (KIND == BW_XYZ) {      // end of the chain: the last encoding tile's epilogue has no chain to ride on
        f32x16& last = accs[NA - 1];
        static_for<0, 16>([&](auto rc) {
            xyz_reg(std::integral_constant<int, NU - 1>{}, rc, last[decltype(rc)::value]);
        });
    }
    if constexpr (NQ % kBCQ != 0 && NQ % kBCQ <= kBCQ / 2) pipe_sync_c<kBCQ, kBRing, 1, kBwdExtra>(p);
    p.ck += 1;
}

template <bool DX, bool FAST, bool XYZ = false>
__device__ __forceinline__ void mlp_bwd_body(const MlpBwdArgs& a) {
    constexpr int NMQ = XYZ ? 10 : 9;           // mask records / gradient buffers of the network
    using G = BGeo<FAST>;
    constexpr int kBCQ = G::CQ, kBRing = G::Ring, kBChunkBytes = G::ChunkBytes;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const uint32_t lane16 = kLdsRing + lane * 16;
    uint32_t cb_h = G::LdsConst + h * 16;
    asm volatile("" : "+v"(cb_h));      // opaque base: immediate ds_read offsets instead of one address VGPR per constant (see mlp_f16_2t.hip)

    const long long ntiles = a.Mp / 128;
    if ((long long)blockIdx.x >= ntiles) return;

    for (int i = tid; i < (XYZ ? kXConstFloats : kBConstFloats) / 4; i += 256)
        reinterpret_cast<f32x4*>(smem + G::LdsConst)[i] = reinterpret_cast<const f32x4*>(a.wconst)[i];
    if (tid < 16) reinterpret_cast<uint32_t*>(smem + G::LdsGmax)[tid] = 0u;

    Pipe p;
    p.ck = 0;
    p.src_next = 0;
    p.n_chunks = bwd_stream_chunks<FAST>(DX, XYZ);
    p.wbase = reinterpret_cast<const char*>(a.wstream);
    p.voff = wave * (kBCQ / 4 * kQuadBytes) + lane * 16;
    p.wave_lds = wave * (kBCQ / 4 * kQuadBytes);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kBRing - 1; ++c) {
        p.cur_src = p.wbase + (size_t)p.src_next * kBChunkBytes;
        p.cur_dst = kLdsRing + c * kBChunkBytes + p.wave_lds;
        p.src_next += 1;
#pragma unroll
        for (int q = 0; q < kBCQ / 4; ++q) dma_piece(p.cur_src, p.voff + q * kQuadBytes, p.cur_dst + q * kQuadBytes);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kBCQ / 4) * (kBRing - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    frag4 xh[16], xl[16], nh[14], nl[14];
    f32x16 accs[G::NAcc];
    const float alpha = a.alpha;

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long m = tile * 128 + wave * 32 + j;           // rows beyond M exist (padding) and carry zero Graw
        const long long off128 = (m - j) * a.ld9 + (h * 32 + j) * 4; // this lane's slot in the fragment-major G9 rows
        // everything this tile reads from HBM, in one batch (a compiler-placed vmcnt wait drains the DMA ring once)
        const f32x4 graw = *reinterpret_cast<const f32x4*>(a.graw + m * 4);
        if (a.dsig && h == 0) a.dsig[m] = graw[3];
        frag4 mq[NMQ];
#pragma unroll
        for (int l = 0; l < NMQ; ++l)       // (read once: non-temporal, like the stash stores)
            mq[l] = __builtin_nontemporal_load(reinterpret_cast<const frag4*>(a.mask_ptr[NMQ - 1 - l]) + m * 2 + h);
        // mq[0] = mask of h9 (layer 8's output), mq[1] = h8, ..., mq[8] = h1 (layer 0's output); the xyz-only network has
        // its extra layer's record at mq[1] and everything else one further down

        // ---- rgb head backward on the VALU: G9 = (W9 . d_rgb) * LeakyReLU'(h9), in fragment (k) order ----
        float g9[64];
        float mt = 0.f;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int t = n >> 1, s = n & 1;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                constexpr int kWrgb = XYZ ? kXConstWrgb : kBConstWrgb;
                const f32x4 w0 = lds_read4(cb_h + (kWrgb + 0 * 128 + 32 * t + 16 * s + 8 * g) * 4);
                const f32x4 w1 = lds_read4(cb_h + (kWrgb + 1 * 128 + 32 * t + 16 * s + 8 * g) * 4);
                const f32x4 w2 = lds_read4(cb_h + (kWrgb + 2 * 128 + 32 * t + 16 * s + 8 * g) * 4);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int bit = mask_bit(t, 8 * s + 4 * g + e);
                    float v = w0[e] * graw[0] + w1[e] * graw[1] + w2[e] * graw[2];
                    v = ((mq[0][t >> 1] >> bit) & 1u) ? alpha * v : v;
                    o[e] = v;
                    g9[n * 8 + g * 4 + e] = v;
                    mt = fmaxf(mt, fabsf(v));
                }
                if constexpr (FAST)
                    stream_store(reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.d_ptr[NMQ - 1]) + off128 + 32 * (32 * t + 16 * s + 8 * g)),
                                 make_uint2(pack_h2(o[0], o[1]), pack_h2(o[2], o[3])));
                else if constexpr (!kPair16)
                    stream_store(reinterpret_cast<f32x4*>(a.d_ptr[NMQ - 1] + off128 + 32 * (32 * t + 16 * s + 8 * g)), o);
            }
        }
        mt = max_with_other_half(mt);
        lds_gmax_update<FAST>(0, __float_as_uint(mt));
        const float m_true = XYZ ? mt : fmaxf(mt, fabsf(graw[3]));   // (xyz-only: d_sigma enters one body later)
        BwdLane L;
        const float sig = pow2_to_peak(m_true);
        L.inv_sig = pow2_inverse(sig);
        L.rho = pow2_to_peak(m_true * sig);
        L.inv_prev = L.inv_sig; L.rho_prev = L.rho;
        L.mrun = 0.f;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                if constexpr (FAST) {
                    nh[n][e >> 1] = pack_h2(g9[n * 8 + e] * sig, g9[n * 8 + e + 1] * sig);
                    nl[n][e >> 1] = 0u;
                    continue;
                }
                float h0, l0, h1, l1;
                split_trunc(g9[n * 8 + e] * sig, h0, l0);
                split_trunc(g9[n * 8 + e + 1] * sig, h1, l1);
                nh[n][e >> 1] = pack_h2(h0, h1);
                nl[n][e >> 1] = pack_h2(l0, l1);
            }
            if constexpr (!FAST && kPair16) {      // G9 in pair16 form: the operand fragments are its rows (scale sig, row factor 1 / sig)
                const int t = n >> 1, s = n & 1;
                stream_store(reinterpret_cast<frag4*>(a.d_ptr[NMQ - 1] + off128 + 32 * (32 * t + 16 * s)), frag4{nh[n][0], nh[n][1], nl[n][0], nl[n][1]});
                stream_store(reinterpret_cast<frag4*>(a.d_ptr[NMQ - 1] + off128 + 32 * (32 * t + 16 * s + 8)), frag4{nh[n][2], nh[n][3], nl[n][2], nl[n][3]});
            }
        }
        // row factor of a pair16 gradient buffer: the upper half of a power of two's fp32 bits (one 2-byte store per row)
        auto store_rs = [&](int buf, float r) {
            if constexpr (!FAST && kPair16) {
                if (h == 0) a.rs_ptr[buf][m] = (uint16_t)(__float_as_uint(r) >> 16);
            }
        };
        store_rs(NMQ - 1, L.inv_sig);
        {   // the sigma head's rank-1 term rides as a 9th k-step: element 0 of lane half 0 = d_sigma
            float h0, l0;
            const float ds = XYZ ? 0.f : graw[3];      // xyz-only: this k-step multiplies zero weights
            if constexpr (FAST) { h0 = h ? 0.f : ds * sig; l0 = 0.f; }
            else split_trunc(h ? 0.f : ds * sig, h0, l0);
            nh[8] = frag4{pack_h2(h0, 0.f), 0u, 0u, 0u};
            nl[8] = frag4{pack_h2(l0, 0.f), 0u, 0u, 0u};
        }

        const long long off256 = (m - j) * a.ld + (h * 32 + j) * 4;      // fragment-major D rows: block of 32 rows, lane slot
        float* d_prev = nullptr;
        // (FAST: the D rows are fp16; the pointers stay float* and are advanced in halfs)
        auto d_row = [&](int l) -> float* {
            if constexpr (FAST) return reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(a.d_ptr[l]) + off256);
            else return a.d_ptr[l] + off256;
        };
        float* d_cur = d_row(XYZ ? 8 : 7);
        const long long offx = (m - j) * kBwdXyzLd + (h * 32 + j) * 4;      // the dx rows are fragment-major as well
        float* dxa = DX ? a.dx_ptr[0] + offx : nullptr;
        float* dxb = DX ? a.dx_ptr[1] + offx : nullptr;
        frag4 mk_prev = mq[1], mk_cur = mq[1];
        bwd_body<BW_HEAD, 0, false, FAST>(p, lane16, alpha, L, false, d_prev, d_cur, nullptr, mk_prev, mk_cur, 0, accs, xh, xl,
                                    nh, nl);
        int l_cur = XYZ ? 8 : 7;           // index of the buffer d_cur points into
        auto rotate = [&](int l_out) {
            // the per-lane scale state and the mask queue (mq[1] is always the mask of the operand just produced)
            L.inv_prev = L.inv_sig; L.rho_prev = L.rho;
            L.inv_sig = L.inv_sig * pow2_inverse(L.rho);
            store_rs(l_cur, L.inv_sig);     // the body that just ended packed its outputs with rho: true D = packed * inv / rho
            l_cur = l_out;
            mk_prev = mk_cur;
#pragma unroll
            for (int i = 1; i < NMQ - 1; ++i) mq[i] = mq[i + 1];
            mk_cur = mq[1];
            d_prev = d_cur;
            d_cur = d_row(l_out);
        };
        if constexpr (XYZ) {
            // the extra layer's body: consumes D8b (pre-activation gradient of the 256-wide layer after h8), produces
            // D7 = (W8 . D8b + w_sigma d_sigma) * LeakyReLU'(h8); the previous body was BW_HEAD (no staged tail to copy)
            rotate(7);
            bwd_body<BW_HID, 0, true, FAST, true, false>(p, lane16, alpha, L, false, d_prev, d_cur, nullptr, mk_prev, mk_cur, 1,
                                                         accs, xh, xl, nh, nl, graw[3], cb_h);
        }
#pragma unroll 1
        for (int l = 7; l >= 1; --l) {         // body of layer l: consumes D_l, produces D_(l-1)
            rotate(l - 1);
            const int gslot_prev = NMQ - 1 - l;    // slot k <-> d_ptr[NMQ - 1 - k]
            if (DX && l == 4)
                bwd_body<BW_HID, DX ? 2 : 0, true, FAST>(p, lane16, alpha, L, true, d_prev, d_cur, dxa, mk_prev, mk_cur,
                                                   gslot_prev, accs, xh, xl, nh, nl);
            else if (XYZ && l == 7)            // ... whose last tile (finished here) still needs the sigma term
                bwd_body<BW_HID, 0, true, FAST, false, XYZ>(p, lane16, alpha, L, true, d_prev, d_cur, nullptr, mk_prev, mk_cur,
                                                            gslot_prev, accs, xh, xl, nh, nl, graw[3], cb_h);
            else
                bwd_body<BW_HID, 0, true, FAST>(p, lane16, alpha, L, XYZ || l != 7, d_prev, d_cur, nullptr, mk_prev, mk_cur, gslot_prev,
                                          accs, xh, xl, nh, nl);
        }
        L.inv_prev = L.inv_sig; L.rho_prev = L.rho;
        L.inv_sig = L.inv_sig * pow2_inverse(L.rho);
        store_rs(0, L.inv_sig);
        if constexpr (DX) {
            bwd_body<BW_XYZ, 2, true, FAST>(p, lane16, alpha, L, true, d_cur, nullptr, dxb, mk_cur, mk_cur, NMQ - 1, accs, xh, xl, nh, nl);
        } else {
            // flush: D0's last tile has no chain to ride on; nothing is packed any more
            f32x16& last = accs[G::NAcc - 1];
            float tmax = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int bit = mask_bit(7, r + e);
                    // pair16: in the scale of D0's other tiles (rho of the last body); true D = that * inv_sig
                    const float v = last[r + e] * (((mk_cur[3] >> bit) & 1u) ? alpha : 1.0f) * (!FAST && kPair16 ? L.rho_prev : L.inv_prev);
                    o[e] = v;
                    tmax = fmaxf(tmax, fabsf(v));
                }
                if constexpr (!FAST && kPair16) {
                    float h0, l0, h1, l1, h2_, l2, h3, l3;
                    split_trunc(o[0], h0, l0); split_trunc(o[1], h1, l1); split_trunc(o[2], h2_, l2); split_trunc(o[3], h3, l3);
                    stream_store(reinterpret_cast<frag4*>(d_cur + 32 * (32 * 7 + 8 * (r >> 2))),
                                 frag4{pack_h2(h0, h1), pack_h2(h2_, h3), pack_h2(l0, l1), pack_h2(l2, l3)});
                } else if constexpr (FAST)
                    stream_store(reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(d_cur) + 32 * (32 * 7 + 8 * (r >> 2))),
                                 make_uint2(pack_h2(o[0], o[1]), pack_h2(o[2], o[3])));
                else
                    stream_store(reinterpret_cast<f32x4*>(d_cur + 32 * (32 * 7 + 8 * (r >> 2))), o);
            }
            // max|D0| of this sample: the already packed part (in the next operand's scale) and the flushed tile
            if constexpr (!FAST && kPair16) tmax *= L.inv_sig;
            tmax = fmaxf(tmax, L.mrun * L.inv_sig);
            lds_gmax_update<FAST>(NMQ - 1, __float_as_uint(max_with_other_half(tmax)));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // (lgkmcnt: the asm ds_max_u32 updates of the gmax slots)
    __syncthreads();
    if (tid < NMQ && a.gmax) {
        const uint32_t v = reinterpret_cast<const uint32_t*>(smem + G::LdsGmax)[tid];
        unsigned* slot = a.gmax + 64 * tid + (blockIdx.x & 63);
        if (v > *slot) atomicMax(slot, v);
    }
}

__global__ __launch_bounds__(256, 1) void mlp_bwd_f16x3_kernel(const MlpBwdArgs a) { mlp_bwd_body<false, false>(a); }
__global__ __launch_bounds__(256, 1) void mlp_bwd_f16x3_dx_kernel(const MlpBwdArgs a) { mlp_bwd_body<true, false>(a); }
__global__ __launch_bounds__(256, BGeo<true>::Occ) void mlp_bwd_f16_kernel(const MlpBwdArgs a) { mlp_bwd_body<false, true>(a); }
__global__ __launch_bounds__(256, BGeo<true>::Occ) void mlp_bwd_f16_dx_kernel(const MlpBwdArgs a) { mlp_bwd_body<true, true>(a); }
// the xyz-only network (n_angles_for_model = 0): one more 256-wide body, sigma's rank-1 term one body later
__global__ __launch_bounds__(256, 1) void mlp_bwd_f16x3_xyz_kernel(const MlpBwdArgs a) { mlp_bwd_body<false, false, true>(a); }
__global__ __launch_bounds__(256, 1) void mlp_bwd_f16x3_xyz_dx_kernel(const MlpBwdArgs a) { mlp_bwd_body<true, false, true>(a); }
__global__ __launch_bounds__(256, BGeo<true>::Occ) void mlp_bwd_f16_xyz_kernel(const MlpBwdArgs a) { mlp_bwd_body<false, true, true>(a); }
__global__ __launch_bounds__(256, BGeo<true>::Occ) void mlp_bwd_f16_xyz_dx_kernel(const MlpBwdArgs a) { mlp_bwd_body<true, true, true>(a); }

void launch_mlp_bwd_f16x3(const MlpBwdArgs& a, bool dx, bool single_pass, int num_cus, hipStream_t stream, bool xyz_only) {
    if (a.Mp <= 0) return;
    const long long ntiles = a.Mp / 128;
    const long long slots = (long long)num_cus * (single_pass ? BGeo<true>::Occ : 1);      // resident workgroups
    const int grid = (int)(ntiles < slots ? ntiles : slots);
    auto* k = xyz_only ? (single_pass ? (dx ? mlp_bwd_f16_xyz_dx_kernel : mlp_bwd_f16_xyz_kernel)
                                      : (dx ? mlp_bwd_f16x3_xyz_dx_kernel : mlp_bwd_f16x3_xyz_kernel))
              : single_pass ? (dx ? mlp_bwd_f16_dx_kernel : mlp_bwd_f16_kernel)
                            : (dx ? mlp_bwd_f16x3_dx_kernel : mlp_bwd_f16x3_kernel);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), single_pass ? BGeo<true>::LdsTotal : BGeo<false>::LdsTotal, stream, a);
}

void mlp_bwd_f16x3_set_attributes() {
    for (auto* k : {mlp_bwd_f16x3_kernel, mlp_bwd_f16x3_dx_kernel, mlp_bwd_f16x3_xyz_kernel, mlp_bwd_f16x3_xyz_dx_kernel})
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, BGeo<false>::LdsTotal);
    for (auto* k : {mlp_bwd_f16_kernel, mlp_bwd_f16_dx_kernel, mlp_bwd_f16_xyz_kernel, mlp_bwd_f16_xyz_dx_kernel})
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, BGeo<true>::LdsTotal);
}

// ------------------------------------------------------------------------------------------------
// The backward stream as an index map (host): idx[slot] = 2 * (src + 1) + is_lo, 0 = padding; src = index into the
// blob (Keras get_weights() order).  Re-packed on the device after every optimizer step by repack_bwd_kernel.
// ------------------------------------------------------------------------------------------------
void build_bwd_gather(int n_angles, bool dx, bool hi_only, int32_t* idx /* kBwdStreamBytes / 2 entries */) {
    memset(idx, 0, (kBwdStreamBytes / 2) * sizeof(int32_t));
    const int kd = 256 + 8 * (n_angles + 1);
    const bool xyz_only = n_angles == 0;
    const int shapes_dir[12][2] = {{33, 256}, {256, 256}, {256, 256}, {256, 256}, {289, 256}, {256, 256},
                                   {256, 256}, {256, 256}, {kd, 128}, {128, 3}, {kd, 1}, {0, 0}};
    // get_network_only_xyz (src/NeRF.py:248-288): ..., 8: 256 -> 256, 9: 256 -> 128, 10: 128 -> 3, 11: 256 -> 1
    const int shapes_xyz[12][2] = {{33, 256}, {256, 256}, {256, 256}, {256, 256}, {289, 256}, {256, 256},
                                   {256, 256}, {256, 256}, {256, 256}, {256, 128}, {128, 3}, {256, 1}};
    const int (*shapes)[2] = xyz_only ? shapes_xyz : shapes_dir;
    long long koff[12], off = 0;
    for (int i = 0; i < 12; ++i) { koff[i] = off; off += (long long)shapes[i][0] * shapes[i][1] + shapes[i][1]; }
    const int cq = hi_only ? BGeo<true>::CQ : BGeo<false>::CQ;        // the two arithmetics run different ring geometries
    const int chunk_halfs = cq * (kQuadBytes / 2);
    size_t chunk = 0;
    // one body: NU tiles x NSTEP k-steps; src(u, i, n, e, h) gives the blob index of A[i][k] or -1
    auto emit = [&](int NU, int NSTEP, auto src) {
        const long long b0 = (long long)chunk * chunk_halfs;
        for (int u = 0; u < NU; ++u)
            for (int n = 0; n < NSTEP; ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const long long s = src(u, lane & 31, n, e, lane >> 5);
                        if (s < 0) continue;
                        if (hi_only) {
                            idx[b0 + (long long)(u * NSTEP + n) * (kQuadBytes / 2) + lane * 8 + e] = (int32_t)(2 * (s + 1));
                            continue;
                        }
                        const long long q = (long long)(u * NSTEP + n) * 2;
                        idx[b0 + (q + 0) * (kQuadBytes / 2) + lane * 8 + e] = (int32_t)(2 * (s + 1));
                        idx[b0 + (q + 1) * (kQuadBytes / 2) + lane * 8 + e] = (int32_t)(2 * (s + 1) + 1);
                    }
        chunk += (NU * NSTEP * (hi_only ? 1 : 2) + cq - 1) / cq;
    };
    // layer 8 transposed (+ the sigma head's hidden rows as the 9th k-step); xyz-only network: layer 9 (256 -> 128)
    // transposed, the 9th k-step multiplies zeros (its sigma term is added one body later, on the VALU)
    emit(8, kBStepsHead, [&](int u, int i, int n, int e, int h) -> long long {
        if (n < 8) return koff[xyz_only ? 9 : 8] + (long long)(32 * u + i) * 128 + frag_feature(n, e, h);
        if (xyz_only) return -1;
        return (e == 0 && h == 0) ? koff[10] + (32 * u + i) : -1;
    });
    auto hidden = [&](int l, int row0) {
        return [=](int u, int i, int n, int e, int h) -> long long {
            return koff[l] + (long long)(row0 + 32 * u + i) * 256 + frag_feature(n, e, h);
        };
    };
    auto with_xyz = [&](int l, int nx, int row0) {       // nx encoding tiles first, then (for layer 4) the hidden rows
        return [=](int u, int i, int n, int e, int h) -> long long {
            if (u < nx) {
                const int row = 32 * u + i;
                return row < kXyzDim ? koff[l] + (long long)row * 256 + frag_feature(n, e, h) : -1;
            }
            return koff[l] + (long long)(row0 + 32 * (u - nx) + i) * 256 + frag_feature(n, e, h);
        };
    };
    for (int l = xyz_only ? 8 : 7; l >= 5; --l) emit(8, 16, hidden(l, 0));
    if (dx) emit(10, 16, with_xyz(4, 2, kXyzDim));
    else emit(8, 16, hidden(4, kXyzDim));
    for (int l = 3; l >= 1; --l) emit(8, 16, hidden(l, 0));
    if (dx) emit(2, 16, with_xyz(0, 2, 0));
}

__global__ void repack_bwd_kernel(const float* __restrict__ blob, const int32_t* __restrict__ idx,
                                  uint16_t* __restrict__ stream, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t t = idx[i];
    uint16_t out = 0;
    if (t != 0) {
        const float w = blob[(t >> 1) - 1];
        const _Float16 hi = (_Float16)w;
        const _Float16 v = (t & 1) ? (_Float16)(w - (float)hi) : hi;
        out = __builtin_bit_cast(uint16_t, v);
    }
    stream[i] = out;
}

void launch_repack_bwd(const float* blob, const int32_t* idx, void* stream, hipStream_t s) {
    const size_t n = kBwdStreamBytes / 2;
    hipLaunchKernelGGL(repack_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, blob, idx,
                       reinterpret_cast<uint16_t*>(stream), n);
}

}  // namespace nerf

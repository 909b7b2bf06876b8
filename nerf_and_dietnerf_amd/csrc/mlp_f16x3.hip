// mlp_f16x3.hip -- fused positional-encoding + 11-layer NeRF MLP forward on the fp16 matrix cores with
// fp32-class accuracy: every fp32 operand is split x = hi + lo (two fp16 values, 22 significant bits)
// and each product is formed in three MFMA passes  hi*hi + lo*hi + hi*lo  (the lo*lo term, 2^-22
// relative, is dropped), accumulated in fp32 by v_mfma_f32_32x32x16_f16.
//
// Same reference chain as mlp_fp32.hip (src/UtilsCV.py:584-599,124-143; src/UtilsNRF.py:52-85;
// src/NeRF.py:316-339) and the same kernel skeleton: one wave per SIMD, 32 samples per wave on the
// lane, accumulator registers re-used as the next layer's B operand (here: converted to packed fp16
// hi/lo fragments by the epilogue), weights streamed L2 -> LDS ring by LDS-DMA.  Differences:
//   * k-step = 16 (one 32x32x16 MFMA): fragment element e of lane half h of k-step (t,s) is feature
//     32t + 16s + 8(e>>2) + 4h + (e&3) -- the accumulator-as-operand order of the 32x32 C/D layout;
//   * the weight stream holds an fp16 hi fragment and an fp16 lo fragment per (tile, k-step);
//   * the sigma head (280 -> 1) rides the MFMA as a 5th output tile of layer 8 (its inputs only
//     exist as fp16 hi/lo fragments); the rgb head (128 -> 3) stays on the VALU in fp32.
// Measured accuracy (tests/test_gpu_parity.py): final RGB within 1e-4 of the fp32 oracle.
#include "mlp_f16_frag.h"

#include <math.h>
#include <string.h>

namespace nerf {

#ifdef NERF_STAMPS
__device__ unsigned long long g_stamps_h[16];
#endif

// One dense layer on the fp16 matrix cores, u-outer (one accumulator chain per 32-wide output tile).
// The epilogue of a tile (bias, LeakyReLU, hi/lo split, fp16 pack) is dealt out ONE accumulator
// register per k-step over the NEXT tile's chain -- also across layer boundaries (PENDING: the
// previous layer's last tile is finished during this layer's first tile), so no epilogue is exposed.
//   fragments of the layer input live in xh/xl[16]; outputs of tiles 0..6 go to nh/nl and are copied
//   back as the last tile's chain retires the k-steps that read them; tile 7 lands in xh/xl[14..15].
// FAST = single-pass fp16 mode (NERF_PRECISION_F16: the reference's production mixed_float16 numerics): only the
// hi*hi product is formed and the activations are rounded (RNE) to fp16 between layers; same stream, same schedule.
// Since round 4 its hidden layers use the two-tile render kernel's packed-pair epilogue (NERF_FAST_PACKED_EPI): the fp32
// sum (bias = C-in) is cast to fp16 FIRST, LeakyReLU runs on the pair in fp16 with alpha rounded to fp16 -- where Keras'
// mixed_float16 LeakyReLU rounds; the oracles' emulations round there too (oracle.mlp_forward_fp16(..., "c_in"),
// oracle.train_oracle._LRelu16).  The stash forward of the mixed_float16 trainer: 966 -> 908 us per step.
// STASH = training forward: every activation is also written as fp32 to HBM (st_prev: the previous layer's output rows
// of this lane's sample, for its PENDING tile 7; st_cur: this layer's), four consecutive features per float4 store, in the
// FRAGMENT-MAJOR layout of frag_layout.h::frag_index (a store instruction writes 1 KiB / 512 B of consecutive bytes).
// FAST + STASH (the mixed_float16 policy's forward) writes the stash in fp16 -- the very dwords it packs as the next
// layer's operand, four consecutive features per 8-byte store -- and st_prev / st_cur then point at fp16 rows.
// It also records the LeakyReLU' masks the fused backward (mlp_bwd_f16x3.hip) multiplies by: one bit per activation
// (its SIGN bit: 1 = negative), bit mask_bit(ut, r) (mlp_f16_frag.h) of word ut >> 1 of the lane's 128-bit record =
// accumulator register r of output tile ut, i.e. indexed exactly as the backward's accumulators hold the gradient of
// that activation; one 16-byte store per lane and layer (32 B per sample row instead of re-reading the 1 KB stash row
// for its signs).
// ROT rotates the four accumulators: tile u works in accs[(u + ROT) & 3] and a PENDING tile is looked for in
// accs[(ROT + 3) & 3]; every body but BODY_LAST0 (which follows the 9-tile BODY_HIDSIG) uses ROT = 0.
template <int BODY, bool PENDING, bool FAST, bool STASH = false, int ROT = 0>
__device__ __forceinline__ void layer_body_h(Pipe& p, uint32_t lane16, uint32_t cb_h, int bias_off_bytes,
                                             float alpha, float* st_prev, float* st_cur, frag4* mk_prev_ptr,
                                             frag4* mk_cur_ptr, frag4& mk_prev, frag4& mk_cur, f32x16 (&accs)[4],
                                             frag4 (&xh)[16], frag4 (&xl)[16], frag4 (&nh)[14], frag4 (&nl)[14],
                                             const frag4 (&peh)[3], const frag4 (&pel)[3], const frag4 (&dh)[2],
                                             const frag4 (&dl)[2], float (&xc)[64], float& sigma_raw) {
    constexpr int NU = BODY == BODY_LAST ? kHTilesLast : BODY == BODY_HIDSIG ? 9 : BODY == BODY_LAST0 ? 4 : 8;
    constexpr int NSTEP = BODY == BODY_PE ? kHStepsPE : (BODY == BODY_HID || BODY == BODY_HIDSIG || BODY == BODY_LAST0) ? kHStepsHid
                          : BODY == BODY_SKIP ? kHStepsPE + kHStepsHid : kHStepsHid + kHStepsDir;
    constexpr int SIG0 = BODY == BODY_HIDSIG ? 1 : 0;      // tiles before the first hidden tile (the leading sigma tile)
    constexpr int TPS = FAST ? 1 : 2;        // quads (A fragments) per k-step: hi [, lo]
    constexpr int QPU = TPS * NSTEP;
    constexpr int NQ = NU * QPU;
    // A fragments are fetched kPf quads (2 k-steps = 6 MFMAs = 192 cycles) ahead of their MFMAs.  The
    // ds_read is inline asm with a hand-counted s_waitcnt: left to itself hipcc sinks the reads next to
    // their MFMA (register pressure) and every k-step then waits out the LDS latency.  Destinations are
    // AGPRs ("a"): plentiful here, legal as MFMA A operands, and never shuffled by the allocator between
    // the read and its wait.  LDS returns in order, so lgkmcnt(kPf-1) before quad Q's use means "Q has
    // landed" however many younger compiler-issued LDS reads are in flight (they only make it stricter).
#ifndef NERF_KPF
#define NERF_KPF 4
#endif
#ifndef NERF_KPF_FAST
#define NERF_KPF_FAST 8
#endif
    constexpr int kPf = FAST ? NERF_KPF_FAST : NERF_KPF;   // single-pass: a quad is consumed every 32 cycles, look further ahead
#ifndef NERF_STASH_EXTRA_FAST
#define NERF_STASH_EXTRA_FAST 0
#endif
#ifndef NERF_STASH_EXTRA_3P
#define NERF_STASH_EXTRA_3P 0
#endif
    constexpr int kStashExtra = STASH ? (FAST ? NERF_STASH_EXTRA_FAST : NERF_STASH_EXTRA_3P) : 0;
    f32x4 pf[kPf];
    const int ck0 = p.ck;   // chunk of quad 0 of this body; quad Q lives in chunk ck0 + Q/16
    uint32_t rdbase[2];     // LDS address of the ring slot of an even / odd chunk (refreshed as chunks retire)
    rdbase[0] = lane16 + (uint32_t)((ck0 + 0) & (kHRing - 1)) * kHChunkBytes;
    rdbase[1] = lane16 + (uint32_t)((ck0 + 1) & (kHRing - 1)) * kHChunkBytes;
    auto issue_read = [&](auto qc) {
        constexpr int Qa = decltype(qc)::value;
        lds_read_frag_asm<(Qa % kHCQ) * kQuadBytes>(pf[Qa % kPf], rdbase[(Qa / kHCQ) & 1]);
    };
    static_for<0, kPf>([&](auto ic) {
        if constexpr (decltype(ic)::value < NQ) issue_read(ic);
    });

    // Epilogue of accumulator registers r, r+1 (r even) of output tile `ut` of a hidden layer, in three
    // phases that the k-step interleaves with its three MFMAs (each 32-cycle MFMA hides ~4 plain VALU
    // ops or ~2 conversions/moves; clustered they stall the matrix pipe):
    //   A/B: y = LeakyReLU(acc + bias) for r / r+1     C: split hi/lo, pack fp16 pairs, store fragment dword
    auto act = [&](float v) -> float {
        const float av = alpha * v;
        float y;
        asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(v), "v"(av));
        return y;
    };
    // bias = initial accumulator (C-in), loaded one tile ahead into the accumulator that tile will use
    auto load_bias = [&](int off_bytes, f32x16& dst) {
        static_for<0, 4>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const f32x4 b = lds_read4(cb_h + off_bytes + g * 32);
            dst[4 * g + 0] = b[0]; dst[4 * g + 1] = b[1]; dst[4 * g + 2] = b[2]; dst[4 * g + 3] = b[3];
        });
    };
    float sq0 = 0.f, sq1 = 0.f;      // first half of the float4 being collected for the stash
    (void)sq0; (void)sq1;
    auto stash4 = [&](auto utc, auto rc, float y0, float y1, float* base) {
        // registers r..r+1 of output tile ut = features 32 ut + 8 (r >> 2) + 4 h + (r & 3) .. (+1); h sits in `base`
        constexpr int ut = decltype(utc)::value;
        constexpr int r = decltype(rc)::value;
        if constexpr ((r & 3) == 0) { sq0 = y0; sq1 = y1; }
        else {
            f32x4 v;
            v[0] = sq0; v[1] = sq1; v[2] = y0; v[3] = y1;
#ifndef NERF_DIAG_NO_STASH_STORES   // timing-only diagnostic: everything but the store instruction itself
            stream_store(reinterpret_cast<f32x4*>(base + 32 * (32 * ut + 8 * (r >> 2))), v);
#else
            asm volatile("" :: "v"(v), "v"(base));
#endif
        }
    };
    uint32_t sqh = 0u;               // fp16 stash: the first packed pair of the four features being collected
    (void)sqh;
    auto stash4h = [&](auto utc, auto rc, uint32_t ph, float* base) {
        constexpr int ut = decltype(utc)::value;
        constexpr int r = decltype(rc)::value;
        if constexpr ((r & 3) == 0) sqh = ph;
#ifndef NERF_DIAG_NO_STASH_STORES
        else stream_store(reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + 32 * (32 * ut + 8 * (r >> 2))), make_uint2(sqh, ph));
#else
        else asm volatile("" :: "v"(sqh), "v"(ph), "v"(base));
#endif
    };
    auto store_pair = [&](auto utc, auto rc, float y0, float y1, auto dest_sel, auto pend_sel) {
        constexpr int ut = decltype(utc)::value;
        constexpr int r = decltype(rc)::value;
        constexpr int n = 2 * ut + (r >> 3), e = r & 7;
        uint32_t ph, pl = 0u;
        if constexpr (FAST) {
            ph = pack_h2(y0, y1);                                         // round to fp16, no lo part
        } else {
            float h0, l0, h1, l1;
            split_trunc(y0, h0, l0);
            split_trunc(y1, h1, l1);
            ph = pack_h2(h0, h1); pl = pack_h2(l0, l1);                   // whole-register writes
        }
        if constexpr (STASH) {
            if constexpr (FAST) stash4h(utc, rc, ph, decltype(pend_sel)::value ? st_prev : st_cur);
            else stash4(utc, rc, y0, y1, decltype(pend_sel)::value ? st_prev : st_cur);
            // LeakyReLU' record: the two SIGN bits of the packed pair, shifted in (mask_push, mlp_f16_frag.h) -- two plain
            // VALU ops per pair (two compares, two selects and an or before: the mask alone was a third of this
            // kernel's VALU work beside the single-pass MFMAs)
            if constexpr (decltype(pend_sel)::value) mk_prev[ut >> 1] = mask_push(mk_prev[ut >> 1], ph);
            else mk_cur[ut >> 1] = mask_push(mk_cur[ut >> 1], ph);
        }
        if constexpr (decltype(dest_sel)::value) { xh[n][e >> 1] = ph; if constexpr (!FAST) xl[n][e >> 1] = pl; }
        else { nh[n][e >> 1] = ph; if constexpr (!FAST) nl[n][e >> 1] = pl; }
    };

    // FAST, packed-pair epilogue (NERF_FAST_PACKED_EPI, the two-tile render kernel's: mlp_f16_2t.hip): the accumulator pair is
    // converted FIRST (v_cvt_pk_f16_f32), then alpha and max act on the pair (v_pk_mul_f16, v_pk_max_f16): 1.5 plain vector
    // instructions per value instead of 2.5 (the bias is the accumulator's C-in here)
#ifndef NERF_FAST_PACKED_EPI
#define NERF_FAST_PACKED_EPI 1
#endif
    constexpr bool kPackedEpi = FAST && NERF_FAST_PACKED_EPI != 0;
    const uint32_t alpha2 = pack_h2(alpha, alpha);
    (void)alpha2;
    auto act_pair = [&](float v0, float v1) -> uint32_t {
        const uint32_t pk = pack_h2(v0, v1);
        uint32_t q, y;
        asm("v_pk_mul_f16 %0, %1, %2" : "=v"(q) : "v"(pk), "v"(alpha2));
        asm("v_pk_max_f16 %0, %1, %2" : "=v"(y) : "v"(pk), "v"(q));
        return y;
    };
    auto store_pair_p = [&](auto utc, auto rc, uint32_t ph, auto dest_sel, auto pend_sel) {
        constexpr int ut = decltype(utc)::value;
        constexpr int r = decltype(rc)::value;
        constexpr int n = 2 * ut + (r >> 3), e = r & 7;
        if constexpr (STASH) {
            stash4h(utc, rc, ph, decltype(pend_sel)::value ? st_prev : st_cur);
            if constexpr (decltype(pend_sel)::value) mk_prev[ut >> 1] = mask_push(mk_prev[ut >> 1], ph);
            else mk_cur[ut >> 1] = mask_push(mk_cur[ut >> 1], ph);
        }
        if constexpr (decltype(dest_sel)::value) xh[n][e >> 1] = ph;
        else nh[n][e >> 1] = ph;
    };
    (void)act_pair; (void)store_pair_p;
    float ycarry = 0.f;
    (void)ycarry;
    static_for<0, NU>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        f32x16& acc = accs[(u + ROT) & 3];
        f32x16& prv = accs[(u + ROT + 3) & 3];
        f32x16& nxt = accs[(u + ROT + 1) & 3];
        if constexpr (u == 0 && !PENDING) load_bias(bias_off_bytes, acc);   // first layer of a tile: exposed once
        static_for<0, NSTEP>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            f32x4 araw[2];
            static_for<0, TPS>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int Q = u * QPU + TPS * n + t;
                constexpr int left = NQ - Q;
                (void)left;
                constexpr int qc = Q % kHCQ;        // position inside the chunk
                if constexpr (qc == 0 && Q > 0) p.ck += 1;
                if constexpr (qc == kHCQ / 2) {
                    // pieces 1.. go out at quads +2, +4, ...; those that would fall past the body's end go now
                    constexpr int room = (NQ - 1 - Q) / 2;                 // pieces that still find a quad
                    pipe_sync_c<kHCQ, kHRing, (room + 1 < kHCQ / 4 ? room + 1 : kHCQ / 4), kStashExtra>(p);
                }
                if constexpr (qc > kHCQ / 2 && (qc - kHCQ / 2) % 2 == 0) {
                    // was this chunk's sync a tail case (pieces issued out of order)?  then re-establish M0
                    constexpr int Qs = Q - (qc - kHCQ / 2);
                    constexpr bool tail = (NQ - 1 - Qs) / 2 + 1 < kHCQ / 4;
                    pipe_piece_c<(qc - kHCQ / 2) / 2, tail>(p);
                }
                // quad Q has landed once at most the kPf-1 younger fragment reads are outstanding
                lds_wait_frag_asm<(NQ - Q >= kPf ? kPf - 1 : NQ - Q - 1)>(pf[Q % kPf]);
                araw[t] = pf[Q % kPf];
                // chunk ck+1 is complete for every wave once this chunk's sync (quad 8) has passed, so a
                // lookahead of 4 quads never reads ahead of a sync it depends on
                if constexpr (Q + kPf < NQ) {
                    constexpr int Qn = Q + kPf;
                    if constexpr (Qn % kHCQ == 0)   // first read of a new chunk: point its base at the slot
                        rdbase[(Qn / kHCQ) & 1] = lane16 + (uint32_t)((ck0 + Qn / kHCQ) & (kHRing - 1)) * kHChunkBytes;
                    issue_read(std::integral_constant<int, Qn>{});
                }
            });
            const h8 a_hi = __builtin_bit_cast(h8, araw[0]);
            const h8 a_lo = __builtin_bit_cast(h8, araw[FAST ? 0 : 1]);
            (void)a_lo;
            frag4 bh_, bl_;
            if constexpr (BODY == BODY_PE) { bh_ = peh[n]; bl_ = pel[n]; }
            else if constexpr (BODY == BODY_HID || BODY == BODY_HIDSIG || BODY == BODY_LAST0) { bh_ = xh[n]; bl_ = xl[n]; }
            else if constexpr (BODY == BODY_SKIP) {
                if constexpr (n < kHStepsPE) { bh_ = peh[n]; bl_ = pel[n]; }
                else { bh_ = xh[n - kHStepsPE]; bl_ = xl[n - kHStepsPE]; }
            } else {
                if constexpr (n < kHStepsHid) { bh_ = xh[n]; bl_ = xl[n]; }
                else { bh_ = dh[n - kHStepsHid]; bl_ = dl[n - kHStepsHid]; }
            }
            const h8 b_hi = __builtin_bit_cast(h8, bh_), b_lo = __builtin_bit_cast(h8, bl_);
            // Deferred epilogue: within a layer ONE accumulator register of the previous tile per k-step over
            // all 16 steps (fragment dwords are packed on odd steps); the previous LAYER's tile 7 goes a pair per
            // step over steps 0..7 because its fragments are read from step 14 on.
            constexpr bool kPend = (u == 0) && PENDING && n < 8;
            constexpr bool kPrevS = (u > SIG0) && BODY != BODY_PE && NSTEP >= 16 && n < 16;
            constexpr int et = kPend ? 7 : (u > SIG0 ? u - 1 - SIG0 : 0);      // hidden-tile index of the previous tile
            constexpr bool kXc = (BODY == BODY_LAST || BODY == BODY_LAST0) && kPrevS;
            // the leading sigma tile (row 0 = the raw density, no activation, src/NeRF.py:283) is complete after tile 0
            if constexpr (BODY == BODY_HIDSIG && u == 1 && n == 0) sigma_raw = prv[0];
            if constexpr (!FAST) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc, 0, 0, 0);
            if constexpr (kPend && kPackedEpi) {
                constexpr int er = 2 * n;
                store_pair_p(std::integral_constant<int, 7>{}, std::integral_constant<int, er>{}, act_pair(prv[er], prv[er + 1]), std::true_type{}, std::true_type{});
            } else if constexpr (kPrevS && !kXc && kPackedEpi) {
                if constexpr ((n & 1) == 1)
                    store_pair_p(std::integral_constant<int, et>{}, std::integral_constant<int, n - 1>{}, act_pair(prv[n - 1], prv[n]), std::false_type{}, std::false_type{});
            } else if constexpr (kPend) {
                constexpr int er = 2 * n;
                const float y0 = act(prv[er]), y1 = act(prv[er + 1]);
                store_pair(std::integral_constant<int, 7>{}, std::integral_constant<int, er>{}, y0, y1, std::true_type{}, std::true_type{});
            } else if constexpr (kPrevS) {
                const float y = act(prv[n]);
                if constexpr (kXc) {
                    xc[et * 16 + n] = y;
                    if constexpr (STASH) {      // layer 8's outputs (128 features) go to the stash in fours as well
                        if constexpr ((n & 1) == 0) ycarry = y;
                        else {
                            const uint32_t pp = pack_h2(ycarry, y);
                            if constexpr (FAST) stash4h(std::integral_constant<int, et>{}, std::integral_constant<int, n - 1>{}, pp, st_cur);
                            else stash4(std::integral_constant<int, et>{}, std::integral_constant<int, n - 1>{}, ycarry, y, st_cur);
                            mk_cur[et >> 1] = mask_push(mk_cur[et >> 1], pp);
                        }
                    }
                }
                else if constexpr ((n & 1) == 0) ycarry = y;
                else store_pair(std::integral_constant<int, et>{}, std::integral_constant<int, n - 1>{}, ycarry, y, std::false_type{}, std::false_type{});
            }
            if constexpr (u == 0 && PENDING) {
                if constexpr (STASH && n == 8) { if (mk_prev_ptr) stream_store(mk_prev_ptr, mk_prev); }   // that layer's mask word is complete
                // previous layer's tile 6 sits complete in nh/nl[12..13]; its k-steps are long retired
                if constexpr (n == 8) { xh[12] = nh[12]; xl[12] = nl[12]; }
                if constexpr (n == 9) { xh[13] = nh[13]; xl[13] = nl[13]; }
            }
            // preload the bias of the NEXT tile (or of the next layer's tile 0: the layers' biases are
            // contiguous in the constant region) into the accumulator it will use -- free since 2 tiles
            if constexpr (n == (NSTEP >= 12 ? 10 : 0)) {
                if constexpr (u + 1 < NU) load_bias(bias_off_bytes + (u + 1) * 128, nxt);
                else if constexpr (BODY != BODY_LAST && BODY != BODY_LAST0) load_bias(bias_off_bytes + NU * 128, nxt);
            }
            // layer 0 has only 3 k-steps per tile: its epilogues go in one block per tile
            if constexpr (BODY == BODY_PE && u > 0 && n == 0) {
                static_for<0, 8>([&](auto pc) {
                    constexpr int r = 2 * decltype(pc)::value;
                    if constexpr (kPackedEpi) {
                        const uint32_t ph = act_pair(prv[r], prv[r + 1]);
                        if constexpr (u - 1 <= 5) store_pair_p(std::integral_constant<int, u - 1>{}, std::integral_constant<int, r>{}, ph, std::true_type{}, std::false_type{});
                        else store_pair_p(std::integral_constant<int, u - 1>{}, std::integral_constant<int, r>{}, ph, std::false_type{}, std::false_type{});
                        return;
                    }
                    const float z0 = act(prv[r]), z1 = act(prv[r + 1]);
                    if constexpr (u - 1 <= 5) store_pair(std::integral_constant<int, u - 1>{}, std::integral_constant<int, r>{}, z0, z1, std::true_type{}, std::false_type{});
                    else store_pair(std::integral_constant<int, u - 1>{}, std::integral_constant<int, r>{}, z0, z1, std::false_type{}, std::false_type{});
                });
            }
            // last tile of an in-place layer: fragment m-1 of x-in died with k-step m-1; tiles 0..5 of
            // the new activations (fragments 0..11) are complete by now
            if constexpr ((BODY == BODY_HID || BODY == BODY_SKIP || BODY == BODY_HIDSIG) && u == NU - 1) {
                constexpr int m = BODY == BODY_SKIP ? n - kHStepsPE : n;
                if constexpr (m >= 1 && m - 1 < 12) { xh[m - 1] = nh[m - 1]; xl[m - 1] = nl[m - 1]; }
            }
        });
    });
    if constexpr (BODY == BODY_LAST0) {
        // the last 32 features have no following chain in this body: their activation is exposed once per tile
        f32x16& lastacc = accs[(NU - 1 + ROT) & 3];
#pragma unroll
        for (int r = 0; r < 16; ++r) xc[(NU - 1) * 16 + r] = act(lastacc[r]);
        if constexpr (STASH) {
            static_for<0, 8>([&](auto pc) {
                constexpr int r = 2 * decltype(pc)::value;
                const float y0 = xc[(NU - 1) * 16 + r], y1 = xc[(NU - 1) * 16 + r + 1];
                const uint32_t pp = pack_h2(y0, y1);
                if constexpr (FAST) stash4h(std::integral_constant<int, NU - 1>{}, std::integral_constant<int, r>{}, pp, st_cur);
                else stash4(std::integral_constant<int, NU - 1>{}, std::integral_constant<int, r>{}, y0, y1, st_cur);
                mk_cur[(NU - 1) >> 1] = mask_push(mk_cur[(NU - 1) >> 1], pp);
            });
            if (mk_cur_ptr) stream_store(mk_cur_ptr, mk_cur);      // the 128-wide layer's record is complete
        }
    }
    if constexpr (BODY == BODY_LAST) {
        // sigma row: feature row 0 of tile 4 = register 0 of lane half 0; raw, no activation (NeRF.py:336)
        sigma_raw = accs[(NU - 1) & 3][0];
        if constexpr (STASH) { if (mk_cur_ptr) stream_store(mk_cur_ptr, mk_cur); }   // layer 8 (128 features): all four tiles are finished
    }
    if constexpr (NQ % kHCQ != 0 && NQ % kHCQ <= kHCQ / 2) pipe_sync_c<kHCQ, kHRing, 1, kStashExtra>(p);
    p.ck += 1;
}

// fp32 values -> fp16 hi / lo fragments (FAST: hi = the value rounded to fp16, no lo)
template <bool FAST>
__device__ __forceinline__ void split8(const float (&v)[8], frag4& hi, frag4& lo) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        if constexpr (FAST) {
            hi[e >> 1] = pack_h2(v[e], v[e + 1]);
            lo[e >> 1] = 0u;
            continue;
        }
        float h0, l0, h1, l1;
        split_trunc(v[e], h0, l0);
        split_trunc(v[e + 1], h1, l1);
        hi[e >> 1] = pack_h2(h0, h1);
        lo[e >> 1] = pack_h2(l0, l1);
    }
}

template <bool FAST, bool STASH = false, bool XYZ = false>
__device__ __forceinline__ void mlp_f16_body(const MlpArgs& a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const uint32_t lane16 = kLdsRing + lane * 16;
    // opaque to the compiler: with the constant visible, hipcc folds kLdsConst (128 KiB, beyond a ds_read's 16-bit offset
    // field) into every constant-block address and keeps one base VGPR per distinct address (see mlp_f16_2t.hip)
    uint32_t cb_h = kLdsConst + h * 16;
    asm volatile("" : "+v"(cb_h));

    const long long ntiles = (a.M + 127) / 128;
    if ((long long)blockIdx.x >= ntiles) return;

    for (int i = tid; i < (XYZ ? kXConstFloats : kHConstFloats) / 4; i += 256)
        reinterpret_cast<f32x4*>(smem + kLdsConst)[i] = reinterpret_cast<const f32x4*>(a.wconst)[i];

    Pipe p;
    p.ck = 0;
    p.src_next = 0;
    p.n_chunks = XYZ ? (FAST ? kXFStreamChunks : kXStreamChunks) : (FAST ? kFStreamChunks : kHStreamChunks);
    p.wbase = reinterpret_cast<const char*>(a.wstream);
    p.voff = wave * (kHCQ / 4 * kQuadBytes) + lane * 16;
    p.wave_lds = wave * (kHCQ / 4 * kQuadBytes);
    __syncthreads();
    // pipeline prologue: chunks 0 .. R-2 in flight, chunk 0 landed for everyone
#pragma unroll
    for (int c = 0; c < kHRing - 1; ++c) {
        p.cur_src = p.wbase + (size_t)p.src_next * kHChunkBytes;
        p.cur_dst = kLdsRing + c * kHChunkBytes + p.wave_lds;
        p.src_next += 1;
#pragma unroll
        for (int j = 0; j < kHCQ / 4; ++j) dma_piece(p.cur_src, p.voff + j * kQuadBytes, p.cur_dst + j * kQuadBytes);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kHCQ / 4) * (kHRing - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    frag4 xh[16], xl[16], nh[14], nl[14], peh[3], pel[3], dh[2], dl[2];
    f32x16 accs[4];
    float xc[64];
    float sigma_raw = 0.f;

    unsigned long long t0 = 0, t1 = 0, acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)t0; (void)t1; (void)acc_t;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        STAMP(t0);
        const long long m = tile * 128 + wave * 32 + j;
        const bool valid = m < a.M;
        const long long mm = valid ? m : a.M - 1;
        float px, py, pz, dx, dy, dz;
        if (a.mode == 0) {
            const long long ray = mm / a.S;
            const f32x4 o = *reinterpret_cast<const f32x4*>(a.in_a + ray * 4);
            const f32x4 d = *reinterpret_cast<const f32x4*>(a.in_b + ray * 4);
            const float zz = a.z[mm];
            px = __fadd_rn(o[0], __fmul_rn(d[0], zz));
            py = __fadd_rn(o[1], __fmul_rn(d[1], zz));
            pz = __fadd_rn(o[2], __fmul_rn(d[2], zz));
            dx = d[0]; dy = d[1]; dz = d[2];
        } else {
            px = a.in_a[mm * 3 + 0]; py = a.in_a[mm * 3 + 1]; pz = a.in_a[mm * 3 + 2];
            if constexpr (XYZ) { dx = dy = dz = 0.f; }        // no view directions in this network (may be null)
            else { dx = a.in_b[mm * 3 + 0]; dy = a.in_b[mm * 3 + 1]; dz = a.in_b[mm * 3 + 2]; }
        }
        const float kPi = 3.1415927410125732f;
        // 24 slots per lane half: h=0: sin of the 15 angles, x, y, z; h=1: cos of the 15 angles
        float pv[24];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = c == 0 ? px : c == 1 ? py : pz;
            if constexpr (FAST) {
                sin_ladder_fp16_modes<kLx>(v * kPi, h, &pv[c * kLx]);     // angle doubling: single-pass fp16 modes only
            } else {
#pragma unroll
                for (int k = 0; k < kLx; ++k) pv[c * kLx + k] = sin_shifted(v * (kPi * (float)(1 << k)), h);
            }
        }
        pv[15] = h ? 0.f : px; pv[16] = h ? 0.f : py; pv[17] = h ? 0.f : pz;
#pragma unroll
        for (int i = 18; i < 24; ++i) pv[i] = 0.f;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            float t8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t8[e] = pv[n * 8 + e];
            split8<FAST>(t8, peh[n], pel[n]);
        }
        float dv[16];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = c == 0 ? dx : c == 1 ? dy : dz;
            if constexpr (FAST) {
                sin_ladder_fp16_modes<kLd>(v * kPi, h, &dv[c * kLd]);
            } else {
#pragma unroll
                for (int k = 0; k < kLd; ++k) dv[c * kLd + k] = sin_shifted(v * (kPi * (float)(1 << k)), h);
            }
        }
#pragma unroll
        for (int i = 12; i < 16; ++i) dv[i] = 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float t8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t8[e] = dv[n * 8 + e];
            split8<FAST>(t8, dh[n], dl[n]);
        }

        // stash rows of this lane's sample (rows beyond M exist: the buffers are padded to whole 128-row tiles)
        auto st_of = [&](int l) -> float* {
            if constexpr (!STASH) return nullptr;
            else {
                // row offset in elements (st_ld[l] elements per row): fp32 elements, or fp16 ones under FAST (the pointer
                // arithmetic below is in floats, so halve it there)
                // fragment-major rows (frag_layout.h::frag_index): the wave's 32 samples are one block, lane (h, j) owns
                // the 4-element slot h * 32 + j of every 8-feature group; a feature offset c (c % 8 == 0) is 32 c elements
                const long long row0 = a.diag_wrap ? (long long)(blockIdx.x & 63) * 128 + wave * 32 : tile * 128 + wave * 32;
                const long long off = row0 * (long long)a.st_ld[l] + (h * 32 + j) * 4;
                if constexpr (FAST) return reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(a.st_ptr[l]) + off);
                else return a.st_ptr[l] + off;
            }
        };
        auto mk_of = [&](int l) -> frag4* {
            if constexpr (!STASH) return nullptr;
            else return a.mask_ptr[l] ? reinterpret_cast<frag4*>(a.mask_ptr[l]) + (tile * 128 + wave * 32 + j) * 2 + h
                                      : nullptr;      // no record wanted (layer-by-layer backward)
        };
        frag4 mk_prev = {0u, 0u, 0u, 0u}, mk_cur = {0u, 0u, 0u, 0u};
        frag4 *mk_prev_ptr = nullptr, *mk_cur_ptr = mk_of(0);
        float* st_prev = nullptr;
        float* st_cur = st_of(0);
        STAMP(t1); acc_t[0] += t1 - t0;
        layer_body_h<BODY_PE, false, FAST, STASH>(p, lane16, cb_h, (kHConstBias + 0 * 256) * 4, a.alpha, st_prev, st_cur, mk_prev_ptr, mk_cur_ptr, mk_prev, mk_cur, accs, xh, xl, nh, nl, peh, pel, dh, dl, xc, sigma_raw);
        STAMP(t0); acc_t[1] += t0 - t1;
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            st_prev = st_cur;
            st_cur = st_of(l);
            mk_prev_ptr = mk_cur_ptr; mk_cur_ptr = mk_of(l);
            mk_prev = mk_cur; mk_cur = frag4{0u, 0u, 0u, 0u};
            if (l == 4) {
                layer_body_h<BODY_SKIP, true, FAST, STASH>(p, lane16, cb_h, (kHConstBias + 4 * 256) * 4, a.alpha, st_prev, st_cur, mk_prev_ptr, mk_cur_ptr, mk_prev, mk_cur, accs, xh, xl, nh, nl, peh, pel, dh, dl, xc, sigma_raw);
                STAMP(t1); acc_t[3] += t1 - t0; t0 = t1;
            } else {
                layer_body_h<BODY_HID, true, FAST, STASH>(p, lane16, cb_h, (kHConstBias + l * 256) * 4, a.alpha, st_prev, st_cur, mk_prev_ptr, mk_cur_ptr, mk_prev, mk_cur, accs, xh, xl, nh, nl, peh, pel, dh, dl, xc, sigma_raw);
                STAMP(t1); acc_t[2] += t1 - t0; t0 = t1;
            }
        }
        if constexpr (XYZ) {
            // (training: stash slot 8 = the extra 256-wide layer, slot 9 = the 128-wide one; each body finishes its
            // predecessor's last tile, so the records rotate as above)
            st_prev = st_cur; st_cur = st_of(8);
            mk_prev_ptr = mk_cur_ptr; mk_cur_ptr = mk_of(8);
            mk_prev = mk_cur; mk_cur = frag4{0u, 0u, 0u, 0u};
            layer_body_h<BODY_HIDSIG, true, FAST, STASH, 0>(p, lane16, cb_h, kXConstBiasSig * 4, a.alpha, st_prev, st_cur, mk_prev_ptr, mk_cur_ptr, mk_prev, mk_cur, accs, xh, xl, nh, nl, peh, pel, dh, dl, xc, sigma_raw);
            st_prev = st_cur; st_cur = st_of(9);
            mk_prev_ptr = mk_cur_ptr; mk_cur_ptr = mk_of(9);
            mk_prev = mk_cur; mk_cur = frag4{0u, 0u, 0u, 0u};
            layer_body_h<BODY_LAST0, true, FAST, STASH, 1>(p, lane16, cb_h, kXConstBias9 * 4, a.alpha, st_prev, st_cur, mk_prev_ptr, mk_cur_ptr, mk_prev, mk_cur, accs, xh, xl, nh, nl, peh, pel, dh, dl, xc, sigma_raw);
        } else {
            st_prev = st_cur;
            st_cur = st_of(8);
            mk_prev_ptr = mk_cur_ptr; mk_cur_ptr = mk_of(8);
            mk_prev = mk_cur; mk_cur = frag4{0u, 0u, 0u, 0u};
            layer_body_h<BODY_LAST, true, FAST, STASH>(p, lane16, cb_h, kHConstBias8 * 4, a.alpha, st_prev, st_cur, mk_prev_ptr, mk_cur_ptr, mk_prev, mk_cur, accs, xh, xl, nh, nl, peh, pel, dh, dl, xc, sigma_raw);
            STAMP(t1); acc_t[4] += t1 - t0;
        }
        // rgb head (128 -> 3) on the VALU in fp32
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                constexpr int kW = XYZ ? kXConstWrgb : kHConstWrgb;
                const f32x4 w0 = lds_read4(cb_h + (kW + 0 * 128 + t * 32 + g * 8) * 4);
                const f32x4 w1 = lds_read4(cb_h + (kW + 1 * 128 + t * 32 + g * 8) * 4);
                const f32x4 w2 = lds_read4(cb_h + (kW + 2 * 128 + t * 32 + g * 8) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = xc[t * 16 + g * 4 + e];
                    o0 = fmaf(w0[e], x, o0);
                    o1 = fmaf(w1[e], x, o1);
                    o2 = fmaf(w2[e], x, o2);
                }
            }
        }
        o0 += __shfl_xor(o0, 32);
        o1 += __shfl_xor(o1, 32);
        o2 += __shfl_xor(o2, 32);
        const f32x4 bh = lds_read4(kLdsConst + (XYZ ? kXConstBHead : kHConstBHead) * 4);
        if (valid && h == 0) {
            f32x4 out;
            out[0] = o0 + bh[0]; out[1] = o1 + bh[1]; out[2] = o2 + bh[2]; out[3] = sigma_raw;
            *reinterpret_cast<f32x4*>(a.raw + m * 4) = out;
            // overflow / NaN watch (the f16x3 mode saturates above 65504): count, never hide
            const float chk = out[0] + out[1] + out[2] + out[3];
            if (a.nonfinite && !(fabsf(chk) <= 3.0e38f)) atomicAdd(a.nonfinite, 1ull);
        }
        STAMP(t0); acc_t[5] += t0 - t1; acc_t[6] += 1;
    }
#ifdef NERF_STAMPS
    if (blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 8; ++i) g_stamps_h[i] = acc_t[i];
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(256, 1) void mlp_f16x3_kernel(const MlpArgs a) { mlp_f16_body<false>(a); }
__global__ __launch_bounds__(256, 1) void mlp_f16_kernel(const MlpArgs a) { mlp_f16_body<true>(a); }
__global__ __launch_bounds__(256, 1) void mlp_f16x3_stash_kernel(const MlpArgs a) { mlp_f16_body<false, true>(a); }
__global__ __launch_bounds__(256, 1) void mlp_f16_stash_kernel(const MlpArgs a) { mlp_f16_body<true, true>(a); }
__global__ __launch_bounds__(256, 1) void mlp_f16x3_xyz_kernel(const MlpArgs a) { mlp_f16_body<false, false, true>(a); }
__global__ __launch_bounds__(256, 1) void mlp_f16_xyz_kernel(const MlpArgs a) { mlp_f16_body<true, false, true>(a); }
__global__ __launch_bounds__(256, 1) void mlp_f16x3_xyz_stash_kernel(const MlpArgs a) { mlp_f16_body<false, true, true>(a); }
__global__ __launch_bounds__(256, 1) void mlp_f16_xyz_stash_kernel(const MlpArgs a) { mlp_f16_body<true, true, true>(a); }

void launch_mlp_f16x3_stash(const MlpArgs& a, int num_cus, hipStream_t stream, bool single_pass, bool xyz_only) {
    if (a.M <= 0) return;
    const long long ntiles = (a.M + 127) / 128;
    const int grid = (int)(ntiles < (long long)num_cus ? ntiles : (long long)num_cus);
    if (xyz_only) {
        if (single_pass) hipLaunchKernelGGL(mlp_f16_xyz_stash_kernel, dim3(grid), dim3(256), kLdsTotal, stream, a);
        else hipLaunchKernelGGL(mlp_f16x3_xyz_stash_kernel, dim3(grid), dim3(256), kLdsTotal, stream, a);
        return;
    }
    if (single_pass) hipLaunchKernelGGL(mlp_f16_stash_kernel, dim3(grid), dim3(256), kLdsTotal, stream, a);
    else hipLaunchKernelGGL(mlp_f16x3_stash_kernel, dim3(grid), dim3(256), kLdsTotal, stream, a);
}

void launch_mlp_f16x3(const MlpArgs& a, int num_cus, hipStream_t stream, bool single_pass, bool xyz_only) {
    if (a.M <= 0) return;
    const long long ntiles = (a.M + 127) / 128;
    const int grid = (int)(ntiles < (long long)num_cus ? ntiles : (long long)num_cus);
    if (xyz_only) {
        if (single_pass) hipLaunchKernelGGL(mlp_f16_xyz_kernel, dim3(grid), dim3(256), kLdsTotal, stream, a);
        else hipLaunchKernelGGL(mlp_f16x3_xyz_kernel, dim3(grid), dim3(256), kLdsTotal, stream, a);
        return;
    }
    if (single_pass) hipLaunchKernelGGL(mlp_f16_kernel, dim3(grid), dim3(256), kLdsTotal, stream, a);
    else hipLaunchKernelGGL(mlp_f16x3_kernel, dim3(grid), dim3(256), kLdsTotal, stream, a);
}

#ifdef NERF_STAMPS
extern "C" void nerf_debug_read_stamps_h(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_h), sizeof(unsigned long long) * 16);
}
#endif

void mlp_f16x3_set_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_f16x3_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_f16_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_f16x3_stash_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_f16_stash_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_f16x3_xyz_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_f16_xyz_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_f16x3_xyz_stash_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_f16_xyz_stash_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
}

// ------------------------------------------------------------------------------------------------
// Host-side packing: blob (Keras get_weights() order) -> fp16 hi/lo fragment stream + fp32 constants
// ------------------------------------------------------------------------------------------------
namespace {
// fp32 -> fp16 round-to-nearest-even and back, by bit manipulation (no host _Float16 runtime needed)
uint16_t f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (x > 0x7F800000u ? 0x200u : 0));   // inf / nan
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                     // overflow -> inf
    if (x < 0x33000001u) return (uint16_t)sign;                                                  // < 2^-25 -> 0
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
    int shift;
    uint32_t base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }          // subnormal half
    else { shift = 13; base = (uint32_t)(e + 15) << 10; m &= 0x7FFFFFu; }
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1))) q += 1;   // may carry into the exponent: correct
    return (uint16_t)(sign | (base + q));
}
float f16_to_f32(uint16_t hbits) {
    const uint32_t sign = (uint32_t)(hbits & 0x8000u) << 16;
    const uint32_t e = (hbits >> 10) & 0x1F, m = hbits & 0x3FF;
    uint32_t x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { float f = (float)m * 5.9604644775390625e-08f; memcpy(&x, &f, 4); x |= sign; }   // m * 2^-24
    } else if (e == 31) x = sign | 0x7F800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}
int h_pe_row(int v, int h) {   // slot v (0..23) of lane half h -> row of the (33, .) kernel; -1 = pad
    if (v < 15) { const int c = v / 5, k = v % 5; return c * 11 + 1 + 2 * k + h; }
    if (v < 18 && h == 0) return (v - 15) * 11;
    return -1;
}
int h_hid_row(int n, int e, int h) { const int t = n >> 1, s = n & 1; return 32 * t + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3); }
int h_dir_row(int v, int h, int n_angles) {   // slot v (0..15) -> row of the dir block (24 or 16 rows), -1 = pad
    if (v >= 12) return -1;
    const int c = v / 4, k = v % 4;
    if (n_angles == 2) return c * 8 + 2 * k + h;
    if (c == 1) return -1;                       // n_angles == 1: (x,z) only, src/UtilsCV.py:134-135
    return (c == 0 ? 0 : 1) * 8 + 2 * k + h;
}
struct HLayer { const float* k; const float* b; int in, out; };
}  // namespace

// The packing as a pure index map: EmitW(pos_hi, pos_lo (-1 in the hi-only stream), src) for every 16-bit slot pair of
// the stream, EmitC(pos, src) for every float of the constant region; src = index into the blob, -1 = zero padding.
template <class EmitW, class EmitC>
static void pack_f16_map(int n_angles, bool hi_only, EmitW emit_w, EmitC emit_c) {
    const int kd = 256 + 8 * (n_angles + 1);
    const bool xyz_only = n_angles == 0;
    // Keras creation order; xyz-only (src/NeRF.py:248-288): ..., 8: 256 -> 256, 9: 256 -> 128, 10: 128 -> 3, 11: 256 -> 1
    const int shapes_dir[12][2] = {{33, 256}, {256, 256}, {256, 256}, {256, 256}, {289, 256}, {256, 256},
                                   {256, 256}, {256, 256}, {kd, 128}, {128, 3}, {kd, 1}, {0, 0}};
    const int shapes_xyz[12][2] = {{33, 256}, {256, 256}, {256, 256}, {256, 256}, {289, 256}, {256, 256},
                                   {256, 256}, {256, 256}, {256, 256}, {256, 128}, {128, 3}, {256, 1}};
    const int (*shapes)[2] = xyz_only ? shapes_xyz : shapes_dir;
    struct Lay { long long k, b; int in, out; } L[12];
    long long off = 0;
    for (int i = 0; i < (xyz_only ? 12 : 11); ++i) {
        L[i].in = shapes[i][0]; L[i].out = shapes[i][1];
        L[i].k = off; off += (long long)L[i].in * L[i].out;
        L[i].b = off; off += L[i].out;
    }
    size_t chunk = 0;
    auto emit_body = [&](int layer, int body) {
        const int NU = body == BODY_LAST ? kHTilesLast : body == BODY_HIDSIG ? 9 : body == BODY_LAST0 ? 4 : 8;
        const int NSTEP = body == BODY_PE ? kHStepsPE : (body == BODY_HID || body == BODY_HIDSIG || body == BODY_LAST0) ? kHStepsHid
                          : body == BODY_SKIP ? kHStepsPE + kHStepsHid : kHStepsHid + kHStepsDir;
        const long long b0 = (long long)chunk * (kHChunkBytes / 2);
        for (int u = 0; u < NU; ++u)
            for (int n = 0; n < NSTEP; ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int i = lane & 31, h = lane >> 5;
                        int row;
                        if (body == BODY_PE) row = h_pe_row(n * 8 + e, h);
                        else if (body == BODY_HID || body == BODY_HIDSIG || body == BODY_LAST0) row = h_hid_row(n, e, h);
                        else if (body == BODY_SKIP) {
                            if (n < kHStepsPE) row = h_pe_row(n * 8 + e, h);
                            else row = kXyzDim + h_hid_row(n - kHStepsPE, e, h);
                        } else {
                            if (n < kHStepsHid) row = h_hid_row(n, e, h);
                            else { const int r = h_dir_row((n - kHStepsHid) * 8 + e, h, n_angles); row = r < 0 ? -1 : kHidden + r; }
                        }
                        long long src = -1;
                        if (row >= 0) {
                            if (body == BODY_LAST && u == kHTilesLast - 1) src = i == 0 ? L[10].k + row : -1;   // sigma row
                            else if (body == BODY_HIDSIG && u == 0) src = i == 0 ? L[11].k + row : -1;           // leading sigma row
                            else if (body == BODY_HIDSIG) src = L[layer].k + (long long)row * L[layer].out + 32 * (u - 1) + i;
                            else src = L[layer].k + (long long)row * L[layer].out + 32 * u + i;
                        }
                        if (hi_only) {
                            emit_w(b0 + (long long)(u * NSTEP + n) * (kQuadBytes / 2) + lane * 8 + e, -1LL, src);
                        } else {
                            const long long q = (long long)(u * NSTEP + n) * 2;
                            emit_w(b0 + (q + 0) * (kQuadBytes / 2) + lane * 8 + e, b0 + (q + 1) * (kQuadBytes / 2) + lane * 8 + e, src);
                        }
                    }
        chunk += (NU * NSTEP * (hi_only ? 1 : 2) + kHCQ - 1) / kHCQ;
    };
    emit_body(0, BODY_PE);
    for (int l = 1; l <= 3; ++l) emit_body(l, BODY_HID);
    emit_body(4, BODY_SKIP);
    for (int l = 5; l <= 7; ++l) emit_body(l, BODY_HID);
    for (int l = 0; l < 8; ++l)
        for (int f = 0; f < 256; ++f) emit_c(kHConstBias + l * 256 + f, L[l].b + f);
    if (xyz_only) {
        emit_body(8, BODY_HIDSIG);
        emit_body(9, BODY_LAST0);
        emit_c(kXConstBiasSig + 0, L[11].b);
        for (int f = 0; f < 256; ++f) emit_c(kXConstBias8 + f, L[8].b + f);
        for (int f = 0; f < 128; ++f) emit_c(kXConstBias9 + f, L[9].b + f);
        for (int c = 0; c < 3; ++c)
            for (int f = 0; f < 128; ++f) emit_c(kXConstWrgb + c * 128 + f, L[10].k + f * 3 + c);
        for (int c = 0; c < 3; ++c) emit_c(kXConstBHead + c, L[10].b + c);
        for (int f = 0; f < 256; ++f) emit_c(kXConstWsig + f, L[11].k + f);
        return;
    }
    emit_body(8, BODY_LAST);
    for (int f = 0; f < 128; ++f) emit_c(kHConstBias8 + f, L[8].b + f);
    emit_c(kHConstBiasSig + 0, L[10].b);
    for (int c = 0; c < 3; ++c)
        for (int f = 0; f < 128; ++f) emit_c(kHConstWrgb + c * 128 + f, L[9].k + f * 3 + c);
    for (int c = 0; c < 3; ++c) emit_c(kHConstBHead + c, L[9].b + c);
}

static void pack_weights_f16_impl(const float* blob, int n_angles, void* stream_out, float* const_out, bool hi_only) {
    memset(stream_out, 0, n_angles == 0 ? (hi_only ? kStreamBytesF16HiXyz : kStreamBytesF16Xyz)
                                        : (hi_only ? kStreamBytesF16Hi : kStreamBytesF16));
    memset(const_out, 0, kConstBytes);
    uint16_t* base = reinterpret_cast<uint16_t*>(stream_out);
    pack_f16_map(n_angles, hi_only,
                 [&](long long ph, long long pl, long long src) {
                     const float w = src < 0 ? 0.f : blob[src];
                     const uint16_t hi = f32_to_f16(w);
                     base[ph] = hi;
                     if (pl >= 0) base[pl] = f32_to_f16(w - f16_to_f32(hi));
                 },
                 [&](long long pos, long long src) { const_out[pos] = blob[src]; });
}

// gather tables of the 3-pass stream for the device-side re-pack (the trainer's forward runs on this kernel and its
// weights change every step): stream_idx[slot] = 2 * (src + 1) + is_lo, const_idx[float] = src + 1; 0 = padding
size_t f16_stream_bytes(int n_angles, bool hi_only) {
    if (n_angles == 0) return hi_only ? kStreamBytesF16HiXyz : kStreamBytesF16Xyz;
    return hi_only ? kStreamBytesF16Hi : kStreamBytesF16;
}

void build_f16x3_gather(int n_angles, bool hi_only, int32_t* stream_idx /* f16_stream_bytes / 2 */,
                        int32_t* const_idx /*kConstFloats*/) {
    memset(stream_idx, 0, (f16_stream_bytes(n_angles, hi_only) / 2) * sizeof(int32_t));
    memset(const_idx, 0, kConstFloats * sizeof(int32_t));
    pack_f16_map(n_angles, hi_only,
                 [&](long long ph, long long pl, long long src) {
                     if (src < 0) return;
                     stream_idx[ph] = (int32_t)(2 * (src + 1));
                     if (pl >= 0) stream_idx[pl] = (int32_t)(2 * (src + 1) + 1);
                 },
                 [&](long long pos, long long src) { const_idx[pos] = (int32_t)(src + 1); });
}

__global__ void repack_f16x3_kernel(const float* __restrict__ blob, const int32_t* __restrict__ stream_idx,
                                    uint16_t* __restrict__ stream, const int32_t* __restrict__ const_idx,
                                    float* __restrict__ cst, size_t n_slots) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_slots) {
        const int32_t t = stream_idx[i];
        uint16_t out = 0;
        if (t != 0) {
            const float w = blob[(t >> 1) - 1];
            const _Float16 hi = (_Float16)w;                       // RNE, as the host packer
            const _Float16 v = (t & 1) ? (_Float16)(w - (float)hi) : hi;
            out = __builtin_bit_cast(uint16_t, v);
        }
        stream[i] = out;
    }
    if (i < (size_t)kConstFloats) {
        const int32_t t = const_idx[i];
        cst[i] = t ? blob[t - 1] : 0.f;
    }
}

void launch_repack_f16x3(const float* blob, const int32_t* stream_idx, void* stream, const int32_t* const_idx, float* cst,
                         size_t stream_bytes, hipStream_t s) {
    const size_t n = stream_bytes / 2;
    hipLaunchKernelGGL(repack_f16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, blob, stream_idx,
                       reinterpret_cast<uint16_t*>(stream), const_idx, cst, n);
}

void pack_weights_f16x3(const float* blob, int n_angles, void* stream_out, float* const_out) {
    pack_weights_f16_impl(blob, n_angles, stream_out, const_out, false);
}
void pack_weights_f16(const float* blob, int n_angles, void* stream_out, float* const_out) {
    pack_weights_f16_impl(blob, n_angles, stream_out, const_out, true);
}

}  // namespace nerf

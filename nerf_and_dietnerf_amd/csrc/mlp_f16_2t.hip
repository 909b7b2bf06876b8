// mlp_f16_2t.hip -- the single-pass fp16 render kernel (NERF_PRECISION_F16) with TWO 32-sample tiles per wave.
//
// Same reference chain and the same hi-only weight stream / constant block as mlp_f16x3.hip's single-pass mode
// (src/UtilsCV.py:584-599,124-143; src/UtilsNRF.py:52-85; src/NeRF.py:316-339 under the reference's mixed_float16
// policy, src/ExecutionRun.py:220-221).  What changes is the reuse of the weight operand: the one-tile kernel reads
// every A fragment from LDS for ONE MFMA and re-streams the 1.08 MB of weights for every 128 rows -- 8.2 TB/s of
// L2 -> LDS traffic at its measured rate, the wall it sat on.  Here a wave owns 64 samples as
// two independent 32-sample column sets: every A fragment feeds TWO MFMAs and a workgroup covers 256 rows per
// weight pass, halving both the L2 -> LDS stream and the LDS fragment reads per row.
//
// Register budget (one wave per SIMD, 512 registers): the B operands of both sets (2 x 64) live in architectural
// VGPRs, where the VALU packs them; the staging copy of the next layer's operand (2 x 56 dwords) -- written once,
// read once -- is parked in accumulation registers (v_accvgpr_write / _read); the rotating accumulators (3 live of
// 4 per set) and the fragment prefetch ring are AGPRs as before.  The rgb head (128 -> 3) is folded into layer 8's
// epilogue as running sums, so the 64-float xc buffer of the one-tile kernel does not exist twice.
#include "mlp_f16_frag.h"

#ifndef NERF_2T_PE_LADDER
#define NERF_2T_PE_LADDER 1     // positional encodings by angle doubling (nerf_device.h::sin_ladder_fp16_modes)
#endif
// Packed-pair epilogue of the 256-wide layers (round 4): a pair of accumulator registers is converted to an fp16 pair
// FIRST (v_cvt_pk_f16_f32), then bias, alpha and max act on the pair (v_pk_add_f16, v_pk_mul_f16, v_pk_max_f16) -- 1.5
// plain vector instructions per value instead of 3, and what Keras computes under mixed_float16 (Dense output cast to
// fp16, BiasAdd and LeakyReLU in fp16; src/NeRF.py:309-310 under src/ExecutionRun.py:220-221).  The oracle's emulation
// rounds where this rounds (oracle.mlp_forward_fp16(..., packed_epilogue=True)).  0 = the fp32 epilogue of rounds 2-3.
#ifndef NERF_2T_PACKED_EPI
#define NERF_2T_PACKED_EPI 1
#endif
namespace nerf {

constexpr int kLdsBias16 = kLdsTotal;                 // fp16 copy of the biases of layers 0..7 (natural order), built at kernel start
constexpr int kLdsTotal2T = kLdsTotal + 8 * 256 * 2;

namespace {

__device__ __forceinline__ uint32_t park(uint32_t v) {          // VGPR -> AGPR (the value stays in the accumulation file)
    uint32_t a;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
    return a;
}

// One dense layer for both sample sets.  Schedule, deferred epilogue, in-place operand hand-over and the LDS ring
// protocol are those of layer_body_h<.., FAST = true> (mlp_f16x3.hip); every k-step issues the two sets' MFMAs on
// the one fragment it fetched.
template <int BODY, bool PENDING>
__device__ __forceinline__ void layer_body_2t(Pipe& p, uint32_t lane16, uint32_t cb_h, uint32_t b16h, int bias_off_bytes, float alpha,
                                              f32x16 (&accs)[2][4], frag4 (&xh)[2][16], uint32_t (&nh)[2][14][4],
                                              const frag4 (&peh)[2][3], const frag4 (&dh)[2][2], float (&orgb)[2][3],
                                              float (&sigma_raw)[2]) {
    constexpr int NU = BODY == BODY_LAST ? kHTilesLast : 8;
    constexpr int NSTEP = BODY == BODY_PE ? kHStepsPE : BODY == BODY_HID ? kHStepsHid
                          : BODY == BODY_SKIP ? kHStepsPE + kHStepsHid : kHStepsHid + kHStepsDir;
    constexpr int QPU = NSTEP;
    constexpr int NQ = NU * QPU;
#ifndef NERF_KPF_2T
#define NERF_KPF_2T 4
#endif
    constexpr int kPf = NERF_KPF_2T;      // a fragment now lasts two MFMAs (64 cycles): 4 in flight = 256 cycles of lookahead
    f32x4 pf[kPf];
    const int ck0 = p.ck;
    uint32_t rdbase[2];
    rdbase[0] = lane16 + (uint32_t)((ck0 + 0) & (kHRing - 1)) * kHChunkBytes;
    rdbase[1] = lane16 + (uint32_t)((ck0 + 1) & (kHRing - 1)) * kHChunkBytes;
    auto issue_read = [&](auto qc) {
        constexpr int Qa = decltype(qc)::value;
        lds_read_frag_asm<(Qa % kHCQ) * kQuadBytes>(pf[Qa % kPf], rdbase[(Qa / kHCQ) & 1]);
    };
    static_for<0, kPf>([&](auto ic) {
        if constexpr (decltype(ic)::value < NQ) issue_read(ic);
    });

    auto act = [&](float v) -> float {
        const float av = alpha * v;
        float y;
        asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(v), "v"(av));
        return y;
    };
#if NERF_2T_PACKED_EPI
    // registers (r, r + 1) of a finished tile -> LeakyReLU(fp16(acc) + fp16(bias)) as one packed fp16 pair
    const uint32_t alpha2 = pack_h2(alpha, alpha);
    auto act_pair = [&](float v0, float v1, uint32_t bias2) -> uint32_t {
        const uint32_t pk = pack_h2(v0, v1);                                  // v_cvt_pk_f16_f32 (RNE)
        uint32_t t, q, y;
        asm("v_pk_add_f16 %0, %1, %2" : "=v"(t) : "v"(pk), "v"(bias2));
        asm("v_pk_mul_f16 %0, %1, %2" : "=v"(q) : "v"(t), "v"(alpha2));
        asm("v_pk_max_f16 %0, %1, %2" : "=v"(y) : "v"(t), "v"(q));
        return y;
    };
    // fp16 bias pairs of registers 4g .. 4g+3 (lane half h) of tile t of the layer whose fp32 bias block starts at
    // bias_off_bytes: .x = registers (4g, 4g+1), .y = (4g+2, 4g+3)
    // (b16h = kLdsBias16 + h * 8, an opaque VGPR like cb_h; + the layer's block + tile + group as immediate offsets)
    auto bias_pairs = [&](int tile_off64, int g) -> uint2 {
        extern __shared__ __attribute__((aligned(16))) char smem_[];
        return *reinterpret_cast<const uint2*>(smem_ + (b16h + (uint32_t)(bias_off_bytes / 2)) + tile_off64 + g * 16);
    };
    uint2 bq2;
    (void)bq2;
    auto put_pair = [&](auto sc, auto utc, auto rc, uint32_t ph, auto to_x) {
        constexpr int s = decltype(sc)::value;
        constexpr int ut = decltype(utc)::value;
        constexpr int r = decltype(rc)::value;
        constexpr int n = 2 * ut + (r >> 3), d = (r & 7) >> 1;
        if constexpr (decltype(to_x)::value) xh[s][n][d] = ph;
        else nh[s][n][d] = park(ph);
    };
#endif
    // Unlike the one-tile kernel the bias is NOT preloaded as C-in: that keeps a third accumulator live per set (32
    // AGPRs for the two sets) and this kernel has none to spare.  A tile's chain starts from zero and its epilogue
    // adds the bias: four floats (registers 4g .. 4g+3, the same for both sets) fetched per four k-steps.
    f32x4 bq;
    (void)bq;
    // registers r, r+1 (r even) of output tile ut of set s -> one packed dword of the next operand
    auto store_pair = [&](auto sc, auto utc, auto rc, float y0, float y1, auto to_x) {
        constexpr int s = decltype(sc)::value;
        constexpr int ut = decltype(utc)::value;
        constexpr int r = decltype(rc)::value;
        constexpr int n = 2 * ut + (r >> 3), d = (r & 7) >> 1;
        const uint32_t ph = pack_h2(y0, y1);
        if constexpr (decltype(to_x)::value) xh[s][n][d] = ph;
        else nh[s][n][d] = park(ph);
    };

    float ycarry[2] = {0.f, 0.f};
    (void)ycarry; (void)store_pair;
    f32x4 wr0, wr1, wr2;      // rgb head weights of the four features being finished (BODY_LAST)
    (void)wr0; (void)wr1; (void)wr2;
    static_for<0, NU>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        static_for<0, NSTEP>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            constexpr int Q = u * QPU + n;
            constexpr int qc = Q % kHCQ;
            if constexpr (qc == 0 && Q > 0) p.ck += 1;
            if constexpr (qc == kHCQ / 2) {
                constexpr int room = (NQ - 1 - Q) / 2;
                pipe_sync_c<kHCQ, kHRing, (room + 1 < kHCQ / 4 ? room + 1 : kHCQ / 4)>(p);
            }
            if constexpr (qc > kHCQ / 2 && (qc - kHCQ / 2) % 2 == 0) {
                constexpr int Qs = Q - (qc - kHCQ / 2);
                constexpr bool tail = (NQ - 1 - Qs) / 2 + 1 < kHCQ / 4;
                pipe_piece_c<(qc - kHCQ / 2) / 2, tail>(p);
            }
            lds_wait_frag_asm<(NQ - Q >= kPf ? kPf - 1 : NQ - Q - 1)>(pf[Q % kPf]);
            const h8 a_hi = __builtin_bit_cast(h8, pf[Q % kPf]);
            if constexpr (Q + kPf < NQ) {
                constexpr int Qn = Q + kPf;
                if constexpr (Qn % kHCQ == 0)
                    rdbase[(Qn / kHCQ) & 1] = lane16 + (uint32_t)((ck0 + Qn / kHCQ) & (kHRing - 1)) * kHChunkBytes;
                // the slot being refilled was consumed by THIS k-step: its value is already in a_hi
                issue_read(std::integral_constant<int, Qn>{});
            }
            constexpr bool kPend = (u == 0) && PENDING && n < 8;
            constexpr bool kPrevS = (u > 0) && BODY != BODY_PE && NSTEP >= 16 && n < 16;
            constexpr int et = kPend ? 7 : (u > 0 ? u - 1 : 0);
            constexpr bool kXc = BODY == BODY_LAST && kPrevS;
            static_for<0, 2>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                f32x16& acc = accs[s][u & 3];
                frag4 bh_;
                if constexpr (BODY == BODY_PE) bh_ = peh[s][n];
                else if constexpr (BODY == BODY_HID) bh_ = xh[s][n];
                else if constexpr (BODY == BODY_SKIP) {
                    if constexpr (n < kHStepsPE) bh_ = peh[s][n];
                    else bh_ = xh[s][n - kHStepsPE];
                } else {
                    if constexpr (n < kHStepsHid) bh_ = xh[s][n];
                    else bh_ = dh[s][n - kHStepsHid];
                }
                if constexpr (n == 0) {
                    f32x16 zero;
#pragma unroll
                    for (int i = 0; i < 16; ++i) zero[i] = 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, __builtin_bit_cast(h8, bh_), zero, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, __builtin_bit_cast(h8, bh_), acc, 0, 0, 0);
                }
            });
            // bias of the registers this k-step (and the next ones) finish
#if NERF_2T_PACKED_EPI
            if constexpr (kPend && (n & 1) == 0) bq2 = bias_pairs(-64, n >> 1);                  // previous layer's tile 7
            else if constexpr (kPrevS && !kXc && (n & 3) == 0) bq2 = bias_pairs(et * 64, n >> 2);
            else if constexpr (kXc && (n & 3) == 0) bq = lds_read4(cb_h + bias_off_bytes + et * 128 + (n >> 2) * 32);
#else
            if constexpr (kPend && (n & 1) == 0) bq = lds_read4(cb_h + bias_off_bytes - 128 + (n >> 1) * 32);   // previous layer's tile 7
            else if constexpr (kPrevS && (n & 3) == 0) bq = lds_read4(cb_h + bias_off_bytes + et * 128 + (n >> 2) * 32);
#endif
            // rgb head weights for registers n..n+3 of the tile being finished (same for both sets)
            if constexpr (kXc && (n & 3) == 0) {
                wr0 = lds_read4(cb_h + (kHConstWrgb + 0 * 128 + et * 32 + (n >> 2) * 8) * 4);
                wr1 = lds_read4(cb_h + (kHConstWrgb + 1 * 128 + et * 32 + (n >> 2) * 8) * 4);
                wr2 = lds_read4(cb_h + (kHConstWrgb + 2 * 128 + et * 32 + (n >> 2) * 8) * 4);
            }
            static_for<0, 2>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                f32x16& prv = accs[s][(u + 3) & 3];
#if NERF_2T_PACKED_EPI
                if constexpr (kPend) {
                    constexpr int er = 2 * n;
                    put_pair(sc, std::integral_constant<int, 7>{}, std::integral_constant<int, er>{},
                             act_pair(prv[er], prv[er + 1], (n & 1) ? bq2.y : bq2.x), std::true_type{});
                } else if constexpr (kPrevS && !kXc) {
                    if constexpr ((n & 1) == 1)
                        put_pair(sc, std::integral_constant<int, et>{}, std::integral_constant<int, n - 1>{},
                                 act_pair(prv[n - 1], prv[n], (n & 2) ? bq2.y : bq2.x), std::false_type{});
                } else if constexpr (kXc) {
                    const float y = act(prv[n] + bq[n & 3]);          // layer 8 feeds the fp32 rgb head: fp32 epilogue
                    orgb[s][0] = fmaf(wr0[n & 3], y, orgb[s][0]);
                    orgb[s][1] = fmaf(wr1[n & 3], y, orgb[s][1]);
                    orgb[s][2] = fmaf(wr2[n & 3], y, orgb[s][2]);
                    asm volatile("" : "+v"(orgb[s][0]), "+v"(orgb[s][1]), "+v"(orgb[s][2]));
                }
#else
                if constexpr (kPend) {
                    constexpr int er = 2 * n;
                    const float y0 = act(prv[er] + bq[er & 3]), y1 = act(prv[er + 1] + bq[(er & 3) + 1]);
                    store_pair(sc, std::integral_constant<int, 7>{}, std::integral_constant<int, er>{}, y0, y1, std::true_type{});
                } else if constexpr (kPrevS) {
                    const float y = act(prv[n] + bq[n & 3]);
                    if constexpr (kXc) {
                        orgb[s][0] = fmaf(wr0[n & 3], y, orgb[s][0]);
                        orgb[s][1] = fmaf(wr1[n & 3], y, orgb[s][1]);
                        orgb[s][2] = fmaf(wr2[n & 3], y, orgb[s][2]);
                        // The running sums are pinned to this k-step: left free, hipcc defers the second set's whole
                        // 64-term chain to the end of the tile (activations parked in spare AGPRs, all 48 weight registers
                        // kept live) -- the long-lived values that the exec-mask bug described at the tile's end corrupted.
                        asm volatile("" : "+v"(orgb[s][0]), "+v"(orgb[s][1]), "+v"(orgb[s][2]));
                    } else if constexpr ((n & 1) == 0) ycarry[s] = y;
                    else store_pair(sc, std::integral_constant<int, et>{}, std::integral_constant<int, n - 1>{}, ycarry[s], y, std::false_type{});
                }
#endif
                if constexpr (u == 0 && PENDING) {
                    if constexpr (n == 8) { xh[s][12] = frag4{nh[s][12][0], nh[s][12][1], nh[s][12][2], nh[s][12][3]}; }
                    if constexpr (n == 9) { xh[s][13] = frag4{nh[s][13][0], nh[s][13][1], nh[s][13][2], nh[s][13][3]}; }
                }
                if constexpr (BODY == BODY_PE && u > 0 && n == 0) {
                    static_for<0, 8>([&](auto pc) {
                        constexpr int r = 2 * decltype(pc)::value;
#if NERF_2T_PACKED_EPI
                        const uint2 b2 = bias_pairs((u - 1) * 64, r >> 2);
                        const uint32_t ph = act_pair(prv[r], prv[r + 1], (r & 2) ? b2.y : b2.x);
                        if constexpr (u - 1 <= 5) put_pair(sc, std::integral_constant<int, u - 1>{}, std::integral_constant<int, r>{}, ph, std::true_type{});
                        else put_pair(sc, std::integral_constant<int, u - 1>{}, std::integral_constant<int, r>{}, ph, std::false_type{});
#else
                        const f32x4 b4 = lds_read4(cb_h + bias_off_bytes + (u - 1) * 128 + (r >> 2) * 32);
                        const float z0 = act(prv[r] + b4[r & 3]), z1 = act(prv[r + 1] + b4[(r & 3) + 1]);
                        if constexpr (u - 1 <= 5) store_pair(sc, std::integral_constant<int, u - 1>{}, std::integral_constant<int, r>{}, z0, z1, std::true_type{});
                        else store_pair(sc, std::integral_constant<int, u - 1>{}, std::integral_constant<int, r>{}, z0, z1, std::false_type{});
#endif
                    });
                }
                if constexpr ((BODY == BODY_HID || BODY == BODY_SKIP) && u == NU - 1) {
                    constexpr int m = BODY == BODY_SKIP ? n - kHStepsPE : n;
                    if constexpr (m >= 1 && m - 1 < 12)
                        xh[s][m - 1] = frag4{nh[s][m - 1][0], nh[s][m - 1][1], nh[s][m - 1][2], nh[s][m - 1][3]};
                }
            });
            // (A scheduling barrier per k-step was needed while the bias rode as C-in -- the scheduler hoisted a whole
            // tile's accumulator reads to the top of the body and spilled -- and costs 4 % now that nothing spills.)
#ifdef NERF_2T_SCHED_BARRIER
            __builtin_amdgcn_sched_barrier(0);
#endif
        });
    });
    if constexpr (BODY == BODY_LAST) {
        const float bsig = lds_read4(cb_h + kHConstBiasSig * 4)[0];      // row 0 of the sigma tile's bias block
        sigma_raw[0] = accs[0][(NU - 1) & 3][0] + bsig;
        sigma_raw[1] = accs[1][(NU - 1) & 3][0] + bsig;
    }
    if constexpr (NQ % kHCQ != 0 && NQ % kHCQ <= kHCQ / 2) pipe_sync_c<kHCQ, kHRing, 1>(p);
    p.ck += 1;
}

__device__ __forceinline__ void pack8(const float (&v)[8], frag4& hi) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) hi[e >> 1] = pack_h2(v[e], v[e + 1]);
}

}  // namespace

__global__ __launch_bounds__(256, 1) void mlp_f16_2t_kernel(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const uint32_t lane16 = kLdsRing + lane * 16;
    // The constant block sits above 128 KiB of ring: with cb_h a visible expression hipcc folds kLdsConst into every
    // constant-block address, none of which then fits the 16-bit offset field of a ds_read -- it materialises one base
    // VGPR per distinct address (30+ registers, hoisted out of the tile loop) and, at this kernel's register limit, spills
    // them to scratch (reloaded behind `s_waitcnt vmcnt(0)`, i.e. behind the whole DMA queue).  Opaque base + immediate
    // offsets instead.
    uint32_t cb_h = kLdsConst + h * 16;
    asm volatile("" : "+v"(cb_h));
    uint32_t b16h = kLdsBias16 + h * 8;            // fp16 bias pairs of this lane half (packed epilogue)
    asm volatile("" : "+v"(b16h));

    const long long ntiles = (a.M + 255) / 256;
    if ((long long)blockIdx.x >= ntiles) return;

    for (int i = tid; i < kHConstFloats / 4; i += 256)
        reinterpret_cast<f32x4*>(smem + kLdsConst)[i] = reinterpret_cast<const f32x4*>(a.wconst)[i];
#if NERF_2T_PACKED_EPI
    for (int i = tid; i < 8 * 256; i += 256)       // fp16 biases of layers 0..7 (RNE), same natural order as the fp32 block
        reinterpret_cast<_Float16*>(smem + kLdsBias16)[i] = (_Float16)a.wconst[kHConstBias + i];
#endif

    Pipe p;
    p.ck = 0;
    p.src_next = 0;
    p.n_chunks = kFStreamChunks;
    p.wbase = reinterpret_cast<const char*>(a.wstream);
    p.voff = wave * (kHCQ / 4 * kQuadBytes) + lane * 16;
    p.wave_lds = wave * (kHCQ / 4 * kQuadBytes);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kHRing - 1; ++c) {
        p.cur_src = p.wbase + (size_t)p.src_next * kHChunkBytes;
        p.cur_dst = kLdsRing + c * kHChunkBytes + p.wave_lds;
        p.src_next += 1;
#pragma unroll
        for (int q = 0; q < kHCQ / 4; ++q) dma_piece(p.cur_src, p.voff + q * kQuadBytes, p.cur_dst + q * kQuadBytes);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kHCQ / 4) * (kHRing - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    frag4 xh[2][16], peh[2][3], dh[2][2];
    uint32_t nh[2][14][4];
    f32x16 accs[2][4];
    float orgb[2][3];
    float sigma_raw[2] = {0.f, 0.f};

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        long long m_[2];
        bool valid_[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const long long m = tile * 256 + wave * 64 + s * 32 + j;
            m_[s] = m;
            valid_[s] = m < a.M;
            const long long mm = valid_[s] ? m : a.M - 1;
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
                dx = a.in_b[mm * 3 + 0]; dy = a.in_b[mm * 3 + 1]; dz = a.in_b[mm * 3 + 2];
            }
            const float kPi = 3.1415927410125732f;
            float pv[24];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = c == 0 ? px : c == 1 ? py : pz;
#if NERF_2T_PE_LADDER
                sin_ladder_fp16_modes<kLx>(v * kPi, h, &pv[c * kLx]);
#else
#pragma unroll
                for (int k = 0; k < kLx; ++k) pv[c * kLx + k] = sin_shifted(v * (kPi * (float)(1 << k)), h);
#endif
            }
            pv[15] = h ? 0.f : px; pv[16] = h ? 0.f : py; pv[17] = h ? 0.f : pz;
#pragma unroll
            for (int i = 18; i < 24; ++i) pv[i] = 0.f;
#pragma unroll
            for (int n = 0; n < 3; ++n) {
                float t8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t8[e] = pv[n * 8 + e];
                pack8(t8, peh[s][n]);
            }
            float dv[16];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = c == 0 ? dx : c == 1 ? dy : dz;
#if NERF_2T_PE_LADDER
                sin_ladder_fp16_modes<kLd>(v * kPi, h, &dv[c * kLd]);
#else
#pragma unroll
                for (int k = 0; k < kLd; ++k) dv[c * kLd + k] = sin_shifted(v * (kPi * (float)(1 << k)), h);
#endif
            }
#pragma unroll
            for (int i = 12; i < 16; ++i) dv[i] = 0.f;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                float t8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t8[e] = dv[n * 8 + e];
                pack8(t8, dh[s][n]);
            }
            orgb[s][0] = orgb[s][1] = orgb[s][2] = 0.f;
        }

        layer_body_2t<BODY_PE, false>(p, lane16, cb_h, b16h, (kHConstBias + 0 * 256) * 4, a.alpha, accs, xh, nh, peh, dh, orgb, sigma_raw);
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            if (l == 4)
                layer_body_2t<BODY_SKIP, true>(p, lane16, cb_h, b16h, (kHConstBias + 4 * 256) * 4, a.alpha, accs, xh, nh, peh, dh, orgb, sigma_raw);
            else
                layer_body_2t<BODY_HID, true>(p, lane16, cb_h, b16h, (kHConstBias + l * 256) * 4, a.alpha, accs, xh, nh, peh, dh, orgb, sigma_raw);
        }
        layer_body_2t<BODY_LAST, true>(p, lane16, cb_h, b16h, kHConstBias8 * 4, a.alpha, accs, xh, nh, peh, dh, orgb, sigma_raw);

        // Both sets' results are finished BEFORE any divergent code, and the divergent part is flat (one store region, one
        // counter region).  hipcc (ROCm 7.2) was caught restoring a VGPR it had saved around a nested divergent region
        // (`if (valid && h == 0) { store; if (not finite) atomicAdd }`) under the INNER region's exec mask: a value that was
        // live across it (one activation of the second set's deferred rgb chain) came back clobbered in exactly the lanes
        // that had taken the branch.  Nothing but the outputs is live across the stores now.
        const f32x4 bh = lds_read4(kLdsConst + kHConstBHead * 4);
        f32x4 outv[2];
        bool bad = false;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float o0 = orgb[s][0], o1 = orgb[s][1], o2 = orgb[s][2];
            o0 += __shfl_xor(o0, 32);
            o1 += __shfl_xor(o1, 32);
            o2 += __shfl_xor(o2, 32);
            outv[s][0] = o0 + bh[0]; outv[s][1] = o1 + bh[1]; outv[s][2] = o2 + bh[2]; outv[s][3] = sigma_raw[s];
            const float chk = outv[s][0] + outv[s][1] + outv[s][2] + outv[s][3];
            bad |= valid_[s] && h == 0 && !(fabsf(chk) <= 3.0e38f);
        }
        if (h == 0) {
            if (valid_[0]) *reinterpret_cast<f32x4*>(a.raw + m_[0] * 4) = outv[0];
            if (valid_[1]) *reinterpret_cast<f32x4*>(a.raw + m_[1] * 4) = outv[1];
        }
        // overflow / NaN watch: count rows (a lane may hold up to two), never hide
        if (a.nonfinite && bad) {
            unsigned long long n_bad = 0;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float chk = outv[s][0] + outv[s][1] + outv[s][2] + outv[s][3];
                n_bad += (valid_[s] && !(fabsf(chk) <= 3.0e38f)) ? 1ull : 0ull;
            }
            atomicAdd(a.nonfinite, n_bad);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

void launch_mlp_f16_2t(const MlpArgs& a, int num_cus, hipStream_t stream) {
    if (a.M <= 0) return;
    const long long ntiles = (a.M + 255) / 256;
    const int grid = (int)(ntiles < (long long)num_cus ? ntiles : (long long)num_cus);
    hipLaunchKernelGGL(mlp_f16_2t_kernel, dim3(grid), dim3(256), kLdsTotal2T, stream, a);
}

void mlp_f16_2t_set_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_f16_2t_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal2T);
}

}  // namespace nerf

// mlp_fp32.hip -- fused positional-encoding + 11-layer NeRF MLP forward, exact fp32 on MFMA.
//
// Replaces, per sample row, the reference chain
//   sample_along_rays          src/UtilsCV.py:584-599
//   get_view_directions        src/UtilsCV.py:124-143
//   positional_encoding_for_*  src/UtilsNeuralRadianceField.py:52-85
//   Keras model call           src/NeRF.py:316-339 (11 Dense, LeakyReLU(alpha))
// and writes raw (M,4) = [r,g,b,sigma] for ray_marching.
//
// Design (MI355X / gfx950):
//   * one workgroup = 4 waves = one wave per SIMD, 512-register budget; each wave owns 32 samples;
//     persistent workgroups (one per CU) walk the 128-sample tiles.
//   * H^T = W^T X^T on v_mfma_f32_32x32x2_f32: output tile = [32 out-features x 32 samples], samples
//     on the lane.  An accumulator register (after bias + LeakyReLU) IS the next layer's B operand
//     for one k-step -- activations never leave the register file; the k order this implies is
//     baked into the host-side weight packing.
//   * weights (2.08 MB / net) stream L2 -> LDS through an 8 x 16 KiB ring by LDS-DMA
//     (global_load_lds_dwordx4, one piece per MFMA gap), one counted vmcnt + one s_barrier per 64
//     MFMAs, shared by the 4 waves; A operands are read back with one ds_read_b128 per 4 MFMAs;
//     output tiles are processed in pairs so two accumulator chains alternate.
//   * heads (128->3, 280->1) on the VALU; positional encoding in-register with a Cody-Waite sincos.
#include "mlp_common.h"

#include <math.h>
#include <string.h>

namespace nerf {

#ifdef NERF_STAMPS
__device__ unsigned long long g_stamps[16];
#endif

// The xyz-only network (get_network_only_xyz, src/NeRF.py:248-288: 12 Dense layers, sigma = Dense(1)(h8), the colour
// branch h8 -> 256 -> 128 -> rgb) on the same skeleton: two more body kinds and its own constant layout.
//   BODY_HSIG  : BODY_HID for the extra 256-wide layer; while its second tile pair runs -- h8 is complete in xin from the
//                end of the first pair (the pending tiles 6, 7) until the last pair's copy-back -- the 256 -> 1 sigma head
//                is accumulated on the VALU, one weight quad per k-quad (the view-direction network does it after layer 8,
//                when xin still holds h8; here layer 8 overwrites it)
//   BODY_LASTX : BODY_LAST without the direction rows (256 -> 128)
enum { BODY_HSIG = 6, BODY_LASTX = 7 };
constexpr int kFXConstBias8 = 2048;      // 256: the extra layer (contiguous with layers 0..7: the bias preload runs ahead)
constexpr int kFXConstBias9 = 2304;      // 128
constexpr int kFXConstWrgb = 2432;       // [3][128]
constexpr int kFXConstBHead = 2816;      // b_r, b_g, b_b, b_sigma
constexpr int kFXConstWsigH = 2820;      // [256]
static_assert(kFXConstWsigH + 256 <= kConstFloats, "xyz-only fp32 constants must fit the LDS carve");
constexpr int kChunksLastX = (4 * kQpuHid) / 16;                                                        // 8
constexpr int kStreamChunksXyz = kChunksPE + 3 * kChunksHid + kChunksSkip + 3 * kChunksHid + kChunksHid + kChunksLastX;   // 142
static_assert((size_t)kStreamChunksXyz * kChunkBytes == kStreamBytesXyzF32, "xyz-only fp32 stream size mismatch");

// One dense layer, u-outer: for each 32-wide output tile run the whole K chain into one accumulator.
//   BODY_PE   : B = xpe                      -> xin   (layer 0)
//   BODY_HID  : B = xin                      -> xin   (layers 1-3, 5-7; via xnext + staged copy-back)
//   BODY_SKIP : B = [xpe, xin]               -> xin   (layer 4)
//   BODY_LAST : B = [xin, xdir], 4 tiles     -> xc    (layer 8)
template <int BODY, bool PENDING>
__device__ __forceinline__ void layer_body(Pipe& p, uint32_t lane16,
                                           uint32_t cb_h, int bias_off_bytes, float alpha, f32x16 (&accs)[4],
                                           float (&xin)[128], float (&xnext)[96], const float (&xpe)[17],
                                           const float (&xdir)[12], float (&xc)[64], float& sig) {
    constexpr bool kHid = BODY == BODY_HID || BODY == BODY_HSIG;
    constexpr bool kLast = BODY == BODY_LAST || BODY == BODY_LASTX;
    // Output tiles are processed in PAIRS (u = 2P, 2P+1): the two accumulator chains alternate MFMA by
    // MFMA, so a dependent v_mfma_f32_32x32x2_f32 is never issued back to back on one accumulator,
    // and both chains share every B operand.  Stream order per pair: quad(2P,q), quad(2P+1,q), q++.
    constexpr int NP = kLast ? 2 : 4;
    constexpr int QPU = BODY == BODY_PE ? kQpuPE : (kHid || BODY == BODY_LASTX) ? kQpuHid
                        : BODY == BODY_SKIP ? kQpuSkip : kQpuLast;
    constexpr int NQ = NP * 2 * QPU;
    // accs[0..1]: pair accumulators of even pairs, accs[2..3]: of odd pairs.  They live across layers:
    // PENDING = the previous layer's last pair (tiles 6,7) still sits raw in accs[2..3] and is finished
    // (in place, xin[96..127]) during this layer's first pair, so no epilogue is exposed at a layer end.
    uint32_t rd = lane16 + (uint32_t)(p.ck & (kRingChunks - 1)) * kChunkBytes;
    f32x4 a_nx = lds_read4(rd);          // A operands are fetched one quad ahead of their MFMAs

    // LeakyReLU(v) = max(v, alpha*v) (0 < alpha < 1) on TWO accumulator registers at a time: one
    // v_pk_mul_f32 + two v_max_f32 (gfx950 has no packed max).  Beside the fp32 MFMA every VALU
    // instruction, packed or not, costs ~5-6 cycles of matrix-pipe time (tools/microbench/
    // valu_cost.hip), so instruction count is what matters.  (asm: the builtin fmaxf adds a
    // canonicalising v_max per element.)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 alpha2 = {alpha, alpha};
    auto store_act2 = [&](auto uc, auto rc, float v0, float v1, auto dc) {
        constexpr int u = decltype(uc)::value;
        constexpr int r = decltype(rc)::value;   // even register index; handles r and r+1
#ifdef NERF_DIAG_NOACT   // timing-only diagnostic: no activation math
        const float y0 = v0, y1 = v1;
#else
        f32x2 v = {v0, v1}, av;
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(av) : "v"(v), "v"(alpha2));
        float y0, y1;
        asm("v_max_f32 %0, %1, %2" : "=v"(y0) : "v"(v0), "v"(av[0]));
        asm("v_max_f32 %0, %1, %2" : "=v"(y1) : "v"(v1), "v"(av[1]));
#endif
        constexpr int dst = decltype(dc)::value;   // 0: xin (in place)  1: xnext  2: xc
        if constexpr (dst == 0) { xin[u * 16 + r] = y0; xin[u * 16 + r + 1] = y1; }
        else if constexpr (dst == 2) { xc[u * 16 + r] = y0; xc[u * 16 + r + 1] = y1; }
        else { xnext[u * 16 + r] = y0; xnext[u * 16 + r + 1] = y1; }
    };
    // where tile u of THIS layer goes: layer 0 reads xpe only (in place), layer 8 feeds the heads,
    // the in-place layers stage tiles 0..5 in xnext (tiles 6,7 are finished by the next layer)
    auto dest_of = [](auto uc) {
        constexpr int u = decltype(uc)::value;
        if constexpr (BODY == BODY_PE) return std::integral_constant<int, 0>{};
        else if constexpr (kLast) return std::integral_constant<int, 2>{};
        else if constexpr (u >= 6) return std::integral_constant<int, 0>{};
        else return std::integral_constant<int, 1>{};
    };
    auto load_bias_pair = [&](int off_bytes, f32x16& d0, f32x16& d1) {
        static_for<0, 4>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const f32x4 b0 = lds_read4(cb_h + off_bytes + g * 32);
            const f32x4 b1 = lds_read4(cb_h + off_bytes + 128 + g * 32);
            d0[4 * g + 0] = b0[0]; d0[4 * g + 1] = b0[1]; d0[4 * g + 2] = b0[2]; d0[4 * g + 3] = b0[3];
            d1[4 * g + 0] = b1[0]; d1[4 * g + 1] = b1[1]; d1[4 * g + 2] = b1[2]; d1[4 * g + 3] = b1[3];
        });
    };

    static_for<0, NP>([&](auto pc) {
        constexpr int P = decltype(pc)::value;
        f32x16& acc0 = accs[(P & 1) * 2 + 0];
        f32x16& acc1 = accs[(P & 1) * 2 + 1];
        f32x16& prv0 = accs[((P + 1) & 1) * 2 + 0];   // previous pair's accumulators (also the next pair's)
        f32x16& prv1 = accs[((P + 1) & 1) * 2 + 1];
        // accumulators start as the bias of their 32 features (C-in of the first MFMA); every pair but the
        // very first of a tile gets it preloaded while the previous pair runs (below)
        if constexpr (P == 0 && !PENDING) load_bias_pair(bias_off_bytes, acc0, acc1);
        static_for<0, QPU>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            f32x4 a4[2];
            static_for<0, 2>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int Q = (P * QPU + q) * 2 + t;
                // mid-chunk sync of the chunk quad Q lives in (p.ck still names that chunk here)
                // the chunk's last sync-point may fall in a short tail: then all 4 pieces go at once
                constexpr int left = NQ - Q;   // quads left in this body including Q
                if constexpr (Q % kChunkQuads == 8) pipe_sync(p, left <= 6);
                // (tail of the body measured from the sync quad: left + distance > 6 <=> not all-at-once)
                if constexpr (Q % kChunkQuads == 10 && left + 2 > 6) pipe_piece(p, 1);
                if constexpr (Q % kChunkQuads == 12 && left + 4 > 6) pipe_piece(p, 2);
                if constexpr (Q % kChunkQuads == 14 && left + 6 > 6) pipe_piece(p, 3);
                a4[t] = a_nx;
                if constexpr (Q + 1 < NQ) {
                    if constexpr ((Q + 1) % kChunkQuads == 0) {
                        p.ck += 1;
                        rd = lane16 + (uint32_t)(p.ck & (kRingChunks - 1)) * kChunkBytes;
                    }
                    a_nx = lds_read4(rd + ((Q + 1) % kChunkQuads) * kQuadBytes);
                }
            });
            static_for<0, 4>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                constexpr int s = 4 * q + e;
                bool live = true;
                float b = 0.f;
                if constexpr (BODY == BODY_PE) {
                    if constexpr (s < 17) b = xpe[s]; else live = false;
                } else if constexpr (kHid || BODY == BODY_LASTX) {
                    b = xin[s];
                } else if constexpr (BODY == BODY_SKIP) {
                    if constexpr (q < kQpuPE) { if constexpr (s < 17) b = xpe[s]; else live = false; }
                    else b = xin[s - 4 * kQpuPE];
                } else {
                    if constexpr (q < kQpuHid) b = xin[s]; else b = xdir[s - 4 * kQpuHid];
                }
                if (live) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0][e], b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1][e], b, acc1, 0, 0, 0);
                }
            });
            if constexpr (BODY == BODY_HSIG && P == 1) {
                // sigma head of the xyz-only network on h8 (= xin, complete during this pair): features of k-quad q
                const f32x4 w = lds_read4(cb_h + (kFXConstWsigH + (q >> 2) * 32 + (q & 3) * 8) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) sig = fmaf(w[e], xin[(q >> 2) * 16 + (q & 3) * 4 + e], sig);
            }
            // Deferred epilogue of the previous pair, spread over this pair's MFMA gaps: the VALU work
            // hides only while each 64-cycle MFMA gap carries a few instructions, so it is dealt out
            // kEpi elements per quad-pair (8 MFMAs) instead of in one block.
            constexpr int kEpi = QPU >= 17 ? 1 : (16 + QPU - 2) / (QPU - 1);   // register PAIRS per q, q >= 1
            if constexpr ((P > 0 || PENDING) && q >= 1) {
                static_for<0, kEpi>([&](auto ic) {
                    constexpr int idx = (q - 1) * kEpi + decltype(ic)::value;   // 0..15 over the pair of tiles
                    if constexpr (idx < 16) {
                        constexpr int r = (idx & 7) * 2;
                        if constexpr (P == 0) {
                            // previous layer's tiles 6,7: in place (their k-steps are first read at quad 24)
                            if constexpr (idx < 8) store_act2(std::integral_constant<int, 6>{}, std::integral_constant<int, r>{}, prv0[r], prv0[r + 1], std::integral_constant<int, 0>{});
                            else store_act2(std::integral_constant<int, 7>{}, std::integral_constant<int, r>{}, prv1[r], prv1[r + 1], std::integral_constant<int, 0>{});
                        } else {
                            if constexpr (idx < 8) store_act2(std::integral_constant<int, 2 * P - 2>{}, std::integral_constant<int, r>{}, prv0[r], prv0[r + 1], dest_of(std::integral_constant<int, 2 * P - 2>{}));
                            else store_act2(std::integral_constant<int, 2 * P - 1>{}, std::integral_constant<int, r>{}, prv1[r], prv1[r + 1], dest_of(std::integral_constant<int, 2 * P - 1>{}));
                        }
                    }
                });
            }
            // the previous pair's accumulators are free once its epilogue is done (q = 16 at the latest):
            // preload the bias of the NEXT pair -- or of the next layer's first pair: the layers' biases are
            // contiguous in the constant region -- so that no chain starts on an LDS round trip
            if constexpr (q == (QPU >= 20 ? 18 : QPU - 1)) {
                if constexpr (P + 1 < NP) load_bias_pair(bias_off_bytes + (P + 1) * 256, prv0, prv1);
                else if constexpr (!kLast) load_bias_pair(bias_off_bytes + 1024, prv0, prv1);
            }
            // last pair of an in-place layer: tile t of xin is dead once its 4 quads (4t..4t+3) are
            // consumed; copy xnext back 4 registers per quad over the following 4 quads
            if constexpr ((kHid || BODY == BODY_SKIP) && P == NP - 1) {
                constexpr int qh = BODY == BODY_SKIP ? q - kQpuPE : q;
                if constexpr (qh >= 4 && qh < 28) {
                    constexpr int t = (qh - 4) >> 2;
                    constexpr int r0 = ((qh - 4) & 3) * 4;
                    static_for<0, 4>([&](auto rc) {
                        constexpr int r = r0 + decltype(rc)::value;
                        xin[t * 16 + r] = xnext[t * 16 + r];
                    });
                }
            }
        });
    });
    if constexpr (kLast) {   // the last layer feeds the heads right away: finish its last pair here
        f32x16& l0 = accs[((NP - 1) & 1) * 2 + 0];
        f32x16& l1 = accs[((NP - 1) & 1) * 2 + 1];
        static_for<0, 8>([&](auto rc) {
            constexpr int r = decltype(rc)::value * 2;
            store_act2(std::integral_constant<int, 2 * NP - 2>{}, std::integral_constant<int, r>{}, l0[r], l0[r + 1], std::integral_constant<int, 2>{});
            store_act2(std::integral_constant<int, 2 * NP - 1>{}, std::integral_constant<int, r>{}, l1[r], l1[r + 1], std::integral_constant<int, 2>{});
        });
    }
    if constexpr (NQ % kChunkQuads != 0 && NQ % kChunkQuads <= 8) pipe_sync(p, true);
    p.ck += 1;  // every body starts on a chunk boundary
}

template <bool XYZ>
__device__ __forceinline__ void mlp_fp32_body(const MlpArgs& a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const uint32_t lane16 = kLdsRing + lane * 16;
    const uint32_t cb_h = kLdsConst + h * 16;
    // (Round 4: hiding this base from the compiler -- asm volatile("" : "+v"(cb_h)), which saves the fp16 kernels one
    // v_add_u32 per constant read, mlp_f16x3.hip -- is NOT applied here: with the opaque base in the layer bodies AND the
    // heads this kernel computes wrong values (all rows, ~3e-2), with either alone it is bit-identical to this build.  An
    // interaction with hipcc's (ROCm 7.2) scheduling / allocation at 512 registers that was not root-caused; the knob
    // below reproduces it: make EXTRA=-DNERF_DIAG_FP32_OPAQUE_CB=3 (1 = heads only, 2 = bodies only: both correct).)
#ifdef NERF_DIAG_FP32_OPAQUE_CB
    uint32_t cb_o = kLdsConst + h * 16;
    asm volatile("" : "+v"(cb_o));
    const uint32_t cb_body = (NERF_DIAG_FP32_OPAQUE_CB & 2) ? cb_o : cb_h;
    const uint32_t cb_head = (NERF_DIAG_FP32_OPAQUE_CB & 1) ? cb_o : cb_h;
#else
    const uint32_t cb_body = cb_h, cb_head = cb_h;
#endif

    const long long ntiles = (a.M + 127) / 128;
    if ((long long)blockIdx.x >= ntiles) return;   // uniform per workgroup

    // constant region -> LDS (once per workgroup)
    for (int i = tid; i < kConstFloats / 4; i += 256)
        reinterpret_cast<f32x4*>(smem + kLdsConst)[i] = reinterpret_cast<const f32x4*>(a.wconst)[i];

    Pipe p;
    p.ck = 0;
    p.src_next = 0;
    p.n_chunks = XYZ ? kStreamChunksXyz : kStreamChunks;
    p.wbase = reinterpret_cast<const char*>(a.wstream);
    p.voff = wave * (4 * kQuadBytes) + lane * 16;
    p.wave_lds = wave * (4 * kQuadBytes);
    __syncthreads();   // const region visible; no DMA in flight yet
    // pipeline prologue: chunks 0 .. R-2 in flight, chunk 0 landed for everyone
#pragma unroll
    for (int c = 0; c < kRingChunks - 1; ++c) {
        p.cur_src = p.wbase + (size_t)p.src_next * kChunkBytes;
        p.cur_dst = kLdsRing + c * kChunkBytes + p.wave_lds;
        p.src_next += 1;
        dma_piece(p.cur_src, p.voff, p.cur_dst);
        dma_piece(p.cur_src, p.voff + kQuadBytes, p.cur_dst + kQuadBytes);
        dma_piece(p.cur_src, p.voff + 2 * kQuadBytes, p.cur_dst + 2 * kQuadBytes);
        dma_piece(p.cur_src, p.voff + 3 * kQuadBytes, p.cur_dst + 3 * kQuadBytes);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kRingChunks - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    float xin[128], xnext[96], xpe[17], xdir[12], xc[64];
    f32x16 accs[4];

    unsigned long long t0 = 0, t1 = 0, acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)t0; (void)t1; (void)acc_t;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        STAMP(t0);
        // ---------------- tile prologue: sample position + encodings ----------------
        const long long m = tile * 128 + wave * 32 + j;
        const bool valid = m < a.M;
        const long long mm = valid ? m : a.M - 1;
        float px, py, pz, dx, dy, dz;
        if (a.mode == 0) {
            const long long ray = mm / a.S;
            const f32x4 o = *reinterpret_cast<const f32x4*>(a.in_a + ray * 4);
            const f32x4 d = *reinterpret_cast<const f32x4*>(a.in_b + ray * 4);
            const float zz = a.z[mm];
            px = __fadd_rn(o[0], __fmul_rn(d[0], zz));   // o + d*z, mul then add (UtilsCV.py:598)
            py = __fadd_rn(o[1], __fmul_rn(d[1], zz));
            pz = __fadd_rn(o[2], __fmul_rn(d[2], zz));
            dx = d[0]; dy = d[1]; dz = d[2];
        } else {
            px = a.in_a[mm * 3 + 0]; py = a.in_a[mm * 3 + 1]; pz = a.in_a[mm * 3 + 2];
            if constexpr (XYZ) { dx = dy = dz = 0.f; }        // no view directions in this network (in_b may be null)
            else { dx = a.in_b[mm * 3 + 0]; dy = a.in_b[mm * 3 + 1]; dz = a.in_b[mm * 3 + 2]; }
        }
        const float kPi = 3.1415927410125732f;   // fp32(pi): theta = (2^k * pi_f32) * x
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = c == 0 ? px : c == 1 ? py : pz;
#pragma unroll
            for (int k = 0; k < kLx; ++k) xpe[c * kLx + k] = sin_shifted(v * (kPi * (float)(1 << k)), h);
        }
        xpe[15] = h ? py : px;
        xpe[16] = h ? 0.f : pz;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = c == 0 ? dx : c == 1 ? dy : dz;
#pragma unroll
            for (int k = 0; k < kLd; ++k) xdir[c * kLd + k] = sin_shifted(v * (kPi * (float)(1 << k)), h);
        }

        // ---------------- the 9 MFMA layers ----------------
        STAMP(t1); acc_t[0] += t1 - t0;
        float sig = 0.f;
        layer_body<BODY_PE, false>(p, lane16, cb_body, (kConstBias + 0 * 256) * 4, a.alpha, accs, xin, xnext, xpe, xdir, xc, sig);
        STAMP(t0); acc_t[1] += t0 - t1;
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            if (l == 4) {
                layer_body<BODY_SKIP, true>(p, lane16, cb_body, (kConstBias + 4 * 256) * 4, a.alpha, accs, xin, xnext, xpe, xdir, xc, sig);
                STAMP(t1); acc_t[3] += t1 - t0; t0 = t1;
            } else {
                layer_body<BODY_HID, true>(p, lane16, cb_body, (kConstBias + l * 256) * 4, a.alpha, accs, xin, xnext, xpe, xdir, xc, sig);
                STAMP(t1); acc_t[2] += t1 - t0; t0 = t1;
            }
        }
        if constexpr (XYZ) {
            layer_body<BODY_HSIG, true>(p, lane16, cb_body, kFXConstBias8 * 4, a.alpha, accs, xin, xnext, xpe, xdir, xc, sig);
            layer_body<BODY_LASTX, true>(p, lane16, cb_body, kFXConstBias9 * 4, a.alpha, accs, xin, xnext, xpe, xdir, xc, sig);
        } else {
            layer_body<BODY_LAST, true>(p, lane16, cb_body, kConstBias8 * 4, a.alpha, accs, xin, xnext, xpe, xdir, xc, sig);
        }
        STAMP(t1); acc_t[4] += t1 - t0;

        // ---------------- heads on the VALU ----------------
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                constexpr int kW = XYZ ? kFXConstWrgb : kConstWrgb;
                const f32x4 w0 = lds_read4(cb_head + (kW + 0 * 128 + t * 32 + g * 8) * 4);
                const f32x4 w1 = lds_read4(cb_head + (kW + 1 * 128 + t * 32 + g * 8) * 4);
                const f32x4 w2 = lds_read4(cb_head + (kW + 2 * 128 + t * 32 + g * 8) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = xc[t * 16 + g * 4 + e];
                    o0 = fmaf(w0[e], x, o0);
                    o1 = fmaf(w1[e], x, o1);
                    o2 = fmaf(w2[e], x, o2);
                }
            }
        }
        if constexpr (XYZ) {
            o3 = sig;                         // accumulated beside the extra layer's MFMAs (BODY_HSIG)
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 w = lds_read4(cb_head + (kConstWsigH + t * 32 + g * 8) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o3 = fmaf(w[e], xin[t * 16 + g * 4 + e], o3);
                }
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const f32x4 w = lds_read4(cb_head + (kConstWsigD + g * 8) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o3 = fmaf(w[e], xdir[g * 4 + e], o3);
            }
        }
        o0 += __shfl_xor(o0, 32);
        o1 += __shfl_xor(o1, 32);
        o2 += __shfl_xor(o2, 32);
        o3 += __shfl_xor(o3, 32);
        const f32x4 bh = lds_read4(kLdsConst + (XYZ ? kFXConstBHead : kConstBHead) * 4);
        if (valid && h == 0) {
            f32x4 out;
            out[0] = o0 + bh[0]; out[1] = o1 + bh[1]; out[2] = o2 + bh[2]; out[3] = o3 + bh[3];
            *reinterpret_cast<f32x4*>(a.raw + m * 4) = out;
            // overflow / NaN watch (the f16x3 mode saturates above 65504): count, never hide
            const float chk = out[0] + out[1] + out[2] + out[3];
            if (a.nonfinite && !(fabsf(chk) <= 3.0e38f)) atomicAdd(a.nonfinite, 1ull);
        }
        STAMP(t0); acc_t[5] += t0 - t1; acc_t[6] += 1;
    }
#ifdef NERF_STAMPS
    if (blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 8; ++i) g_stamps[i] = acc_t[i];
#endif
    // drain the run-ahead weight prefetch before the wave (and its LDS allocation) goes away
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(256, 1) void mlp_fp32_kernel(const MlpArgs a) { mlp_fp32_body<false>(a); }
__global__ __launch_bounds__(256, 1) void mlp_fp32_xyz_kernel(const MlpArgs a) { mlp_fp32_body<true>(a); }

void launch_mlp_fp32(const MlpArgs& a, int num_cus, hipStream_t stream, bool xyz_only) {
    if (a.M <= 0) return;
    const long long ntiles = (a.M + 127) / 128;
    const int grid = (int)(ntiles < (long long)num_cus ? ntiles : (long long)num_cus);
    if (xyz_only) hipLaunchKernelGGL(mlp_fp32_xyz_kernel, dim3(grid), dim3(256), kLdsTotal, stream, a);
    else hipLaunchKernelGGL(mlp_fp32_kernel, dim3(grid), dim3(256), kLdsTotal, stream, a);
}

#ifdef NERF_STAMPS
extern "C" void nerf_debug_read_stamps(unsigned long long* out) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16);
}
#endif

void mlp_fp32_set_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fp32_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fp32_xyz_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, kLdsTotal);
}

// ------------------------------------------------------------------------------------------------
// Host-side packing: blob (Keras get_weights() order, kernel (in,out) row-major) -> stream + const.
// ------------------------------------------------------------------------------------------------
namespace {
// input row of the layer kernel for PE k-step s (0..19) and lane half h; -1 = zero pad
int pe_row(int s, int h) {
    if (s < 15) { const int c = s / 5, k = s % 5; return c * 11 + 1 + 2 * k + h; }
    if (s == 15) return h ? 11 : 0;
    if (s == 16) return h ? -1 : 22;
    return -1;
}
int hid_row(int s, int h) { const int t = s >> 4, r = s & 15; return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h; }
// dir k-step s (0..11) = component c (x,y,z) x octave k -> row of the dir block of the kernel, or -1.
// n_angles == 2: view dirs are (x,y,z) -> 24 rows.  n_angles == 1: the reference feeds only (x,z)
// (src/UtilsCV.py:134-135) -> 16 rows; the kernel still encodes y, its weights are packed as zero.
int dir_row(int s, int h, int n_angles) {
    const int c = s / 4, k = s % 4;
    if (n_angles == 2) return c * 8 + 2 * k + h;
    if (c == 1) return -1;
    return (c == 0 ? 0 : 1) * 8 + 2 * k + h;
}

struct Layer { const float* k; const float* b; int in, out; };
}  // namespace

void pack_weights_fp32(const float* blob, int n_angles, float* stream_out, float* const_out) {
    const int kd = 256 + 8 * (n_angles + 1);   // width of [hidden, dir_enc]: 280 (n_angles 2) or 272 (1)
    const bool xyz_only = n_angles == 0;
    const int shapes_dir[12][2] = {{33, 256}, {256, 256}, {256, 256}, {256, 256}, {289, 256}, {256, 256},
                                   {256, 256}, {256, 256}, {kd, 128}, {128, 3}, {kd, 1}, {0, 0}};
    // get_network_only_xyz (src/NeRF.py:248-288): ..., 8: 256 -> 256, 9: 256 -> 128, 10: 128 -> 3, 11: 256 -> 1
    const int shapes_xyz[12][2] = {{33, 256}, {256, 256}, {256, 256}, {256, 256}, {289, 256}, {256, 256},
                                   {256, 256}, {256, 256}, {256, 256}, {256, 128}, {128, 3}, {256, 1}};
    const int (*shapes)[2] = xyz_only ? shapes_xyz : shapes_dir;
    Layer L[12];
    size_t off = 0;
    for (int i = 0; i < (xyz_only ? 12 : 11); ++i) {
        L[i].in = shapes[i][0]; L[i].out = shapes[i][1];
        L[i].k = blob + off; off += (size_t)L[i].in * L[i].out;
        L[i].b = blob + off; off += L[i].out;
    }
    memset(stream_out, 0, xyz_only ? kStreamBytesXyzF32 : kStreamBytes);
    memset(const_out, 0, kConstBytes);
    size_t chunk = 0;
    auto emit_body = [&](int layer, int body) {
        const int NU = (body == BODY_LAST || body == BODY_LASTX) ? 4 : 8;
        const int QPU = body == BODY_PE ? kQpuPE : (body == BODY_HID || body == BODY_LASTX) ? kQpuHid
                        : body == BODY_SKIP ? kQpuSkip : kQpuLast;
        float* base = stream_out + chunk * (kChunkBytes / 4);
        for (int u = 0; u < NU; ++u)
            for (int q = 0; q < QPU; ++q) {
                // tiles are consumed in pairs: quad order is (pair, q, tile-of-pair)
                float* quad = base + (size_t)(((u >> 1) * QPU + q) * 2 + (u & 1)) * (kQuadBytes / 4);
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int i = lane & 31, h = lane >> 5;
                        int row;
                        if (body == BODY_PE) row = pe_row(4 * q + e, h);
                        else if (body == BODY_HID || body == BODY_LASTX) row = hid_row(4 * q + e, h);
                        else if (body == BODY_SKIP) row = q < kQpuPE ? pe_row(4 * q + e, h) : kXyzDim + hid_row(4 * (q - kQpuPE) + e, h);
                        else if (q < kQpuHid) row = hid_row(4 * q + e, h);
                        else { const int r = dir_row(4 * (q - kQpuHid) + e, h, n_angles); row = r < 0 ? -1 : kHidden + r; }
                        quad[lane * 4 + e] = row < 0 ? 0.f : L[layer].k[(size_t)row * L[layer].out + 32 * u + i];
                    }
            }
        chunk += (NU * QPU + kChunkQuads - 1) / kChunkQuads;
    };
    emit_body(0, BODY_PE);
    for (int l = 1; l <= 3; ++l) emit_body(l, BODY_HID);
    emit_body(4, BODY_SKIP);
    for (int l = 5; l <= 7; ++l) emit_body(l, BODY_HID);
    for (int l = 0; l < 8; ++l)
        for (int f = 0; f < 256; ++f) const_out[kConstBias + l * 256 + f] = L[l].b[f];
    if (xyz_only) {
        emit_body(8, BODY_HID);          // the kernel runs it as BODY_HSIG: same operand order
        emit_body(9, BODY_LASTX);
        for (int f = 0; f < 256; ++f) const_out[kFXConstBias8 + f] = L[8].b[f];
        for (int f = 0; f < 128; ++f) const_out[kFXConstBias9 + f] = L[9].b[f];
        for (int c = 0; c < 3; ++c)
            for (int f = 0; f < 128; ++f) const_out[kFXConstWrgb + c * 128 + f] = L[10].k[f * 3 + c];
        for (int c = 0; c < 3; ++c) const_out[kFXConstBHead + c] = L[10].b[c];
        const_out[kFXConstBHead + 3] = L[11].b[0];
        for (int f = 0; f < 256; ++f) const_out[kFXConstWsigH + f] = L[11].k[f];
        return;
    }
    emit_body(8, BODY_LAST);
    // constants
    for (int f = 0; f < 128; ++f) const_out[kConstBias8 + f] = L[8].b[f];
    for (int c = 0; c < 3; ++c)
        for (int f = 0; f < 128; ++f) const_out[kConstWrgb + c * 128 + f] = L[9].k[f * 3 + c];
    for (int c = 0; c < 3; ++c) const_out[kConstBHead + c] = L[9].b[c];
    const_out[kConstBHead + 3] = L[10].b[0];
    for (int f = 0; f < 256; ++f) const_out[kConstWsigH + f] = L[10].k[f];
    for (int g = 0; g < 3; ++g)
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 4; ++e)
            {
                    const int r = dir_row(4 * g + e, h, n_angles);
                    const_out[kConstWsigD + (g * 2 + h) * 4 + e] = r < 0 ? 0.f : L[10].k[kHidden + r];
                }
}

}  // namespace nerf

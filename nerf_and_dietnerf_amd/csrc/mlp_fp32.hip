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
//   * weights (2.08 MB / net) stream L2 -> LDS through a 4 x 16 KiB ring by LDS-DMA
//     (global_load_lds_dwordx4), one counted vmcnt + one s_barrier per 64 MFMAs, shared by the 4
//     waves; A operands are read back with one ds_read_b128 per 4 MFMAs.
//   * heads (128->3, 280->1) on the VALU; positional encoding in-register with a Cody-Waite sincos.
#include "nerf_kernels.h"
#include "nerf_device.h"

#include <math.h>
#include <string.h>

#include <type_traits>

namespace nerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

__device__ __forceinline__ f32x4 lds_read4(uint32_t byte_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    return *reinterpret_cast<const f32x4*>(smem + byte_off);
}

struct Pipe {
    int ck;        // chunk being consumed (monotonic; ring position = ck & 3)
    int src_next;  // next chunk index of the cyclic weight stream to DMA (0..kStreamChunks-1)
};

// Issue this wave's 4 KiB share of one 16 KiB chunk: 4 x global_load_lds_dwordx4 (1 KiB each).
__device__ __forceinline__ void dma_issue(const char* wsrc_lane, int src_chunk, int ring_pos, int wave) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const char* g = wsrc_lane + (size_t)src_chunk * kChunkBytes;
    char* l = smem + kLdsRing + ring_pos * kChunkBytes + wave * (4 * kQuadBytes);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __builtin_amdgcn_global_load_lds((const GLB_AS void*)(g + j * kQuadBytes),
                                         (LDS_AS void*)(l + j * kQuadBytes), 16, 0, 0);
    }
}

// Mid-chunk synchronisation point of chunk p.ck:
//   vmcnt(4): this wave's share of chunk ck+1 has landed (only chunk ck+2's 4 DMAs may be pending);
//   lgkmcnt(8): every ds_read of chunk ck-1 has returned (at most this chunk's first 8 pending);
//   barrier:  => all waves' shares of ck+1 are visible, and ring slot (ck-1)&3 is free for reuse.
// Then issue chunk ck+3 into that free slot.
__device__ __forceinline__ void pipe_sync(Pipe& p, const char* wsrc_lane, int wave) {
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    dma_issue(wsrc_lane, p.src_next, (p.ck + 3) & 3, wave);
    p.src_next = (p.src_next + 1 == kStreamChunks) ? 0 : p.src_next + 1;
}

enum { BODY_PE = 0, BODY_HID = 1, BODY_SKIP = 2, BODY_LAST = 3 };

// One dense layer, u-outer: for each 32-wide output tile run the whole K chain into one accumulator.
//   BODY_PE   : B = xpe                      -> xin   (layer 0)
//   BODY_HID  : B = xin                      -> xin   (layers 1-3, 5-7; via xnext + staged copy-back)
//   BODY_SKIP : B = [xpe, xin]               -> xin   (layer 4)
//   BODY_LAST : B = [xin, xdir], 4 tiles     -> xc    (layer 8)
// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int BODY>
__device__ __forceinline__ void layer_body(Pipe& p, const char* wsrc_lane, int wave, uint32_t lane16,
                                           uint32_t cb_h, int bias_off_bytes, float alpha,
                                           float (&xin)[128], float (&xnext)[112], const float (&xpe)[17],
                                           const float (&xdir)[12], float (&xc)[64]) {
    constexpr int NU = BODY == BODY_LAST ? 4 : 8;
    constexpr int QPU = BODY == BODY_PE ? kQpuPE : BODY == BODY_HID ? kQpuHid
                        : BODY == BODY_SKIP ? kQpuSkip : kQpuLast;
    constexpr int NQ = NU * QPU;
    f32x16 acc0, acc1;
    uint32_t rd = lane16 + (uint32_t)(p.ck & 3) * kChunkBytes;
    f32x4 a_nx = lds_read4(rd);   // A operands are fetched one quad ahead of the MFMAs that use them

    static_for<0, NU>([&](auto uc) {
        constexpr int u = decltype(uc)::value;
        f32x16& acc = (u & 1) ? acc1 : acc0;
        f32x16& accp = (u & 1) ? acc0 : acc1;   // previous tile's accumulator
        // accumulator starts as the bias of the 32 features of this tile (C-in of the first MFMA)
        static_for<0, 4>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const f32x4 b = lds_read4(cb_h + bias_off_bytes + (u * 32 + g * 8) * 4);
            acc[4 * g + 0] = b[0]; acc[4 * g + 1] = b[1]; acc[4 * g + 2] = b[2]; acc[4 * g + 3] = b[3];
        });
        static_for<0, QPU>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int Q = u * QPU + q;
            // mid-chunk sync of the chunk quad Q lives in (p.ck still names that chunk here)
            if constexpr (Q % kChunkQuads == 8) pipe_sync(p, wsrc_lane, wave);
            const f32x4 a4 = a_nx;
            if constexpr (Q + 1 < NQ) {
                if constexpr ((Q + 1) % kChunkQuads == 0) {
                    p.ck += 1;
                    rd = lane16 + (uint32_t)(p.ck & 3) * kChunkBytes;
                }
                a_nx = lds_read4(rd + ((Q + 1) % kChunkQuads) * kQuadBytes);
            }
            static_for<0, 4>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                if constexpr (BODY == BODY_PE) {
                    constexpr int s = 4 * q + e;
                    if constexpr (s < 17) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], xpe[s], acc, 0, 0, 0);
                } else if constexpr (BODY == BODY_HID) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], xin[4 * q + e], acc, 0, 0, 0);
                } else if constexpr (BODY == BODY_SKIP) {
                    if constexpr (q < kQpuPE) {
                        constexpr int s = 4 * q + e;
                        if constexpr (s < 17) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], xpe[s], acc, 0, 0, 0);
                    } else {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], xin[4 * (q - kQpuPE) + e], acc, 0, 0, 0);
                    }
                } else {
                    if constexpr (q < kQpuHid) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], xin[4 * q + e], acc, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], xdir[4 * (q - kQpuHid) + e], acc, 0, 0, 0);
                }
            });
            // deferred epilogue of the previous tile, placed in the shadow of this tile's MFMAs
            if constexpr (q == 2 && u > 0) {
                static_for<0, 16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const float v = accp[r];
                    const float y = fmaxf(v, alpha * v);
                    if constexpr (BODY == BODY_PE) xin[(u - 1) * 16 + r] = y;
                    else if constexpr (BODY == BODY_LAST) xc[(u - 1) * 16 + r] = y;
                    else xnext[(u - 1) * 16 + r] = y;
                });
            }
            // last chain of an in-place layer: tile t of xin is dead once its 4 quads are consumed
            if constexpr ((BODY == BODY_HID || BODY == BODY_SKIP) && u == NU - 1) {
                constexpr int qh = BODY == BODY_SKIP ? q - kQpuPE : q;
                if constexpr (qh >= 0 && (qh & 3) == 3 && (qh >> 2) < 7) {
                    constexpr int t = qh >> 2;
                    static_for<0, 16>([&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        xin[t * 16 + r] = xnext[t * 16 + r];
                    });
                }
            }
        });
    });
    {   // epilogue of the last tile (its inputs are dead: write in place)
        f32x16& accl = ((NU - 1) & 1) ? acc1 : acc0;
        static_for<0, 16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const float v = accl[r];
            const float y = fmaxf(v, alpha * v);
            if constexpr (BODY == BODY_LAST) xc[(NU - 1) * 16 + r] = y;
            else xin[(NU - 1) * 16 + r] = y;
        });
    }
    if constexpr (NQ % kChunkQuads != 0 && NQ % kChunkQuads <= 8) pipe_sync(p, wsrc_lane, wave);
    p.ck += 1;  // every body starts on a chunk boundary
}

__global__ __launch_bounds__(256, 1) void mlp_fp32_kernel(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const uint32_t lane16 = kLdsRing + lane * 16;
    const uint32_t cb_h = kLdsConst + h * 16;

    const long long ntiles = (a.M + 127) / 128;
    if ((long long)blockIdx.x >= ntiles) return;   // uniform per workgroup

    // constant region -> LDS (once per workgroup)
    for (int i = tid; i < kConstFloats / 4; i += 256)
        reinterpret_cast<f32x4*>(smem + kLdsConst)[i] = reinterpret_cast<const f32x4*>(a.wconst)[i];

    const char* wsrc_lane = reinterpret_cast<const char*>(a.wstream) + wave * (4 * kQuadBytes) + lane * 16;
    Pipe p;
    p.ck = 0;
    p.src_next = 0;
    __syncthreads();   // const region visible; no DMA in flight yet
    // pipeline prologue: chunks 0,1,2 in flight, chunk 0 landed for everyone
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dma_issue(wsrc_lane, p.src_next, c, wave);
        p.src_next += 1;
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    float xin[128], xnext[112], xpe[17], xdir[12], xc[64];

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
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
            dx = a.in_b[mm * 3 + 0]; dy = a.in_b[mm * 3 + 1]; dz = a.in_b[mm * 3 + 2];
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
        layer_body<BODY_PE>(p, wsrc_lane, wave, lane16, cb_h, (kConstBias + 0 * 256) * 4, a.alpha, xin, xnext, xpe, xdir, xc);
#pragma unroll 1
        for (int l = 1; l <= 7; ++l) {
            if (l == 4)
                layer_body<BODY_SKIP>(p, wsrc_lane, wave, lane16, cb_h, (kConstBias + 4 * 256) * 4, a.alpha, xin, xnext, xpe, xdir, xc);
            else
                layer_body<BODY_HID>(p, wsrc_lane, wave, lane16, cb_h, (kConstBias + l * 256) * 4, a.alpha, xin, xnext, xpe, xdir, xc);
        }
        layer_body<BODY_LAST>(p, wsrc_lane, wave, lane16, cb_h, kConstBias8 * 4, a.alpha, xin, xnext, xpe, xdir, xc);

        // ---------------- heads on the VALU ----------------
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w0 = lds_read4(cb_h + (kConstWrgb + 0 * 128 + t * 32 + g * 8) * 4);
                const f32x4 w1 = lds_read4(cb_h + (kConstWrgb + 1 * 128 + t * 32 + g * 8) * 4);
                const f32x4 w2 = lds_read4(cb_h + (kConstWrgb + 2 * 128 + t * 32 + g * 8) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = xc[t * 16 + g * 4 + e];
                    o0 = fmaf(w0[e], x, o0);
                    o1 = fmaf(w1[e], x, o1);
                    o2 = fmaf(w2[e], x, o2);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w = lds_read4(cb_h + (kConstWsigH + t * 32 + g * 8) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o3 = fmaf(w[e], xin[t * 16 + g * 4 + e], o3);
            }
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const f32x4 w = lds_read4(cb_h + (kConstWsigD + g * 8) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o3 = fmaf(w[e], xdir[g * 4 + e], o3);
        }
        o0 += __shfl_xor(o0, 32);
        o1 += __shfl_xor(o1, 32);
        o2 += __shfl_xor(o2, 32);
        o3 += __shfl_xor(o3, 32);
        const f32x4 bh = lds_read4(kLdsConst + kConstBHead * 4);
        if (valid && h == 0) {
            f32x4 out;
            out[0] = o0 + bh[0]; out[1] = o1 + bh[1]; out[2] = o2 + bh[2]; out[3] = o3 + bh[3];
            *reinterpret_cast<f32x4*>(a.raw + m * 4) = out;
        }
    }
    // drain the run-ahead weight prefetch before the wave (and its LDS allocation) goes away
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

void launch_mlp_fp32(const MlpArgs& a, int num_cus, hipStream_t stream) {
    if (a.M <= 0) return;
    const long long ntiles = (a.M + 127) / 128;
    const int grid = (int)(ntiles < (long long)num_cus ? ntiles : (long long)num_cus);
    hipLaunchKernelGGL(mlp_fp32_kernel, dim3(grid), dim3(256), kLdsTotal, stream, a);
}

void mlp_fp32_set_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fp32_kernel),
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
int dir_row(int s, int h) { const int c = s / 4, k = s % 4; return c * 8 + 2 * k + h; }

struct Layer { const float* k; const float* b; int in, out; };
}  // namespace

void pack_weights_fp32(const float* blob, float* stream_out, float* const_out) {
    static const int shapes[11][2] = {{33, 256}, {256, 256}, {256, 256}, {256, 256}, {289, 256}, {256, 256},
                                      {256, 256}, {256, 256}, {280, 128}, {128, 3}, {280, 1}};
    Layer L[11];
    size_t off = 0;
    for (int i = 0; i < 11; ++i) {
        L[i].in = shapes[i][0]; L[i].out = shapes[i][1];
        L[i].k = blob + off; off += (size_t)L[i].in * L[i].out;
        L[i].b = blob + off; off += L[i].out;
    }
    memset(stream_out, 0, kStreamBytes);
    memset(const_out, 0, kConstBytes);
    size_t chunk = 0;
    auto emit_body = [&](int layer, int body) {
        const int NU = body == BODY_LAST ? 4 : 8;
        const int QPU = body == BODY_PE ? kQpuPE : body == BODY_HID ? kQpuHid : body == BODY_SKIP ? kQpuSkip : kQpuLast;
        float* base = stream_out + chunk * (kChunkBytes / 4);
        for (int u = 0; u < NU; ++u)
            for (int q = 0; q < QPU; ++q) {
                float* quad = base + (size_t)(u * QPU + q) * (kQuadBytes / 4);
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int i = lane & 31, h = lane >> 5;
                        int row;
                        if (body == BODY_PE) row = pe_row(4 * q + e, h);
                        else if (body == BODY_HID) row = hid_row(4 * q + e, h);
                        else if (body == BODY_SKIP) row = q < kQpuPE ? pe_row(4 * q + e, h) : kXyzDim + hid_row(4 * (q - kQpuPE) + e, h);
                        else row = q < kQpuHid ? hid_row(4 * q + e, h) : kHidden + dir_row(4 * (q - kQpuHid) + e, h);
                        quad[lane * 4 + e] = row < 0 ? 0.f : L[layer].k[(size_t)row * L[layer].out + 32 * u + i];
                    }
            }
        chunk += (NU * QPU + kChunkQuads - 1) / kChunkQuads;
    };
    emit_body(0, BODY_PE);
    for (int l = 1; l <= 3; ++l) emit_body(l, BODY_HID);
    emit_body(4, BODY_SKIP);
    for (int l = 5; l <= 7; ++l) emit_body(l, BODY_HID);
    emit_body(8, BODY_LAST);
    // constants
    for (int l = 0; l < 8; ++l)
        for (int f = 0; f < 256; ++f) const_out[kConstBias + l * 256 + f] = L[l].b[f];
    for (int f = 0; f < 128; ++f) const_out[kConstBias8 + f] = L[8].b[f];
    for (int c = 0; c < 3; ++c)
        for (int f = 0; f < 128; ++f) const_out[kConstWrgb + c * 128 + f] = L[9].k[f * 3 + c];
    for (int c = 0; c < 3; ++c) const_out[kConstBHead + c] = L[9].b[c];
    const_out[kConstBHead + 3] = L[10].b[0];
    for (int f = 0; f < 256; ++f) const_out[kConstWsigH + f] = L[10].k[f];
    for (int g = 0; g < 3; ++g)
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 4; ++e)
                const_out[kConstWsigD + (g * 2 + h) * 4 + e] = L[10].k[kHidden + dir_row(4 * g + e, h)];
}

}  // namespace nerf

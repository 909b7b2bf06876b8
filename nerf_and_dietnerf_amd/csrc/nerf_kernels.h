// nerf_kernels.h -- internal declarations shared by the HIP translation units of libnerf_mi355.so.
// gfx950 (MI355X) only.  Nothing here is part of the C ABI (see include/nerf_mi355.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nerf {

// ---- fixed network geometry of the fused kernel (all BASELINE configs) ----
// n_angles_for_model 2 and 1 share the kernels: with 1 the y component of the view direction is still
// encoded on device but its weight rows are packed as zeros (host side, pack_weights_*).
constexpr int kLx = 5;           // n_pos_enc_dim_xyz
constexpr int kLd = 4;           // n_pos_enc_view_dir
constexpr int kHidden = 256;
constexpr int kLast = 128;
constexpr int kXyzDim = 3 + 6 * kLx;   // 33
constexpr int kDirDim = 6 * kLd;       // 24

// ---- weight stream geometry (see DESIGN.md "weight stream") ----
// A "quad" is the A operand of 4 consecutive MFMA k-steps for one 32-wide output tile:
// 64 lanes x 4 floats = 1 KiB, lane-linear, read with one ds_read_b128 per lane.
constexpr int kQuadBytes = 1024;
constexpr int kChunkQuads = 16;
constexpr int kChunkBytes = kQuadBytes * kChunkQuads;   // 16 KiB = one LDS ring slot
constexpr int kRingChunks = 8;   // power of two
constexpr int kRingBytes = kChunkBytes * kRingChunks;   // 128 KiB

// quads per output tile for each layer body
constexpr int kQpuPE = 5;                 // 17 k-steps (15 sin/cos pairs + raw xyz) padded to 20
constexpr int kQpuHid = 32;               // 128 k-steps
constexpr int kQpuSkip = kQpuPE + kQpuHid;  // layer 4: [xyz_enc(33), hidden(256)]
constexpr int kQpuLast = kQpuHid + 3;     // layer 8: [hidden(256), dir_enc(24)] -> 128
constexpr int kChunksPE = (8 * kQpuPE + 15) / 16;      // 3
constexpr int kChunksHid = (8 * kQpuHid) / 16;         // 16
constexpr int kChunksSkip = (8 * kQpuSkip + 15) / 16;  // 19
constexpr int kChunksLast = (4 * kQpuLast + 15) / 16;  // 9
constexpr int kStreamChunks = kChunksPE + 3 * kChunksHid + kChunksSkip + 3 * kChunksHid + kChunksLast;  // 127
constexpr size_t kStreamBytes = size_t(kStreamChunks) * kChunkBytes;
constexpr size_t kStreamBytesXyzF32 = size_t(142) * kChunkBytes;   // the xyz-only network's fp32 stream (12 Dense layers, mlp_fp32.hip)
constexpr size_t kStreamBytesF16 = size_t(66) * 32 * kQuadBytes;   // f16x3 stream: 66 chunks of 32 KiB (mlp_f16x3.hip)
constexpr size_t kStreamBytesF16Hi = size_t(33) * 32 * kQuadBytes; // single-pass fp16 stream: hi fragments only
constexpr size_t kStreamBytesF16Xyz = size_t(73) * 32 * kQuadBytes;    // the xyz-only network's streams (12 Dense layers)
constexpr size_t kStreamBytesF16HiXyz = size_t(37) * 32 * kQuadBytes;

// ---- constant region (biases + head weights), floats ----
constexpr int kConstBias = 0;                       // 8 x 256 hidden-layer biases (layers 0..7)
constexpr int kConstBias8 = 2048;                   // 128
constexpr int kConstWrgb = 2176;                    // [3][128]
constexpr int kConstBHead = 2560;                   // b_r, b_g, b_b, b_sigma
constexpr int kConstWsigH = 2564;                   // [256]  sigma head, hidden part
constexpr int kConstWsigD = 2820;                   // [3][2][4] sigma head, dir part (g, half, e)
constexpr int kConstFloats = 3136;                  // the LDS carve of the constants (largest user: the xyz-only trainer's backward: 3120 + 16 gmax slots)
constexpr int kConstBytes = kConstFloats * 4;

// LDS carve of the MLP kernel
constexpr int kLdsRing = 0;
constexpr int kLdsConst = kRingBytes;
constexpr int kLdsTotal = kRingBytes + kConstBytes;   // 142,464 B of the 160 KiB

struct MlpArgs {
    const float* wstream;   // packed A-operand stream of one network (kStreamBytes)
    const float* wconst;    // constant region (kConstFloats)
    const float* in_a;      // mode 0: rays_orig (N,4)   | mode 1: xyz (M,3)
    const float* in_b;      // mode 0: rays_dirs (N,4)   | mode 1: view_dirs (M,3)
    const float* z;         // mode 0: (N,S)             | mode 1: unused
    float* raw;             // (M,4) raw [r,g,b,sigma]
    unsigned long long* nonfinite;   // device counter: rows whose raw output is not finite (may be null)
    long long M;            // number of samples (rows)
    int S;                  // samples per ray (mode 0)
    int mode;
    float alpha;            // LeakyReLU slope
    // training forward (mlp_f16x3 "stash" kernels only): where the activation of layer l = 0..8 (0..9 for the xyz-only
    // network: ..., 8 = its extra 256-wide layer, 9 = the 128-wide one) is written, fragment-major with st_ld[l] elements
    // per row (rows padded to a multiple of 128); unused (null) when rendering
    float* st_ptr[10];
    int st_ld[10];
    // ... and where the LeakyReLU' mask bits of layer l's output go: one uint4 per (row, lane half), see mlp_f16x3.hip
    uint32_t* mask_ptr[10];
    int diag_wrap;          // diagnostic only (NERF_DIAG_STASH_WRAP=1): every workgroup's stash rows land in the first 8192 rows (cache-resident)
};

// mlp_fp32.hip
void launch_mlp_fp32(const MlpArgs& a, int num_cus, hipStream_t stream, bool xyz_only = false);
void mlp_fp32_set_attributes();
// host-side packing of one network's blob (11 x (kernel(in,out), bias)) into stream + const
void pack_weights_fp32(const float* blob, int n_angles, float* stream_out /*kStreamBytes/4; n_angles 0: kStreamBytesXyzF32/4*/, float* const_out /*kConstFloats*/);

// mlp_f16x3.hip
// single_pass: hi*hi only; xyz_only: the 12-layer network without view directions (its own streams / constants)
void launch_mlp_f16x3(const MlpArgs& a, int num_cus, hipStream_t stream, bool single_pass = false, bool xyz_only = false);
// forward that also writes a.st_ptr / a.mask_ptr (training); single_pass: the mixed_float16-class arithmetic
void launch_mlp_f16x3_stash(const MlpArgs& a, int num_cus, hipStream_t stream, bool single_pass = false, bool xyz_only = false);
// device-side re-pack of the 3-pass (or hi-only) stream + constants from a blob (tables from build_f16x3_gather, host)
size_t f16_stream_bytes(int n_angles, bool hi_only);     // kStreamBytesF16[Hi][Xyz]
void build_f16x3_gather(int n_angles, bool hi_only, int32_t* stream_idx /* f16_stream_bytes / 2 */,
                        int32_t* const_idx /*kConstFloats*/);
void launch_repack_f16x3(const float* blob, const int32_t* stream_idx, void* stream, const int32_t* const_idx, float* cst,
                         size_t stream_bytes, hipStream_t s);
void mlp_f16x3_set_attributes();
// stream_out: kStreamBytesF16 / kStreamBytesF16Hi bytes (kStreamBytesF16Xyz / kStreamBytesF16HiXyz when n_angles == 0)
void pack_weights_f16x3(const float* blob, int n_angles, void* stream_out, float* const_out /*kConstFloats*/);
void pack_weights_f16(const float* blob, int n_angles, void* stream_out, float* const_out /*kConstFloats*/);

// mlp_f16_2t.hip -- single-pass fp16 render kernel with two 32-sample tiles per wave (same hi-only stream / constants)
void launch_mlp_f16_2t(const MlpArgs& a, int num_cus, hipStream_t stream);
void mlp_f16_2t_set_attributes();

// mlp_bwd_f16x3.hip -- the trainer's fused data-gradient chain (the stash forward's counterpart)
constexpr size_t kBwdStreamBytes = size_t(73) * 32 * kQuadBytes;   // transposed-weight stream incl. the encoding tiles (73 chunks: the xyz-only network's)
constexpr int kBwdXyzLd = 64;                                      // floats per row of an encoding-gradient buffer
// "pair16" gradient buffers of the fused float32-policy trainer (default on; -DNERF_PAIR16=0 builds the fp32 buffers of
// rounds 2-3 back for A/B runs): the backward chain stores every pre-activation gradient as the fp16 (hi, lo) pair it
// packs as the next MFMA operand anyway, in the value's fp32 slot -- {hi01, hi23, lo01, lo23} per four consecutive
// features of a row -- with the chain's per-row power-of-two scale still on it (its inverse is stored per row,
// MlpBwdArgs::rs_ptr), and the weight-gradient GEMM (gemm_atb_p, train_kernels.hip) stages that operand with byte permutes
// and one packed multiply instead of scaling and splitting fp32 rows on the fly.  The chain no longer forms the
// true-scale value at all (a select and a multiply per value less).
#ifndef NERF_PAIR16
#define NERF_PAIR16 1
#endif
constexpr bool kPair16 = NERF_PAIR16 != 0;
struct MlpBwdArgs {
    const void* wstream;     // backward operand stream of one network (build_bwd_gather / launch_repack_bwd)
    const float* wconst;     // the forward kernel's constant block (rgb head weights are read from it)
    const float* graw;       // (Mp, 4) gradient w.r.t. the raw network output [r, g, b, sigma]; padding rows zero
    const uint32_t* mask_ptr[10];  // LeakyReLU' bit records of layers 0..8 (0..9: xyz-only network), written by the stash forward
    float* d_ptr[10];        // d_ptr[l], l = 0..7: (Mp, 256) gradient w.r.t. layer l's pre-activation; d_ptr[8]: G9 (Mp, 128)
                             // xyz-only network: d_ptr[8] = (Mp, 256) of its extra layer, d_ptr[9] = (Mp, 128) of the last one
    float* dx_ptr[2];        // dx variant: (Mp, 64) gradient w.r.t. the xyz encoding through layer 4 / through layer 0
    uint16_t* rs_ptr[10];    // kPair16 (3-pass kernels): per buffer d_ptr[l] and row the power of two r with true D = stored D' * r,
                             // as the upper half of r's fp32 bits
    float* dsig;             // optional (Mp): column 3 of graw as a contiguous vector -- the sigma head's weight gradient rides in the
                             // weight-gradient GEMM of layer 8, which stages C8 anyway (GemmAtb::sig_g)
    unsigned* gmax;          // 9 x 64 slots: bits of max|.| of G9 (slot group 0) and of D_(8-k) (slot group k); xyz-only
                             // network: 10 groups, group 0 = d_ptr[9], group k = d_ptr[9 - k]
    long long Mp;            // rows, multiple of 128
    float alpha;
    int ld, ld9;             // row pitch (floats) of d_ptr[0..7] / d_ptr[8]
};
// single_pass: hi*hi products only, gradients rounded to fp16 between layers (the mixed_float16 policy's backward)
void launch_mlp_bwd_f16x3(const MlpBwdArgs& a, bool dx, bool single_pass, int num_cus, hipStream_t stream,
                          bool xyz_only = false);
void mlp_bwd_f16x3_set_attributes();
void build_bwd_gather(int n_angles, bool dx, bool hi_only, int32_t* idx /* kBwdStreamBytes / 2 */);
void launch_repack_bwd(const float* blob, const int32_t* idx, void* stream, hipStream_t s);

// aux_kernels.hip
void launch_raygen(const float c2w_host[16], float fov, int H, int W,
                   long long ray_begin, long long ray_count, float* orig /*nullable*/, float* dirs,
                   hipStream_t stream);
void launch_z_values(float near_b, float far_b, long long N, int S, const float* u, uint64_t seed,
                     long long ray_base, float* z, hipStream_t stream);
size_t sample_pdf_lds_bytes(int S, int Sf);
void launch_sample_pdf(const float* weights, const float* z, long long N, int S, int Sf, const float* u,
                       uint64_t seed, long long ray_base, float* z_new, float* z_merged,
                       hipStream_t stream);
void launch_composite(const float* raw, const float* z, long long N, int S, float* rgb, float* weights,
                      float* cumprod, float* alpha, float* rgb_samples, float* depth, hipStream_t stream);
void launch_posenc(const float* x, long long M, int n_enc, int passthrough, float* out, hipStream_t stream);

}  // namespace nerf

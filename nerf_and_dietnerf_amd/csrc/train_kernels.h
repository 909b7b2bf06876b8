// train_kernels.h -- internal declarations of the training path (NeRF.train_step, src/NeRF.py:136-178):
// layer-wise fp32 MFMA GEMMs over activations kept in HBM + the per-ray backward kernels.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frag_layout.h"

namespace nerf {

// padded training layout of the activations (floats per row)
constexpr int kLdC4 = 320;    // [h4 (256) | xyz_enc (33) | 0-pad to 64]   input of layer 4; cols 256.. = input of layer 0
constexpr int kLdC8 = 288;    // [h8 (256) | dir_enc (24 or 16) | 0-pad to 32]   input of layers 8 and 10
constexpr int kXyzPad = 64;
constexpr int kDirPad = 32;
constexpr int kTrainSplits = 128;   // row slabs of the weight-gradient reduction
constexpr int kTrainSplitsWide = 256;   // ... with the 256 x 256 tile (one tile per slab for a 256 x 256 layer)

// Fragment-major activation / gradient buffers (the fused training kernels): frag_layout.h::frag_index

enum { EPI_FWD_LEAKY = 0, EPI_FWD_LINEAR = 1, EPI_BWD_MASK = 2, EPI_BWD_PLAIN = 3 };

// Out[M x N] = epi( A[M x K] . Bt[N x K]^T )      (both operands K-contiguous)
struct GemmAbt {
    const float* A; int lda;
    const float* Bt; int ldb;
    float* Out; int ldo;
    long long M;                // multiple of 128
    int N, K;                   // N multiple of the column tile (128 wide / 32 narrow), K multiple of 32
    const float* bias;          // FWD_*: [N]
    const float* H; int ldh;    // BWD_MASK: stored activation of the layer whose pre-activation gradient is produced
    const float* r1a; int r1a_ld;   // BWD_MASK optional rank-1 term  r1a[m*r1a_ld] * r1b[n]
    const float* r1b;
    int n_valid;                // columns actually stored
    int accumulate;             // BWD_PLAIN: Out += result
    float alpha;
    unsigned* gmax;             // BWD_MASK optional: atomicMax of the bits of max|Out| (for the split-fp16 weight gradient)
    // split-fp16 variant (launch_gemm_abt_h): Bt pre-split into fp16 hi / lo planes of the same shape; max|A| slots
    const uint16_t* Bhi; const uint16_t* Blo;
    const unsigned* gmax_in;
};
void launch_gemm_abt(int epi, bool narrow, const GemmAbt& g, hipStream_t s);
// the data-gradient GEMM (EPI_BWD_MASK, 128-wide tiles) on the fp16 matrix cores: A (a gradient buffer, scaled by a power
// of two from gmax_in) is split while it is staged, Bt comes pre-split; three MFMA passes, fp32 accumulation
void launch_gemm_abt_h(const GemmAbt& g, hipStream_t s);

// partial[split][k][n] = sum over the split's rows of A[m][k] * G[m][n]; row Kp of every split = column sums of G
struct GemmAtb {
    const float* A; int lda; int K;     // K = columns of A used (multiple of 4)
    const float* G; int ldg; int N;     // N = columns of G used (multiple of 4)
    float* partial; int Kp; int Nw;     // partial: [splits][Kp + 1][Nw]
    long long M; int rows_per_split;    // multiple of 16
    const unsigned* gmax;               // split-fp16 variant: bits of max|G| (written by G's producer); G is scaled to fp16 range
    int a_f16;                          // head_wgrad: A holds fp16 elements (lda in halfs); gemm_atb_f16: both operands do
    int frag;                           // A and G (not the heads' 4-wide G) are fragment-major (frag_layout.h::frag_index); rows_per_split % 32 == 0
    // pair16 gradient operand of the fused float32-policy trainer (gemm_atb_p, train_kernels.hip; nerf_kernels.h::kPair16):
    // G's fp32 slots hold the backward chain's packed operand as fp16 (hi, lo) pairs -- the 16 bytes of four consecutive
    // features of a row are {hi01, hi23, lo01, lo23} -- carrying one power-of-two scale per row:
    // true G = (hi + lo) * g_rs[row] (g_rs: upper half of the factor's fp32 bits).  A stays fp32.
    const uint16_t* g_rs;               // non-null: G is pair16 (gemm_atb_p)
    // The sigma head's weight gradient as a by-product of layer 8's GEMM (A = C8 is exactly the head's input, [h8 | dir_enc]):
    // the threads that stage A also accumulate sum_rows A[row][k] * sig_g[row] on the VALU (fp32), one extra 16-byte load
    // per thread and step instead of a second pass over C8 (0.9 GB per step under the float32 policy).  128-wide tile, one
    // n tile only.  sig_partial: [splits][Kp + 1] (row Kp = sum of sig_g, the head's bias gradient)
    const float* sig_g;                 // (M) d_sigma per sample row (MlpBwdArgs::dsig)
    float* sig_partial;
};
constexpr int kWgradBatchMax = 10;   // GEMMs per batched weight-gradient launch (the eight 256-wide layers of a pass fit)
struct GemmAtbBatch { int n; int wg_end[kWgradBatchMax]; GemmAtb e[kWgradBatchMax]; };
void launch_gemm_atb(const GemmAtb& g, hipStream_t s);
// same contract on the fp16 matrix cores: both operands split hi + lo (22 bits) on the fly while they are staged into LDS,
// three MFMA passes, fp32 accumulation; G is pre-scaled by a power of two so that its largest entry sits at 2^14
void launch_gemm_atb_h(const GemmAtb& g, hipStream_t s, bool wide = false);   // wide: 256 x 256 tile, 512 threads
void launch_gemm_atb_h_batch(GemmAtbBatch& b, hipStream_t s, bool wide);      // all entries in ONE launch (fills wg_end)
void launch_gemm_atb_p(const GemmAtb& g, hipStream_t s, bool wide = false);   // pair16 operands (g.g_rs), same contract
void launch_gemm_atb_p_batch(GemmAtbBatch& b, hipStream_t s, bool wide);
void launch_head_wgrad(const GemmAtb& g, hipStream_t s);   // N = 4 (the heads): VALU kernel, same partial layout
// mixed_float16 policy: A and G are fp16 rows (lda / ldg in halfs), one MFMA pass, no scaling (G carries the loss scale)
void launch_gemm_atb_f16(const GemmAtb& g, hipStream_t s, bool wide = false);
void launch_gemm_atb_f16_batch(GemmAtbBatch& b, hipStream_t s, bool wide);

// grad[blob layout] = sum over splits of partial (deterministic order)
struct ReduceArgs {
    const float* partial; int Kp; int Nw; int splits;
    float* grad_w; float* grad_b;       // destinations inside the gradient blob
    int K_real, N_real; int n_src_off;  // gradient column n comes from partial column n_src_off + n
    int rowmap;                         // 0: identity; 1: layer 4 (blob rows [xyz(33); hidden(256)] <- training rows [hidden; xyz])
    int accumulate;                     // 1: add to the gradient blob instead of overwriting it
};
void launch_reduce_grad(const ReduceArgs& a, hipStream_t s);
struct ReduceBatch { int n; ReduceArgs e[kWgradBatchMax]; };
bool reduce_grad_is_wide(const ReduceArgs& a);                       // takes the 16-byte-load kernel (batchable)
void launch_reduce_grad_batch(const ReduceBatch& b, hipStream_t s);  // one launch, grid row e = entry e

// blob (Keras order) -> padded training matrices W [Kp x Np], WT [Np x Kp], bias [Np]
struct RelayoutArgs {
    const float* w; const float* b; int K_real, N_real, Kp, Np, rowmap;
    float* W; float* WT; float* bias;
    uint16_t* Whi; uint16_t* Wlo;       // W split into fp16 hi / lo planes (same [Kp x Np] layout)
};
void launch_relayout(const RelayoutArgs& a, hipStream_t s);

// rows [row0, row0 + M) of the sample grid (rays x S, or xyz/view rows in xyz_mode) -> local rows 0..M of C4/C8
void launch_train_encode(const float* o, const float* d, const float* z, long long row0, long long M, int S,
                         long long Mp, int n_angles, int xyz_mode, float* C4, float* C8, hipStream_t s,
                         bool half_out = false,    // half_out: C4 / C8 are fp16 rows of the same element pitch
                         bool frag = false);       // frag: C4 / C8 are fragment-major
// Optimizer / loss-scale state kept ON THE DEVICE: the verdict of a step's gradients (mixed_float16 policy: Keras 2.7
// LossScaleOptimizer, src/NeRF.py:159-163) gates that step's Adam update and moves the loss scale of the next step
// without the host reading anything back -- the host keeps enqueuing steps ahead of the GPU.
struct OptState {
    float scale, inv_scale;      // current loss scale (1 under the float32 policy)
    float adam_corr;             // sqrt(1 - beta_2^t) / (1 - beta_1^t) for the NEXT update, t = iterations + 1
    int finite;                  // being collected by unscale_check for the gradients in flight (1 = all finite so far)
    int apply_ok;                // verdict of the latest gradients: Adam applies them only if set
    int good, growth;            // finite steps since the scale last changed / finite steps that double it
    int dynamic;                 // 1 under the mixed_float16 policy
    long long iterations;        // Adam updates applied
    long long skipped;           // steps dropped because their gradients were not finite
};
void launch_mse(const float* rgb, const float* target, long long N, const OptState* st, float weight, float* d_rgb, float* mse_out,
                hipStream_t s);         // d_rgb carries st->scale (LossScaleOptimizer.get_scaled_loss)
void launch_unscale_check(float* ga, float* gb /* nullable */, size_t n, OptState* st, hipStream_t s,
                          bool check_only = false,     // check_only: finiteness test alone (after a gradient all-reduce)
                          const float* add_a = nullptr, const float* add_b = nullptr);   // blobs added after the test
void launch_metrics_accum(const float* scal, bool fine, float w0, float w1, double* acc /* [4]: loss, psnr_c, psnr_f, steps */,
                          hipStream_t s);
// out[i] = idx[i] ? blob[idx[i] - 1] : 0 (device-side re-pack of an operand stream whose packer only moves values)
void launch_gather_blob(const float* blob, const int32_t* idx, float* out, size_t n, hipStream_t s);
void launch_scale_by_loss_scale(const float* in, long long n, const OptState* st, float* out, hipStream_t s);
void launch_opt_begin(OptState* st, hipStream_t s);                            // before a gradient computation: finite = 1
void launch_opt_verdict(OptState* st, hipStream_t s);                          // after the gradients: scale bookkeeping, apply_ok
void launch_opt_tick(OptState* st, float beta1, float beta2, hipStream_t s);   // after Adam: iterations, adam_corr
void launch_composite_bwd(const float* raw, const float* z, const float* T, long long N, int S, const float* d_rgb,
                          const float* d_w_ext, float* Graw, float* d_z, hipStream_t s);
void launch_head_bwd(const float* Graw, const float* W9 /*[128][Np9] row-major, Np9 = 32*/, const float* H9,
                     long long M, float alpha, float* G9, unsigned* gmax, hipStream_t s);
void launch_pe_bwd(const float* dA0, const float* dA0b /* added to dA0, or null */, const float* o, const float* d,
                   const float* z, long long N, int S, float* d_z, hipStream_t s,
                   bool frag = false);     // frag: dA0 / dA0b are fragment-major (the fused backward chain's dx buffers)
void launch_unmerge_grad(const float* z_new, const float* z_c, const float* d_zm, long long N, int S, int Sf, float* d_zf,
                         hipStream_t s);
void launch_sample_pdf_bwd(const float* weights, const float* z, long long N, int S, int Sf, const float* u,
                           uint64_t seed, long long ray_base, const float* d_zf, float* d_w, hipStream_t s);
size_t sample_pdf_bwd_lds_bytes(int S, int Sf);
void launch_adam(float* w, float* m, float* v, const float* g, size_t n, float lr, float beta1, float beta2, float eps,
                 const OptState* st, hipStream_t s);      // no-op unless st->apply_ok; lr_t = lr * st->adam_corr

}  // namespace nerf

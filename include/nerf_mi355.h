/*
 * nerf_mi355.h -- C ABI of libnerf_mi355.so, the MI355X (gfx950) NeRF volumetric renderer.
 *
 * Drop-in boundary for the render hot path of Sahar-E/NeRF-and-DietNeRF.  The reference has no
 * FFI of its own; its seam is the late-bound Python call `src.UtilsNeuralRadianceField.render_rays`
 * (src/NeRF.py:180-188).  Each entry point below names the reference function it replaces.
 * Plain pointers and sizes only: no torch / TensorFlow types cross this boundary.
 *
 * Conventions
 *   - all arrays are C-contiguous fp32 unless stated; rays are (N,4) homogeneous rows exactly as the
 *     reference passes them (origin w=1, direction w=0); only xyz is read (src/UtilsNRF.py:204).
 *   - `mem` says where EVERY pointer of that call lives: NERF_MEM_HOST (library stages through its
 *     own device arena) or NERF_MEM_DEVICE (pointers are used in place on the ctx's stream).
 *   - every function returns 0 on success, non-zero on error; nerf_last_error() gives the
 *     thread-local message.  The library never aborts the process and never falls back to a CPU path.
 *   - a ctx is single-caller (not re-entrant), owns one HIP stream, the device copies of both
 *     networks' weights and a scratch arena; the caller owns all in/out buffers.
 *   - calls are synchronous on return for NERF_MEM_HOST; for NERF_MEM_DEVICE they are enqueued on the
 *     ctx stream and the caller synchronises with nerf_ctx_synchronize() (or its own stream, if it
 *     installed one with nerf_ctx_set_stream()).
 */
#ifndef NERF_MI355_H
#define NERF_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERF_ABI_VERSION 5

enum { NERF_NET_COARSE = 0, NERF_NET_FINE = 1 };
enum { NERF_MEM_HOST = 0, NERF_MEM_DEVICE = 1 };
/* arithmetic of the 256-wide contractions; everything else is always fp32 */
enum {
    NERF_PRECISION_FP32 = 0,   /* v_mfma_f32_32x32x2_f32: exact fp32 fma chains (parity mode)      */
    NERF_PRECISION_F16X3 = 1,  /* 3-pass split-fp16 MFMA (hi*hi + hi*lo + lo*hi), fp32 accumulate */
    NERF_PRECISION_F16 = 2     /* 1-pass fp16 MFMA, fp32 accumulate, activations rounded to fp16 between layers: the
                                  numerics class of the reference's production policy (mixed_float16,
                                  src/ExecutionRun.py:220-221); NOT the fp32 parity mode */
};

/* Network + frustum description: the 9 net/render keys of src/ConfigurationKeys.py:64-111. */
typedef struct nerf_config {
    int32_t n_pos_enc_xyz;    /* n_pos_enc_dim_xyz   (5)   */
    int32_t n_pos_enc_dir;    /* n_pos_enc_view_dir  (4)   */
    int32_t n_angles;         /* n_angles_for_model  (2)   */
    int32_t hidden_dim;       /* hidden_layer_dim    (256) */
    int32_t last_hidden_dim;  /* last_hidden_layer_dim (128) */
    float leaky_relu_alpha;   /* leaky_relu_alpha    (0.05) */
    float near_boundary;      /* NeRF.near_boundary, src/NeRF.py:45 */
    float far_boundary;       /* NeRF.far_boundary,  src/NeRF.py:46 */
    int32_t precision;        /* NERF_PRECISION_*    */
    int32_t device;           /* HIP device ordinal  */
} nerf_config;

/* Output set of render_rays()/render(); any pointer may be NULL (= not wanted).
 * Shapes for N rays and S samples of the LAST pass (S = Sc if no fine net, else Sc+Sf). */
typedef struct nerf_outputs {
    float* rgb;          /* (N,3)   render_result           src/UtilsNRF.py:114  */
    float* weights;      /* (N,S)   alpha * cumprod         :113                 */
    float* cumprod;      /* (N,S)   exclusive transmittance :112                 */
    float* alpha;        /* (N,S)                           :111                 */
    float* rgb_samples;  /* (N,S,3) sigmoid(net rgb)        :101                 */
    float* z;            /* (N,S)   sample depths           src/NeRF.py:132-134  */
    float* depth;        /* (N)     sum_s w*z               src/ExecutionRun.py:346 (optional 7th) */
} nerf_outputs;

typedef struct nerf_ctx nerf_ctx;

/* ---- lifecycle --------------------------------------------------------------------------- */
int nerf_abi_version(void);
const char* nerf_last_error(void);
/* replaces NeRF.__init__/init_network (src/NeRF.py:27-79): validates cfg, creates stream + arena */
int nerf_ctx_create(const nerf_config* cfg, nerf_ctx** out);
void nerf_ctx_destroy(nerf_ctx* ctx);
int nerf_ctx_synchronize(nerf_ctx* ctx);
/* run on a caller-owned hipStream_t (e.g. torch's current stream; NULL = HIP's default stream);
 * NERF_STREAM_OWN restores the ctx's own stream */
#define NERF_STREAM_OWN ((void*)(intptr_t)-1)
int nerf_ctx_set_stream(nerf_ctx* ctx, void* hip_stream);
/* change the frustum (near/far) or precision after creation */
int nerf_ctx_set_bounds(nerf_ctx* ctx, float near_boundary, float far_boundary);
int nerf_ctx_set_precision(nerf_ctx* ctx, int precision);

/* replaces Keras load_weights / model.get_weights() order (src/ExecutionRun.py:228-231):
 * `blob` = the 22 tensors of one network, kernel(in,out) row-major then bias, layer order of
 * src/NeRF.py:319-337 (dense .. dense_10).  HOST pointer.  n_floats must equal nerf_blob_size(). */
size_t nerf_blob_size(const nerf_config* cfg);
int nerf_load_weights(nerf_ctx* ctx, int which, const float* blob, size_t n_floats);

/* ---- the functions on the path, one entry each ---------------------- */
/* get_rays_directions, src/UtilsCV.py:467-499.  c2w row-major (4,4) HOST; dirs (H*W,4). */
int nerf_get_rays_directions(nerf_ctx* ctx, const float* c2w, float fov, int32_t H, int32_t W,
                             float* dirs, int mem);
/* get_z_values(near,far,N,1,S)[:,0,:], src/UtilsCV.py:565-581.  u (N,S) uniform draws or NULL
 * (= on-device Philox keyed by seed and global ray index ray_base+r).  z (N,S). */
int nerf_get_z_values(nerf_ctx* ctx, int64_t N, int32_t S, const float* u, uint64_t seed,
                      int64_t ray_base, float* z, int mem);
/* get_z_vals_from_prob_dist_func, src/UtilsCV.py:502-539.  weights,z (N,S); u (N,Sf) or NULL;
 * z_new (N,Sf) sorted.  If z_merged != NULL also writes sort(concat(z_new,z)) (N,S+Sf)
 * (src/NeRF.py:132). */
int nerf_sample_pdf(nerf_ctx* ctx, const float* weights, const float* z, int64_t N, int32_t S,
                    int32_t Sf, const float* u, uint64_t seed, int64_t ray_base, float* z_new,
                    float* z_merged, int mem);
/* positional_encoding_for_xyz / _for_views, src/UtilsNRF.py:52-85 (standalone, for tests/tools;
 * the render path computes the encoding in-register inside the MLP kernel). x (M,3). */
int nerf_positional_encoding(nerf_ctx* ctx, const float* x, int64_t M, int32_t n_enc,
                             int32_t with_passthrough, float* out, int mem);
/* model_predict, src/UtilsNRF.py:214-234 + Keras model call src/NeRF.py:316-339.
 * xyz (M,3), view_dirs (M,3) -> raw (M,4) [r,g,b,sigma]. */
int nerf_model_predict(nerf_ctx* ctx, int which, const float* xyz, const float* view_dirs,
                       int64_t M, float* raw, int mem);
/* ray_marching, src/UtilsNRF.py:88-115.  raw (N,S,4), z (N,S). */
int nerf_ray_marching(nerf_ctx* ctx, const float* raw, const float* z, int64_t N, int32_t S,
                      const nerf_outputs* outs, int mem);
/* render_rays, src/UtilsNRF.py:181-211 (the seam NeRF.render_rays binds, src/NeRF.py:180-188). */
int nerf_render_rays(nerf_ctx* ctx, int which, const float* rays_orig, const float* rays_dirs,
                     const float* z, int64_t N, int32_t S, const nerf_outputs* outs, int mem);
/* NeRF.render, src/NeRF.py:109-134: stratified coarse pass -> inverse-CDF resample -> fine pass on
 * sort(concat).  Sf == 0 (or no fine weights loaded) = coarse only.  u_coarse (N,Sc), u_fine (N,Sf)
 * or NULL for on-device Philox(seed, ray_base + r). */
int nerf_render(nerf_ctx* ctx, const float* rays_orig, const float* rays_dirs, int64_t N,
                int32_t Sc, int32_t Sf, const float* u_coarse, const float* u_fine, uint64_t seed,
                int64_t ray_base, const nerf_outputs* outs, int mem);
/* NeRF.render_image, src/NeRF.py:190-246, for the ray slab [ray_begin, ray_begin+ray_count) of the
 * row-major H*W image (ray_count <= 0 = whole image).  Outputs are slab-sized.  batch = rays per
 * internal pass (0 = library default: the whole slab up to 262144 rays; a quarter of it when per-sample outputs go to
 * host memory, so that copies overlap compute); results do not depend on it.  u_* index by GLOBAL ray. */
int nerf_render_image(nerf_ctx* ctx, const float* c2w, float fov, int32_t H, int32_t W,
                      int64_t ray_begin, int64_t ray_count, int64_t batch, int32_t Sc, int32_t Sf,
                      const float* u_coarse, const float* u_fine, uint64_t seed,
                      const nerf_outputs* outs, int mem);

/* ABI 3: page-locked (pinned) host memory for the buffers of NERF_MEM_HOST calls.  Any host memory works; with
 * buffers from nerf_host_alloc the outputs of nerf_render_image leave the device by DMA on a second stream WHILE the
 * next batch of rays computes (the reference concatenates its per-batch results at the end, src/NeRF.py:226-237), and
 * the host-memory entry point runs at the device-resident rate.  Not tied to a ctx; usable from any GPU of the node. */
int nerf_host_alloc(size_t bytes, void** out);
int nerf_host_free(void* p);

/* ---- multi-GPU assembly from C ------------------------------------------
 * One process (or thread + ctx) per GPU.  Rank 0 obtains an id and hands it to the others by any means (file, MPI,
 * a torch store); every rank then joins.  nerf_render_image_sharded renders this rank's contiguous slab of the H*W
 * rays (equal slabs of ceil(H*W/world) rays, as nerf_and_dietnerf_amd/sharding.py), all-gathers the RGB slabs with ONE
 * ncclAllGather on the ctx stream and writes the whole (H*W,3) image on every rank.  The Philox counter is the
 * global ray index: the image does not depend on the number of GPUs.  RCCL is bound at run time (dlopen); the
 * environment variable NERF_RCCL_LIB names another library with the same six nccl* entry points (a site's own RCCL
 * build; the tests' two-ranks-on-one-GPU stand-in). */
#define NERF_COMM_ID_BYTES 128
int nerf_comm_unique_id(void* id /* NERF_COMM_ID_BYTES, out */);
int nerf_comm_init(nerf_ctx* ctx, const void* id, int32_t rank, int32_t world);
int nerf_comm_destroy(nerf_ctx* ctx);
int nerf_render_image_sharded(nerf_ctx* ctx, const float* c2w, float field_of_view, int32_t H, int32_t W,
                              int64_t batch, int32_t n_coarse, int32_t n_fine, uint64_t seed,
                              float* rgb /* (H*W,3) */, int mem);
/* ABI 4: the same assembly for EVERY requested output of NeRF.render_image (src/NeRF.py:239-246) -- one ncclAllGather of
 * equal padded slabs per non-NULL pointer of `outs`; each destination holds the WHOLE image
 * ((H*W,3), (H*W,S), (H*W,S,3), (H*W) for depth; S = n_coarse + n_fine, or n_coarse without a fine network) on every
 * rank.  What the reference's video loop needs per frame is weights and z (depth = sum_s w*z, src/ExecutionRun.py:339-356)
 * -- or the fused `depth` output alone; its special ray plots take all six (:487).  Host destinations leave on the
 * ctx's copy stream, one output's copy under the next output's gather (page-locked buffers from nerf_host_alloc move by
 * DMA).  nerf_render_image_sharded is this call with rgb alone. */
int nerf_render_image_sharded_outputs(nerf_ctx* ctx, const float* c2w, float field_of_view, int32_t H, int32_t W,
                                      int64_t batch, int32_t n_coarse, int32_t n_fine, uint64_t seed,
                                      const nerf_outputs* outs, int mem);

/* ---- status ------------------------------------------------------------------------------ */
/* Synchronises and returns (then clears) the number of sample rows whose network output was not finite
 * since the last read.  NERF_PRECISION_F16X3 needs |activations| < 65504 (fp16 range); a non-zero count
 * there means: switch this model to NERF_PRECISION_FP32.  The reference has no such check (TF propagates
 * NaN silently). */
int nerf_ctx_read_nonfinite(nerf_ctx* ctx, int64_t* rows);

/* ---- training ---------------------------------------------------
 * Replaces NeRF.train_step (src/NeRF.py:136-178) under model.compile(optimizer=Adam(lr))
 * (src/ExecutionRun.py:226-227), fp32 policy:
 *   z = get_z_values(jitter); coarse render -> MSE; z_from_dist = inverse-CDF(weights_coarse) -- differentiated
 *   through, as the reference's tape does (no stop_gradient in src/UtilsCV.py:502-539); fine render on the
 *   Sf new samples only -> MSE; loss = sum; gradients of both networks; Adam; metrics loss/psnr_coarse/psnr_fine.
 * Weights, gradients and Adam moments are flat blobs in Keras get_weights() order (as nerf_load_weights). */
typedef struct nerf_train_config {
    float learning_rate;       /* Adam(optimizer_lr), src/ExecutionRun.py:226 */
    float beta_1, beta_2;      /* Keras defaults 0.9, 0.999 */
    float epsilon;             /* Keras default 1e-7 */
    int32_t sampler_gradient;  /* 1 = reference behaviour; 0 = treat z_from_dist as data (classic NeRF) */
    /* ABI 2: the reference's production policy (mixed_float16, src/ExecutionRun.py:220-221; loss-scaled branch of
     * train_step, src/NeRF.py:159-163; LossScaleOptimizer, src/ExecutionRun.py:262).  0 = the fp32 policy (fp32-class
     * products).  1 = fp16 compute: forward and data gradients with ONE fp16 MFMA pass per product, activations /
     * gradients rounded to fp16 between layers, fp32 accumulation, fp32 master weights and weight gradients; the
     * loss is scaled before the backward pass, gradients are unscaled and tested: a step with a non-finite gradient is
     * SKIPPED and halves the scale, dynamic_growth_steps finite steps in a row double it (Keras 2.7 dynamic loss
     * scaling: initial 2^15, growth interval 2000).  All three network variants (n_angles 2, 1, 0). */
    int32_t mixed_float16;
    float initial_loss_scale;      /* 0 -> 32768 */
    int32_t dynamic_growth_steps;  /* 0 -> 2000 */
} nerf_train_config;

/* Starts a trainer on the weights currently loaded (coarse required, fine optional); zero Adam moments.  Called while a
 * trainer is running it restarts the optimizer on the TRAINED weights (and resets the loss weights to 1, 1).
 * Rendering between optimizer steps is allowed at any time (DietNeRF's consistency render, the epoch plots,
 * src/ExecutionRun.py:193-201): the render path's operand streams are re-packed from the trained weights on the device,
 * enqueued on the ctx stream, without a host round trip. */
int nerf_train_begin(nerf_ctx* ctx, const nerf_train_config* cfg);
/* Packs the trained weights for the render path and frees optimizer state and activation buffers. */
int nerf_train_end(nerf_ctx* ctx);
int nerf_train_set_learning_rate(nerf_ctx* ctx, float learning_rate);
/* ABI 5: the ray loss of nerf_train_step / nerf_train_gradients is coarse_mse_weight * MSE(coarse render) +
 * fine_mse_weight * MSE(fine render).  (1, 1) -- the state nerf_train_begin leaves -- is NeRF.train_step
 * (src/NeRF.py:151,157).  DietNeRF's train_step builds its ray loss differently (src/DietNeRF.py:160-170,
 * `loss = loss_for_rays` BEFORE `loss_for_rays += <fine MSE>`, then `loss += loss_for_rays`): 2 * MSE(coarse) +
 * MSE(fine) is what its tape differentiates -- (2, 1) here.  The weights act on the gradients and on the `loss` metric
 * (and its running sum); psnr_coarse / psnr_fine stay the plain per-pass values.  Finite, >= 0. */
int nerf_train_set_loss_weights(nerf_ctx* ctx, float coarse_mse_weight, float fine_mse_weight);
/* mixed_float16 policy: the current loss scale, the optimizer steps applied and the steps skipped so far (1 / n / 0
 * under the fp32 policy).  Any pointer may be NULL. */
int nerf_train_loss_scale(nerf_ctx* ctx, float* loss_scale, int64_t* steps_applied, int64_t* steps_skipped);
/* One NeRF.train_step on N rays: rays_orig/rays_dirs (N,4), target_rgb (N,3); u_coarse (N,Sc) / u_fine (N,Sf)
 * NULL -> on-device Philox(seed, ray index in the batch).  metrics (host, nullable): loss, psnr_coarse, psnr_fine;
 * passing it synchronises.  Sf = 0 (or no fine network) trains the coarse network alone (src/NeRF.py:153).
 * With a communicator (nerf_comm_init, world > 1) the step is data-parallel: every rank passes its own shard of the
 * batch and the two gradient blobs are averaged with one ncclAllReduce each before the (identical) Adam update;
 * metrics are this rank's.  Under mixed_float16 the finiteness test is repeated on the reduced blobs, so a non-finite
 * shard on ANY rank makes EVERY rank skip the step and halve its loss scale. */
int nerf_train_step(nerf_ctx* ctx, const float* rays_orig, const float* rays_dirs, const float* target_rgb,
                    int64_t N, int32_t Sc, int32_t Sf, const float* u_coarse, const float* u_fine, uint64_t seed,
                    float* metrics, int mem);
/* ABI 4: Keras' History keeps the per-epoch MEANS of train_step's metrics (model.fit, src/ExecutionRun.py:186-201).  Every
 * nerf_train_step / nerf_train_gradients adds its loss, psnr_coarse and psnr_fine (the very values `metrics` would receive)
 * to running sums on the device; this call synchronises, returns the sums and the number of steps since the last read,
 * and clears them -- a training loop passes metrics = NULL per step (no synchronisation) and reads once per epoch. */
int nerf_train_read_metric_sums(nerf_ctx* ctx, double* sums /* [3]: loss, psnr_coarse, psnr_fine */, int64_t* steps);
/* The two halves of a step, for data-parallel training: gradients (kept in the ctx and optionally copied out
 * as blobs), then -- after the caller averaged them over ranks -- the Adam update (NULL = use the ctx's own).
 * mixed_float16 (ABI 3): nerf_train_gradients returns UNSCALED gradients and takes no verdict; nerf_train_apply tests
 * the blobs it is about to apply (the caller's all-reduced ones, or the ctx's own), skips a non-finite step and moves
 * the loss scale -- all on the device -- so every rank of a data-parallel job reaches the same verdict from the same
 * reduced blobs (a non-finite shard makes the sum non-finite everywhere). */
int nerf_train_gradients(nerf_ctx* ctx, const float* rays_orig, const float* rays_dirs, const float* target_rgb,
                         int64_t N, int32_t Sc, int32_t Sf, const float* u_coarse, const float* u_fine,
                         uint64_t seed, float* grad_coarse, float* grad_fine, float* metrics, int mem);
int nerf_train_apply(nerf_ctx* ctx, const float* grad_coarse, const float* grad_fine, int mem);
/* Backward through NeRF.render() itself (src/NeRF.py:109-134): the graph DietNeRF's consistency loss differentiates
 * when it renders an image under the tape (src/DietNeRF.py:204-222, render_image -> render per ray batch).  Unlike
 * train_step, the fine pass runs on sort(concat(z_fine, z_coarse)) (Sc + Sf samples) and only its rgb is an output.
 * d_rgb (N,3) = dL/d(render()[0]) supplied by the caller (e.g. from its embedding network); the call re-runs the
 * forward with activation stash on these N rays -- same draws as nerf_render with the same (seed, ray_base) -- and
 * leaves dL/d(weights) in the ctx's gradient blobs: overwriting them (accumulate = 0) or adding to what
 * nerf_train_gradients left there (accumulate = 1: the reference sums both losses before one Adam step), ready for
 * nerf_train_apply(ctx, NULL, NULL, mem).  The coarse network receives gradient only through the inverse-CDF sampler
 * (none with sampler_gradient = 0).  rgb_out / grad_coarse / grad_fine: optional copies.
 * mixed_float16 (ABI 4; the policy the reference always runs under, src/ExecutionRun.py:220-221): DietNeRF scales the
 * SUM of ray loss and consistency loss and unscales once (src/DietNeRF.py:142-153,192-202).  Here the caller's d_rgb is
 * multiplied by the current loss scale ON THE DEVICE, the single-pass fp16 chain runs on it (fp16 gradient buffers that
 * carry the scale, as in nerf_train_gradients), and the call leaves UNSCALED gradients: stored (accumulate = 0) or added to
 * the unscaled gradients nerf_train_gradients left (accumulate = 1) -- like with like.  The finiteness flag is reset by
 * accumulate = 0 (a new gradient computation; gradients that were computed and never applied do not decide this one's
 * verdict) and COLLECTS over nerf_train_gradients + accumulate = 1 calls; nerf_train_apply takes the one verdict on the
 * summed blobs: a non-finite d_rgb (or an overflow in the chain) skips the step and halves the scale. */
int nerf_train_render_gradients(nerf_ctx* ctx, const float* rays_orig, const float* rays_dirs, const float* d_rgb,
                                int64_t N, int32_t Sc, int32_t Sf, const float* u_coarse, const float* u_fine,
                                uint64_t seed, int64_t ray_base, int32_t accumulate, float* rgb_out,
                                float* grad_coarse, float* grad_fine, int mem);
/* ABI 5: the same graph in two calls, the activations kept in between -- for callers whose d_rgb depends on the WHOLE image
 * (DietNeRF: the embedding network sees all 150 x 150 pixels before any gradient exists, src/DietNeRF.py:215-221).  The reference
 * renders the image once, under the tape; with nerf_render_image + nerf_train_render_gradients the forward runs twice.  Here:
 *   nerf_train_render_forward(slot = b, batch b of the rays, ...) for every batch -- rgb_out is that batch's part of the image,
 *     the forward of the tape itself (under mixed_float16: the single-pass fp16 network), its activations stay in slot b
 *     (about 1.7 MB per ray at 55 + 110 rows under the float32 policy, half under mixed_float16: a 150 x 150 image is 38 / 19
 *     GB of the device's 288 GB; slots are grow-only buffers, nerf_train_render_release or nerf_train_end frees them);
 *   ... the caller turns the image into d_rgb ...
 *   nerf_train_render_backward(slot = b, d_rgb of batch b, accumulate, ...) for every batch: exactly the backward half of
 *     nerf_train_render_gradients -- bit-identical gradients, same loss-scale handling -- on the kept activations.
 * A slot holds one forward pass: a backward pass consumes it, an optimizer step (nerf_train_apply / nerf_train_step) invalidates
 * every slot (the activations belong to the weights that made them), a new forward into the slot overwrites it.  The slot keeps
 * its own copies of rays and draws.  slot: 0..4095. */
int nerf_train_render_forward(nerf_ctx* ctx, int32_t slot, const float* rays_orig, const float* rays_dirs, int64_t N,
                              int32_t Sc, int32_t Sf, const float* u_coarse, const float* u_fine, uint64_t seed,
                              int64_t ray_base, float* rgb_out, int mem);
int nerf_train_render_backward(nerf_ctx* ctx, int32_t slot, const float* d_rgb, int32_t accumulate, float* grad_coarse,
                               float* grad_fine, int mem);
int nerf_train_render_release(nerf_ctx* ctx);
/* ABI 3: the gradient blob of a network as the ctx holds it now -- after nerf_train_gradients /
 * nerf_train_render_gradients the gradients just computed, after a data-parallel nerf_train_step the all-reduced mean
 * the Adam update used (what the reference's tape.gradient returns, src/NeRF.py:159-165). */
int nerf_train_get_gradients(nerf_ctx* ctx, int which, float* blob, size_t n_floats, int mem);
/* Current weights of a network as a blob (model.get_weights(), src/UtilsFiles.py:153-164 saves these). */
int nerf_get_weights(nerf_ctx* ctx, int which, float* blob, size_t n_floats, int mem);

/* ---- measurement ------------------------------------------------------------------------- */
/* When enabled, every fused PE+MLP kernel launch is bracketed by HIP events on the ctx stream.
 * nerf_ctx_read_timing synchronises, returns the summed kernel time / launch count / MLP rows
 * since the last read, and resets the counters. */
int nerf_ctx_enable_timing(nerf_ctx* ctx, int on);
int nerf_ctx_read_timing(nerf_ctx* ctx, double* mlp_ms, int64_t* n_launches, int64_t* n_rows);

#ifdef __cplusplus
}
#endif
#endif /* NERF_MI355_H */

// train_api.hip -- C ABI of the training path (include/nerf_mi355.h, "training" section):
//   NeRF.train_step            src/NeRF.py:136-178   (loss, gradients of both networks, optimizer step, metrics)
//   Adam(optimizer_lr)         src/ExecutionRun.py:226 (Keras-2.7 defaults beta_1=.9 beta_2=.999 epsilon=1e-7)
// Orchestration only: every arithmetic step is a HIP kernel of train_kernels.hip / aux_kernels.hip on the
// context's stream.  Master weights, gradients and Adam moments live on the device as flat blobs in Keras
// get_weights() order; padded [K x N] / [N x K] copies feed the GEMMs and are rebuilt after every update.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "nerf_ctx.h"
#include "train_kernels.h"

using namespace nerf;

namespace nerf {

struct TLayer {
    int K_real, N_real, Kp, Np, rowmap;
    size_t w_off, b_off;           // offsets into the blob
    float *W, *WT, *bias;          // padded copies (device)
    uint16_t *Whi, *Wlo;           // W split into fp16 hi / lo planes (data gradient on the fp16 cores)
};

struct TNet {
    bool present = false;
    bool render_dirty = false;     // optimizer steps not yet packed into the render path's operand streams
    bool host_stale = false;       // ... and not yet copied to NetWeights::host_blob (the seed of the next trainer)
    float *blob = nullptr, *grad = nullptr, *m = nullptr, *v = nullptr, *mats = nullptr;
    void* fstream = nullptr;       // fused forward (f16x3 stash kernel): operand stream + constants, re-packed on device
    float* fcst = nullptr;
    void* bstream = nullptr;       // fused backward (mlp_bwd_f16x3): transposed operand stream, re-packed on device
    bool bdx = false;              // ... built with the encoding tiles (the fine network under sampler_gradient)
    int n_layers = 11;             // 11: xyz + view-direction network; 12: xyz-only network (n_angles_for_model = 0)
    TLayer L[12];
};

struct TPass {                      // activations of one pass, kept from forward to backward
    DevBuf C4, C8, H1, H2, H3, H5, H6, H7, H8b, H9, raw, T, w, rgb, z;   // H8b: xyz-only network's extra layer
    // fused backward: LeakyReLU' bit records of layers 0..8 (32 B per row and layer, written by the stash forward),
    // the pre-activation gradients D[0..7] (Mp x 256) and G9 = D[8] (Mp x 128), the two encoding-gradient parts
    DevBuf masks, D[10], dxa, dxb;     // D[8], D[9]: see MlpBwdArgs::d_ptr (the xyz-only network has ten gradient buffers)
    DevBuf rs;                         // pair16 gradient buffers (float32 policy): 10 x Mp row factors (MlpBwdArgs::rs_ptr)
};

// nerf_train_render_forward / _backward (ABI 5): the activations of one ray batch of NeRF.render(), kept from a forward to
// its backward while other batches run -- a whole source image of DietNeRF's consistency loss stays resident (150 x 150 rays x
// (55 + 110) rows: 38 GB under the float32 policy, 19 GB under mixed_float16 of this device's 288 GB) instead of being
// rendered once for the embedding network and a second time under the tape.  A slot owns the STASH side of two passes, its
// own copies of the rays and draws, and the new depths; the gradient side of a TPass (D, dxa, dxb, rs) is lent by
// TrainState::pass while the slot runs.
struct RenderSlot {
    TPass pass[2];
    DevBuf o, d, u_c, u_f, z_new;
    long long N = 0, ray_base = 0;
    int Sc = 0, Sf = 0;
    unsigned long long seed = 0;
    bool has_uc = false, has_uf = false, valid = false;
};
constexpr int kMaxRenderSlots = 4096;

struct TrainState {
    nerf_train_config cfg;
    bool training = false;          // set by nerf_train_begin
    bool fused_forward = false;     // forward pass on the fused split-fp16 kernel with activation stash (n_angles > 0)
    int32_t *sidx = nullptr, *cidx = nullptr;   // device gather tables of the fused kernel's stream / constants
    bool fused_backward = false;    // data gradients by the fused chain kernel (needs the fused forward's mask records)
    bool frag = false;              // fused forward + backward: activation / gradient buffers are fragment-major (frag_index)
    bool pair16 = false;            // ... and, under the float32 policy, the gradient buffers hold fp16 (hi, lo) pairs in their fp32 slots (nerf_kernels.h::kPair16)
    int32_t* bidx[2] = {nullptr, nullptr};      // gather tables of the backward stream: [0] plain, [1] with encoding tiles
    long long step = 0;
    size_t nblob = 0;
    TNet net[2];
    TPass pass[2];
    DevBuf Ga, Gb, G9, Graw, dA0, partial, d_rgb, d_wext, d_zf, tgt, o, d, u_c, u_f, scal, gmax;
    DevBuf dsig;                    // (Mp) column 3 of Graw as a vector, written by the fused backward chain (GemmAtb::sig_g)
    // The fine pass's batched weight-gradient launch on a second stream, beside the sampler / compositing backward and the coarse
    // pass's backward chain (default; NERF_TRAIN_OVERLAP=0 keeps one stream).  Its slab sums live in their own buffer, each pass
    // has its own max|D| slots, and the main stream joins before anything reads the fine network's gradient blob.  Both big
    // kernels want a whole CU per workgroup, so this is tail filling, not co-residency: -1.2 % on a mixed_float16 step, +-0
    // under the float32 policy; bit-identical results (tests/test_gpu_train.py::test_side_stream_...).
    bool overlap = false;
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool side_pending = false;
    DevBuf partial_side;
    bool wgrad_f16 = false;         // weight gradients on the fp16 matrix cores (gemm_atb_h), else exact fp32 MFMA
    bool dgrad_f16 = false;         // data gradients on the fp16 matrix cores (gemm_abt_h)
    bool wgrad_wide = true;         // 256 x 256 tile for the 256-wide layers' weight gradients
    // elements per row of the 256-wide / 128-wide activation and gradient buffers (padded pitches were measured against
    // L2-channel hot-spotting in round 2 and lost; the fused path's buffers are fragment-major now, frag_layout.h)
    int ldh = 256, ldh9 = 128;
    bool acc_grads = false;         // the running backward pass ADDS to the gradient blobs (nerf_train_render_gradients)
    // ray loss = loss_w[0] * MSE(coarse) + loss_w[1] * MSE(fine) (nerf_train_set_loss_weights): 1, 1 is NeRF.train_step
    // (src/NeRF.py:151,157); DietNeRF's ray loss counts the coarse term twice (src/DietNeRF.py:160-170)
    float loss_w[2] = {1.f, 1.f};
    // mixed_float16 policy (src/ExecutionRun.py:220-221, src/NeRF.py:159-163): single-pass fp16 forward / data gradients
    // and the dynamic loss scale of Keras' LossScaleOptimizer
    bool mixed = false;
    DevBuf opt;                     // OptState (train_kernels.h): loss scale, verdicts, Adam iteration count -- on the device
    DevBuf z_new, d_zm, zero_rgb;   // backward through NeRF.render(): the Sf new depths, d/dz of the merged fine pass
    DevBuf macc;                    // running sums of the step metrics (4 doubles: loss, psnr_coarse, psnr_fine, steps)
    DevBuf gsave[2];                // ... under mixed_float16 with accumulate = 1: the (unscaled) gradients already there
    // A render between optimizer steps (DietNeRF's consistency render every 13th step, the epoch plots) needs the render
    // path's three operand streams re-packed from the trained blob: on the DEVICE, by gather tables built once from the host
    // packers (round 4; the device -> host -> pack x 3 -> device round trip this replaces cost ~30 ms and two synchronisations
    // per render).  rt_h / rt_h1: build_f16x3_gather (3-pass / hi-only stream), rt_ch their constants; rt_f / rt_cf: the fp32
    // stream and constants (pack_weights_fp32 only moves values: the table is the packed INDEX blob).
    int32_t *rt_h = nullptr, *rt_h1 = nullptr, *rt_ch = nullptr, *rt_f = nullptr, *rt_cf = nullptr;
    std::vector<RenderSlot> slots;  // nerf_train_render_forward / _backward
};

}  // namespace nerf

namespace {

int layer_table(const nerf_config& cfg, TLayer L[12]) {
    const int kd = 8 * (cfg.n_angles + 1);
    // {K_real, N_real, Kp, Np, rowmap}
    const int with_dirs[11][5] = {
        {33, 256, kXyzPad, 256, 0}, {256, 256, 256, 256, 0}, {256, 256, 256, 256, 0}, {256, 256, 256, 256, 0},
        {289, 256, kLdC4, 256, 1},  {256, 256, 256, 256, 0}, {256, 256, 256, 256, 0}, {256, 256, 256, 256, 0},
        {256 + kd, 128, kLdC8, 128, 0}, {128, 3, 128, 32, 0}, {256 + kd, 1, kLdC8, 32, 0}};
    // get_network_only_xyz (src/NeRF.py:248-288): ... h8 -> dense 256 -> dense 128 -> rgb; sigma from h8
    const int xyz_only[12][5] = {
        {33, 256, kXyzPad, 256, 0}, {256, 256, 256, 256, 0}, {256, 256, 256, 256, 0}, {256, 256, 256, 256, 0},
        {289, 256, kLdC4, 256, 1},  {256, 256, 256, 256, 0}, {256, 256, 256, 256, 0}, {256, 256, 256, 256, 0},
        {256, 256, 256, 256, 0}, {256, 128, 256, 128, 0}, {128, 3, 128, 32, 0}, {256, 1, 256, 32, 0}};
    const int n = cfg.n_angles == 0 ? 12 : 11;
    size_t off = 0;
    for (int l = 0; l < n; ++l) {
        const int* sh = cfg.n_angles == 0 ? xyz_only[l] : with_dirs[l];
        L[l].K_real = sh[0]; L[l].N_real = sh[1]; L[l].Kp = sh[2]; L[l].Np = sh[3]; L[l].rowmap = sh[4];
        L[l].w_off = off; off += (size_t)L[l].K_real * L[l].N_real;
        L[l].b_off = off; off += L[l].N_real;
        L[l].W = L[l].WT = L[l].bias = nullptr;
    }
    return n;
}

void free_buf(DevBuf& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }

int relayout_net(nerf_ctx* c, TNet& n) {
    if (n.fstream)
        launch_repack_f16x3(n.blob, c->train->sidx, n.fstream, c->train->cidx, n.fcst,
                            f16_stream_bytes(c->cfg.n_angles, c->train->mixed), c->stream);
    if (n.bstream) launch_repack_bwd(n.blob, c->train->bidx[n.bdx ? 1 : 0], n.bstream, c->stream);
    // the padded W / W^T / hi-lo planes feed the layer-wise GEMMs only (NERF_TRAIN_* switches); the fused forward + backward
    // read the two re-packed streams above and nothing else (22 launches per step that nobody read)
    if (c->train->frag) { HIP_OK(hipGetLastError()); return 0; }
    for (int l = 0; l < n.n_layers; ++l) {
        const TLayer& L = n.L[l];
        RelayoutArgs a;
        a.w = n.blob + L.w_off; a.b = n.blob + L.b_off;
        a.K_real = L.K_real; a.N_real = L.N_real; a.Kp = L.Kp; a.Np = L.Np; a.rowmap = L.rowmap;
        a.W = L.W; a.WT = L.WT; a.bias = L.bias; a.Whi = L.Whi; a.Wlo = L.Wlo;
        launch_relayout(a, c->stream);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

int alloc_optimizer(nerf_ctx* c, TrainState* t, TNet& n) {
    const size_t nb = t->nblob * sizeof(float);
    if (!n.grad) HIP_OK(hipMalloc((void**)&n.grad, nb));
    if (!n.m) HIP_OK(hipMalloc((void**)&n.m, nb));
    if (!n.v) HIP_OK(hipMalloc((void**)&n.v, nb));
    HIP_OK(hipMemsetAsync(n.m, 0, nb, c->stream));
    HIP_OK(hipMemsetAsync(n.v, 0, nb, c->stream));
    HIP_OK(hipMemsetAsync(n.grad, 0, nb, c->stream));
    return 0;
}

// fused-forward operands of one network (and, once, the gather tables they are re-packed with)
int ensure_fused(nerf_ctx* c, TrainState* t, TNet& n) {
    if (!t->fused_forward) return 0;
    if (!t->sidx) {
        std::vector<int32_t> si(f16_stream_bytes(c->cfg.n_angles, t->mixed) / 2), ci(kConstFloats);
        build_f16x3_gather(c->cfg.n_angles, t->mixed, si.data(), ci.data());
        HIP_OK(hipMalloc((void**)&t->sidx, si.size() * sizeof(int32_t)));
        HIP_OK(hipMalloc((void**)&t->cidx, ci.size() * sizeof(int32_t)));
        HIP_OK(hipMemcpy(t->sidx, si.data(), si.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(t->cidx, ci.data(), ci.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if (!n.fstream) HIP_OK(hipMalloc(&n.fstream, kStreamBytesF16Xyz));     // the largest of the four streams
    if (!n.fcst) HIP_OK(hipMalloc((void**)&n.fcst, kConstBytes));
    if (t->fused_backward && t->training) {
        // the fine network's chain also produces the gradient w.r.t. the xyz encoding when the sampler is differentiated
        n.bdx = (&n == &t->net[1]) && t->cfg.sampler_gradient != 0;
        int32_t*& bi = t->bidx[n.bdx ? 1 : 0];
        if (!bi) {
            std::vector<int32_t> idx(kBwdStreamBytes / 2);
            build_bwd_gather(c->cfg.n_angles, n.bdx, t->mixed, idx.data());
            HIP_OK(hipMalloc((void**)&bi, idx.size() * sizeof(int32_t)));
            HIP_OK(hipMemcpy(bi, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        if (!n.bstream) HIP_OK(hipMalloc(&n.bstream, kBwdStreamBytes));
    }
    return 0;
}

int init_net(nerf_ctx* c, TrainState* t, int which) {
    TNet& n = t->net[which];
    n.n_layers = layer_table(c->cfg, n.L);
    const size_t nb = t->nblob * sizeof(float);
    HIP_OK(hipMalloc((void**)&n.blob, nb));
    if (t->training)
        if (int r = alloc_optimizer(c, t, n)) return r;
    size_t mats = 0;
    for (int l = 0; l < n.n_layers; ++l) mats += 3 * (size_t)n.L[l].Kp * n.L[l].Np + n.L[l].Np;   // W, WT, (Whi + Wlo), bias
    HIP_OK(hipMalloc((void**)&n.mats, mats * sizeof(float)));
    float* p = n.mats;
    for (int l = 0; l < n.n_layers; ++l) {
        TLayer& L = n.L[l];
        L.W = p; p += (size_t)L.Kp * L.Np;
        L.WT = p; p += (size_t)L.Kp * L.Np;
        L.Whi = reinterpret_cast<uint16_t*>(p);
        L.Wlo = L.Whi + (size_t)L.Kp * L.Np;
        p += (size_t)L.Kp * L.Np;          // two half planes = one float plane
        L.bias = p; p += L.Np;
    }
    HIP_OK(hipMemcpyAsync(n.blob, c->net[which].host_blob.data(), nb, hipMemcpyHostToDevice, c->stream));
    n.present = true;
    n.render_dirty = n.host_stale = false;
    if (int r = ensure_fused(c, t, n)) return r;
    return relayout_net(c, n);
}

// ---- one pass: forward ---------------------------------------------------------------------------
struct PassDims { long long N; int S; long long M, Mp; };

int ensure_pass(nerf_ctx* c, TPass& p, const PassDims& d) {
    const size_t f = sizeof(float);
    // the mixed_float16 policy keeps activations and pre-activation gradients in fp16 (same element pitches, half the
    // bytes); the buffers are grow-only and are released by train_free before a trainer of the other policy starts
    const size_t ea = c->train && c->train->mixed && c->train->training ? 2 : f;
    int r = 0;
    r |= ensure(c, p.C4, d.Mp * kLdC4 * ea);
    r |= ensure(c, p.C8, d.Mp * kLdC8 * ea);
    DevBuf* hs[] = {&p.H1, &p.H2, &p.H3, &p.H5, &p.H6, &p.H7};
    const int ldh = c->train ? c->train->ldh : 256, ldh9 = c->train ? c->train->ldh9 : 128;
    for (DevBuf* h : hs) r |= ensure(c, *h, d.Mp * ldh * ea);
    r |= ensure(c, p.H9, d.Mp * ldh9 * ea);
    if (c->cfg.n_angles == 0) r |= ensure(c, p.H8b, d.Mp * 256 * f);      // (fp32-sized: also the layer-wise path's)
    r |= ensure(c, p.raw, d.Mp * 4 * f);
    r |= ensure(c, p.T, d.M * f);
    r |= ensure(c, p.w, d.M * f);
    r |= ensure(c, p.rgb, d.N * 3 * f);
    r |= ensure(c, p.z, d.M * f);
    if (c->train && c->train->fused_backward && c->train->training) {
        const bool xyz = c->cfg.n_angles == 0;
        r |= ensure(c, p.masks, (size_t)(xyz ? 10 : 9) * d.Mp * 32);
        for (int l = 0; l < 8; ++l) r |= ensure(c, p.D[l], d.Mp * ldh * ea);
        r |= ensure(c, p.D[8], d.Mp * (xyz ? ldh : ldh9) * ea);
        if (xyz) r |= ensure(c, p.D[9], d.Mp * ldh9 * ea);
        r |= ensure(c, p.dxa, d.Mp * kBwdXyzLd * f);
        r |= ensure(c, p.dxb, d.Mp * kBwdXyzLd * f);
        if (c->train->pair16) r |= ensure(c, p.rs, (size_t)10 * d.Mp * sizeof(uint16_t));
    }
    return r;
}

void fwd_layer(nerf_ctx* c, const TLayer& L, const float* A, int lda, float* Out, int ldo, long long Mp,
               bool linear_head = false, int n_valid = -1) {
    GemmAbt g{};
    g.A = A; g.lda = lda; g.Bt = L.WT; g.ldb = L.Kp; g.Out = Out; g.ldo = ldo;
    g.M = Mp; g.N = L.Np; g.K = L.Kp; g.bias = L.bias; g.alpha = c->cfg.leaky_relu_alpha;
    g.n_valid = n_valid < 0 ? L.Np : n_valid;
    launch_gemm_abt(linear_head ? EPI_FWD_LINEAR : EPI_FWD_LEAKY, linear_head, g, c->stream);
}

// the Dense stack over Mp encoded rows (C4 / C8 hold the encodings) -> raw_out (Mp x 4)
void forward_layers(nerf_ctx* c, TNet& n, TPass& p, long long Mp, float* raw) {
    float *C4 = (float*)p.C4.p, *C8 = (float*)p.C8.p;
    float *H1 = (float*)p.H1.p, *H2 = (float*)p.H2.p, *H3 = (float*)p.H3.p, *H5 = (float*)p.H5.p,
          *H6 = (float*)p.H6.p, *H7 = (float*)p.H7.p, *H9 = (float*)p.H9.p;
    fwd_layer(c, n.L[0], C4 + 256, kLdC4, H1, 256, Mp);
    fwd_layer(c, n.L[1], H1, 256, H2, 256, Mp);
    fwd_layer(c, n.L[2], H2, 256, H3, 256, Mp);
    fwd_layer(c, n.L[3], H3, 256, C4, kLdC4, Mp);          // h4 lands next to xyz_enc: the skip concat
    fwd_layer(c, n.L[4], C4, kLdC4, H5, 256, Mp);
    fwd_layer(c, n.L[5], H5, 256, H6, 256, Mp);
    fwd_layer(c, n.L[6], H6, 256, H7, 256, Mp);
    fwd_layer(c, n.L[7], H7, 256, C8, kLdC8, Mp);          // h8 lands next to dir_enc
    if (n.n_layers == 11) {
        fwd_layer(c, n.L[8], C8, kLdC8, H9, 128, Mp);
        fwd_layer(c, n.L[9], H9, 128, raw, 4, Mp, true, 3);          // rgb head   -> raw[:, 0:3]
        fwd_layer(c, n.L[10], C8, kLdC8, raw + 3, 4, Mp, true, 1);   // sigma head -> raw[:, 3]
    } else {                                                // xyz-only: h8 -> 256 -> 128 -> rgb; sigma from h8
        float* H8b = (float*)p.H8b.p;
        fwd_layer(c, n.L[8], C8, kLdC8, H8b, 256, Mp);
        fwd_layer(c, n.L[9], H8b, 256, H9, 128, Mp);
        fwd_layer(c, n.L[10], H9, 128, raw, 4, Mp, true, 3);
        fwd_layer(c, n.L[11], C8, kLdC8, raw + 3, 4, Mp, true, 1);
    }
}

int forward_pass(nerf_ctx* c, TrainState* t, int which, const PassDims& d, const float* o, const float* dirs) {
    TNet& n = t->net[which];
    TPass& p = t->pass[which];
    float *raw = (float*)p.raw.p, *z = (float*)p.z.p;
    launch_train_encode(o, dirs, z, 0, d.M, d.S, d.Mp, c->cfg.n_angles, 0, (float*)p.C4.p, (float*)p.C8.p, c->stream,
                        t->mixed, t->frag);
    if (t->frag && !n.fstream) return fail("internal: fused training forward without its weight stream");
    if (t->fused_forward && n.fstream) {
        // the render path's fused PE + MLP kernel (3-pass split fp16, fp32-class results) with every activation also
        // written to the buffers the backward GEMMs read: 4x the rate of the layer-wise forward
        MlpArgs a{};
        a.wstream = (const float*)n.fstream; a.wconst = n.fcst;
        a.in_a = o; a.in_b = dirs; a.z = z; a.raw = raw; a.nonfinite = c->nonfinite;
        a.M = d.M; a.S = d.S; a.mode = 0; a.alpha = c->cfg.leaky_relu_alpha;
        const bool xyz = c->cfg.n_angles == 0;
        float* dst[10] = {(float*)p.H1.p, (float*)p.H2.p, (float*)p.H3.p, (float*)p.C4.p, (float*)p.H5.p,
                          (float*)p.H6.p, (float*)p.H7.p, (float*)p.C8.p, xyz ? (float*)p.H8b.p : (float*)p.H9.p,
                          xyz ? (float*)p.H9.p : nullptr};
        const int ld[10] = {t->ldh, t->ldh, t->ldh, kLdC4, t->ldh, t->ldh, t->ldh, kLdC8, xyz ? t->ldh : t->ldh9, t->ldh9};
        for (int i = 0; i < (xyz ? 10 : 9); ++i) {
            a.st_ptr[i] = dst[i]; a.st_ld[i] = ld[i];
            a.mask_ptr[i] = p.masks.p ? (uint32_t*)p.masks.p + (size_t)i * d.Mp * 8 : nullptr;
        }
#ifdef NERF_DIAG_STASH_WRAP   // diagnostic BUILD only (make EXTRA=-DNERF_DIAG_STASH_WRAP): timing without HBM stores, wrong results
        a.diag_wrap = d.Mp >= 8192;
#endif
        launch_mlp_f16x3_stash(a, c->num_cus, c->stream, t->mixed, xyz);
    } else {
        forward_layers(c, n, p, d.Mp, raw);
    }
    launch_composite(raw, z, d.N, d.S, (float*)p.rgb.p, (float*)p.w.p, (float*)p.T.p, nullptr, nullptr, nullptr,
                     c->stream);
    HIP_OK(hipGetLastError());
    return 0;
}

// ---- one pass: backward --------------------------------------------------------------------------
// The 256-wide layers' weight gradients of a pass (fused backward: every D_l exists before the first of them starts) are
// queued and leave as ONE GEMM launch + ONE reduction launch: two rounds of workgroups over the chip with ~56 row slabs per
// layer instead of 256 (a quarter of the partial-sum bytes: 2.1 GB -> 0.45 GB per step), 2 kernel boundaries instead of 16.
struct WgradQueue {
    GemmAtbBatch gemm{};
    ReduceBatch red{};
    bool open = false;
};

// floats of partial sums a batched launch needs: the layout loop of wgrad_flush, for the slab count it would choose
size_t wgrad_batch_floats(const WgradQueue& q, int splits) {
    size_t off = 0;
    for (int e = 0; e < q.gemm.n; ++e) off += (size_t)splits * (q.gemm.e[e].Kp + 1) * q.gemm.e[e].Nw;
    return off;
}

int join_side(nerf_ctx* c, TrainState* t) {       // the main stream waits for the side stream's weight gradients
    if (t->side_pending) {
        HIP_OK(hipStreamWaitEvent(c->stream, t->ev_join, 0));
        t->side_pending = false;
    }
    return 0;
}

int wgrad_flush(nerf_ctx* c, TrainState* t, WgradQueue& q, long long Mp, bool on_side = false) {
    q.open = false;
    if (q.gemm.n == 0) return 0;
    hipStream_t st = c->stream;
    DevBuf* pb = &t->partial;
    if (on_side) {
        if (!t->side) {
            HIP_OK(hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming));
        }
        if (int r = join_side(c, t)) return r;         // (one launch in flight on the side stream at a time)
        HIP_OK(hipEventRecord(t->ev_fork, c->stream));  // everything the GEMM reads has been enqueued on the main stream
        HIP_OK(hipStreamWaitEvent(t->side, t->ev_fork, 0));
        st = t->side;
        pb = &t->partial_side;
    }
    int units = 0;
    for (int e = 0; e < q.gemm.n; ++e) units += (q.gemm.e[e].Kp + 255) / 256 * ((q.gemm.e[e].Nw + 255) / 256);
    // one 512-thread workgroup per CU, workgroups dealt round-robin to the 8 XCDs: a multiple of 8 slabs per layer such
    // that no XCD gets more than two rounds of its 32 CUs (10 units x 51 slabs put 70 workgroups on three of the XCDs:
    // a third round, +47 % on the launch)
    const int per_xcd = std::max(1, 2 * (c->num_cus / 8) / units);
    const int want_splits = 8 * per_xcd;
    long long rps = (Mp + want_splits - 1) / want_splits;
    rps = (rps + 31) / 32 * 32;
    const int splits = (int)((Mp + rps - 1) / rps);
    // the slab count follows the device's CU count: grow the partial-sum buffer to what THIS launch lays out (a device with
    // more CUs, a larger batch of layers) instead of trusting the size the per-layer launches were given
    if (int r = ensure(c, *pb, wgrad_batch_floats(q, splits) * sizeof(float))) { q.gemm.n = q.red.n = 0; return r; }
    size_t off = 0;
    for (int e = 0; e < q.gemm.n; ++e) {
        GemmAtb& g = q.gemm.e[e];
        g.rows_per_split = (int)rps;
        g.partial = (float*)pb->p + off;
        q.red.e[e].partial = g.partial;
        q.red.e[e].splits = splits;
        off += (size_t)splits * (g.Kp + 1) * g.Nw;
    }
    if (t->mixed) launch_gemm_atb_f16_batch(q.gemm, st, true);
    else if (q.gemm.e[0].g_rs) launch_gemm_atb_p_batch(q.gemm, st, true);     // (one format per trainer: all entries agree)
    else launch_gemm_atb_h_batch(q.gemm, st, true);
    launch_reduce_grad_batch(q.red, st);
    q.gemm.n = q.red.n = 0;
    if (on_side) {
        HIP_OK(hipEventRecord(t->ev_join, t->side));
        t->side_pending = true;
    }
    return 0;
}

// -> true if the GEMM also produced the weight gradient of head `sig_layer` (else the caller launches that head by itself)
bool wgrad(nerf_ctx* c, TrainState* t, TNet& n, int l, const float* A, int lda, const float* G, int ldg, int Ncols,
           int n_src_off, long long Mp, const unsigned* gmax = nullptr, WgradQueue* q = nullptr,
           const uint16_t* g_rs = nullptr /* pair16: G's row factors */,
           int sig_layer = -1 /* this GEMM also produces the weight gradient of head `sig_layer` (input = A) from t->dsig */) {
    const TLayer& L = n.L[l];
    GemmAtb g{};
    g.A = A; g.lda = lda; g.K = L.Kp; g.G = G; g.ldg = ldg; g.N = Ncols;
    g.partial = (float*)t->partial.p; g.Kp = L.Kp; g.Nw = Ncols; g.M = Mp;
    // the heads' (K x 4) results come from a VALU kernel that wants many small slabs; the GEMMs use kTrainSplits
    const bool f16 = (t->wgrad_f16 && gmax && Ncols >= 128) || (t->mixed && Ncols >= 128);
    const bool wide = f16 && t->wgrad_wide && Ncols >= 256;
    g.a_f16 = t->mixed ? 1 : 0;
    g.frag = t->frag ? 1 : 0;
    g.g_rs = f16 && !t->mixed ? g_rs : nullptr;
    const int want_splits = Ncols == 4 ? 1024 : wide ? kTrainSplitsWide : kTrainSplits;
    long long rps = (Mp + want_splits - 1) / want_splits;
    rps = t->frag ? (rps + 31) / 32 * 32 : (rps + 15) / 16 * 16;     // fragment-major operands: whole 32-row blocks
    g.rows_per_split = (int)rps;
    g.gmax = gmax;
    ReduceArgs r{};
    r.Kp = L.Kp; r.Nw = Ncols;
    r.grad_w = n.grad + L.w_off; r.grad_b = n.grad + L.b_off;
    r.K_real = L.K_real; r.N_real = L.N_real; r.n_src_off = n_src_off; r.rowmap = L.rowmap;
    r.accumulate = t->acc_grads ? 1 : 0;
    if (q && q->open && wide && reduce_grad_is_wide(r) && q->gemm.n < kWgradBatchMax) {
        q->gemm.e[q->gemm.n++] = g;               // slabs and partial regions are laid out by wgrad_flush
        q->red.e[q->red.n++] = r;
        return false;
    }
    const int n_splits = (int)((Mp + rps - 1) / rps);
    // the sigma head rides in this GEMM (gemm_atb_p / gemm_atb_f16, 128-wide tile): its slab sums land behind this GEMM's
    const bool sig = sig_layer >= 0 && t->frag && !wide && Ncols == 128 && (t->mixed || g.g_rs) && t->dsig.p;
    if (sig) {
        g.sig_g = (const float*)t->dsig.p;
        g.sig_partial = g.partial + (size_t)n_splits * (g.Kp + 1) * g.Nw;
    }
    if (Ncols == 4) launch_head_wgrad(g, c->stream);
    else if (t->mixed) launch_gemm_atb_f16(g, c->stream, wide);
    else if (g.g_rs) launch_gemm_atb_p(g, c->stream, wide);
    else if (f16) launch_gemm_atb_h(g, c->stream, wide);
    else launch_gemm_atb(g, c->stream);
    r.partial = g.partial; r.splits = n_splits;
    launch_reduce_grad(r, c->stream);
    if (sig) {
        const TLayer& Ls = n.L[sig_layer];
        ReduceArgs rs{};
        rs.partial = g.sig_partial; rs.Kp = g.Kp; rs.Nw = 1; rs.splits = n_splits;
        rs.grad_w = n.grad + Ls.w_off; rs.grad_b = n.grad + Ls.b_off;
        rs.K_real = Ls.K_real; rs.N_real = 1; rs.n_src_off = 0; rs.rowmap = Ls.rowmap;
        rs.accumulate = t->acc_grads ? 1 : 0;
        launch_reduce_grad(rs, c->stream);
    }
    return sig;
}

void dgrad(nerf_ctx* c, const float* G, int ldg, int Kg, const float* Wrows, int ldb, int Nout, const float* H, int ldh,
           float* Out, int ldo, long long Mp, unsigned* gmax_out, const float* r1a = nullptr, const float* r1b = nullptr,
           const TLayer* split = nullptr, const unsigned* gmax_in = nullptr) {
    GemmAbt g{};
    g.gmax = gmax_out;
    if (split && gmax_in && Nout % 128 == 0 && Kg % 32 == 0) {     // fp16 matrix cores: W comes pre-split (rows 0.. of W)
        g.A = G; g.lda = ldg; g.ldb = ldb; g.Out = Out; g.ldo = ldo;
        g.M = Mp; g.N = Nout; g.K = Kg; g.H = H; g.ldh = ldh; g.r1a = r1a; g.r1a_ld = 4; g.r1b = r1b;
        g.n_valid = Nout; g.alpha = c->cfg.leaky_relu_alpha;
        g.Bhi = split->Whi; g.Blo = split->Wlo; g.gmax_in = gmax_in;
        launch_gemm_abt_h(g, c->stream);
        return;
    }
    g.A = G; g.lda = ldg; g.Bt = Wrows; g.ldb = ldb; g.Out = Out; g.ldo = ldo;
    g.M = Mp; g.N = Nout; g.K = Kg; g.H = H; g.ldh = ldh; g.r1a = r1a; g.r1a_ld = 4; g.r1b = r1b;
    g.n_valid = Nout; g.alpha = c->cfg.leaky_relu_alpha;
    launch_gemm_abt(EPI_BWD_MASK, false, g, c->stream);
}

void dgrad_xyz(nerf_ctx* c, const float* G, const float* Wrows, float* dA0, long long Mp, bool accumulate) {
    GemmAbt g{};
    g.A = G; g.lda = 256; g.Bt = Wrows; g.ldb = 256; g.Out = dA0; g.ldo = kXyzPad;
    g.M = Mp; g.N = kXyzPad; g.K = 256; g.n_valid = kXyzPad; g.accumulate = accumulate ? 1 : 0;
    launch_gemm_abt(EPI_BWD_PLAIN, true, g, c->stream);
}

// Graw (Mp x 4, padding rows zero) must be filled; writes n.grad; with d_z != NULL adds dL/dz through the
// sample positions (d_z must already hold the compositing part).
int backward_pass(nerf_ctx* c, TrainState* t, int which, const PassDims& d, const float* o, const float* dirs,
                  float* d_z) {
    TNet& n = t->net[which];
    TPass& p = t->pass[which];
    const long long Mp = d.Mp;
    float *C4 = (float*)p.C4.p, *C8 = (float*)p.C8.p;
    float *H1 = (float*)p.H1.p, *H2 = (float*)p.H2.p, *H3 = (float*)p.H3.p, *H5 = (float*)p.H5.p,
          *H6 = (float*)p.H6.p, *H7 = (float*)p.H7.p, *H9 = (float*)p.H9.p;
    float *Ga = (float*)t->Ga.p, *Gb = (float*)t->Gb.p, *G9 = (float*)t->G9.p, *Graw = (float*)t->Graw.p,
          *dA0 = (float*)t->dA0.p;
    const bool dx = d_z != nullptr;
    // gm[k]: bits of max|G| of the gradient buffer produced k-th in this pass (scale of the split-fp16 weight gradient)
    // exact-fp32 weight gradients need no scale (but the fused chain always reports its maxima)
    unsigned* gm = t->wgrad_f16 || t->fused_backward ? (unsigned*)t->gmax.p + (size_t)which * 16 * 64 : nullptr;   // per pass
    // (the max|D| slots feed the split-fp16 weight gradients' scale; gemm_atb_f16 reads none: no reset under mixed_float16)
    if (gm && !t->mixed) HIP_OK(hipMemsetAsync(gm, 0, 16 * 64 * sizeof(unsigned), c->stream));
    auto GM = [&](int k) -> unsigned* { return gm ? gm + 64 * k : nullptr; };
    auto DS = [&](int l) -> const TLayer* { return t->dgrad_f16 && gm ? &n.L[l] : nullptr; };   // pre-split W of layer l
    if (t->frag && !(n.bstream && n.fcst && p.masks.p && gm))
        return fail("internal: fused training backward without its stream / mask records");
    if (t->fused_backward && n.bstream && n.fcst && p.masks.p && gm) {
        // ONE kernel for the whole data-gradient chain (mlp_bwd_f16x3.hip): the gradient stays on the lane from layer to
        // layer; every D_l is written once for the weight-gradient GEMMs below, with max|D_l| in the same gmax slots
        MlpBwdArgs b{};
        b.wstream = n.bstream; b.wconst = n.fcst; b.graw = Graw; b.gmax = gm; b.Mp = Mp; b.alpha = c->cfg.leaky_relu_alpha;
        b.ld = t->ldh; b.ld9 = t->ldh9;
        const int ldh = t->ldh, ldh9 = t->ldh9;
        const bool xyz = n.n_layers == 12;
        const int nrec = xyz ? 10 : 9;                   // mask records / gradient buffers; gmax group k <-> d_ptr[nrec - 1 - k]
        for (int l = 0; l < nrec; ++l) {
            b.mask_ptr[l] = (const uint32_t*)p.masks.p + (size_t)l * Mp * 8;
            b.d_ptr[l] = (float*)p.D[l].p;
            b.rs_ptr[l] = t->pair16 ? (uint16_t*)p.rs.p + (size_t)l * Mp : nullptr;
        }
        b.dx_ptr[0] = (float*)p.dxa.p; b.dx_ptr[1] = (float*)p.dxb.p;
        b.dsig = xyz ? nullptr : (float*)t->dsig.p;
        launch_mlp_bwd_f16x3(b, n.bdx, t->mixed, c->num_cus, c->stream, xyz);
        WgradQueue wq;
        auto RS = [&](int l) -> const uint16_t* { return t->pair16 ? b.rs_ptr[l] : nullptr; };   // row factors of d_ptr[l]
        if (xyz) {
            // get_network_only_xyz (src/NeRF.py:248-288): 10 = the rgb head on h9, 11 = the sigma head on h8 (the first 256
            // columns of C8), 9 = 256 -> 128 on the extra layer's output, 8 = that extra 256 -> 256 layer on h8
            float* H8b = (float*)p.H8b.p;
            wgrad(c, t, n, 10, H9, ldh9, Graw, 4, 4, 0, Mp);
            wgrad(c, t, n, 11, C8, kLdC8, Graw, 4, 4, 3, Mp);
            wgrad(c, t, n, 9, H8b, ldh, b.d_ptr[9], ldh9, 128, 0, Mp, GM(0), nullptr, RS(9));
            wq.open = t->wgrad_wide;
            wgrad(c, t, n, 8, C8, kLdC8, b.d_ptr[8], ldh, 256, 0, Mp, GM(1), &wq, RS(8));
        } else {
            wgrad(c, t, n, 9, H9, ldh9, Graw, 4, 4, 0, Mp);
            // the sigma head (layer 10: input C8 = [h8 | dir_enc], gradient column 3 of Graw) rides in layer 8's GEMM, which
            // stages C8 anyway -- no second pass over that buffer
            if (!wgrad(c, t, n, 8, C8, kLdC8, b.d_ptr[8], ldh9, 128, 0, Mp, GM(0), nullptr, RS(8), 10))
                wgrad(c, t, n, 10, C8, kLdC8, Graw, 4, 4, 3, Mp);      // (a build without the by-product path: -DNERF_PAIR16=0, float32 policy)
            wq.open = t->wgrad_wide;
        }
        const int g0 = xyz ? 1 : 0;                      // gmax group of D_l is g0 + 8 - l
        wgrad(c, t, n, 7, H7, ldh, b.d_ptr[7], ldh, 256, 0, Mp, GM(g0 + 1), &wq, RS(7));
        wgrad(c, t, n, 6, H6, ldh, b.d_ptr[6], ldh, 256, 0, Mp, GM(g0 + 2), &wq, RS(6));
        wgrad(c, t, n, 5, H5, ldh, b.d_ptr[5], ldh, 256, 0, Mp, GM(g0 + 3), &wq, RS(5));
        wgrad(c, t, n, 4, C4, kLdC4, b.d_ptr[4], ldh, 256, 0, Mp, GM(g0 + 4), &wq, RS(4));
        wgrad(c, t, n, 3, H3, ldh, b.d_ptr[3], ldh, 256, 0, Mp, GM(g0 + 5), &wq, RS(3));
        wgrad(c, t, n, 2, H2, ldh, b.d_ptr[2], ldh, 256, 0, Mp, GM(g0 + 6), &wq, RS(2));
        wgrad(c, t, n, 1, H1, ldh, b.d_ptr[1], ldh, 256, 0, Mp, GM(g0 + 7), &wq, RS(1));
        // (the xyz encoding's columns 256.. of C4: in the fragment-major buffer a column offset c is 32 c ELEMENTS --
        // half the byte offset under the fp16 policy)
        const size_t xyz_off = (size_t)32 * 256;
        const float* c4_xyz = t->mixed ? reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(C4) + xyz_off) : C4 + xyz_off;
        wgrad(c, t, n, 0, c4_xyz, kLdC4, b.d_ptr[0], ldh, 256, 0, Mp, GM(g0 + 8), &wq, RS(0));
        // the fine pass's batched launch can run beside the coarse pass's backward (which reads none of its operands)
        if (int r = wgrad_flush(c, t, wq, Mp, t->overlap && which == 1 && !t->acc_grads)) return r;
        if (dx) {
            if (!n.bdx) return fail("internal: the sampler term needs the backward stream with encoding tiles");
            launch_pe_bwd(b.dx_ptr[0], b.dx_ptr[1], o, dirs, (const float*)p.z.p, d.N, d.S, d_z, c->stream, true);
        }
        HIP_OK(hipGetLastError());
        return 0;
    }
    if (n.n_layers == 11) {
        wgrad(c, t, n, 9, H9, 128, Graw, 4, 4, 0, Mp);
        wgrad(c, t, n, 10, C8, kLdC8, Graw, 4, 4, 3, Mp);
        launch_head_bwd(Graw, n.L[9].W, H9, Mp, c->cfg.leaky_relu_alpha, G9, GM(0), c->stream);
        wgrad(c, t, n, 8, C8, kLdC8, G9, 128, 128, 0, Mp, GM(0));
        // dL/dh8 = G9 . W8[hidden rows]^T + Graw[:,3] * W10[hidden rows]   (WT10 row 0 = the sigma head's column)
        dgrad(c, G9, 128, 128, n.L[8].W, 128, 256, C8, kLdC8, Ga, 256, Mp, GM(1), Graw + 3, n.L[10].WT, DS(8), GM(0));
    } else {
        float* H8b = (float*)p.H8b.p;
        wgrad(c, t, n, 10, H9, 128, Graw, 4, 4, 0, Mp);
        wgrad(c, t, n, 11, C8, kLdC8, Graw, 4, 4, 3, Mp);
        launch_head_bwd(Graw, n.L[10].W, H9, Mp, c->cfg.leaky_relu_alpha, G9, GM(0), c->stream);
        wgrad(c, t, n, 9, H8b, 256, G9, 128, 128, 0, Mp, GM(0));
        dgrad(c, G9, 128, 128, n.L[9].W, 128, 256, H8b, 256, Gb, 256, Mp, GM(9), nullptr, nullptr, DS(9), GM(0));    // -> pre-activation grad of h8b
        wgrad(c, t, n, 8, C8, kLdC8, Gb, 256, 256, 0, Mp, GM(9));
        // dL/dh8 = Gb . W8^T + Graw[:,3] * W11   (WT11 row 0 = the sigma head's column)
        dgrad(c, Gb, 256, 256, n.L[8].W, 256, 256, C8, kLdC8, Ga, 256, Mp, GM(1), Graw + 3, n.L[11].WT, DS(8), GM(9));
    }
    wgrad(c, t, n, 7, H7, 256, Ga, 256, 256, 0, Mp, GM(1));
    dgrad(c, Ga, 256, 256, n.L[7].W, 256, 256, H7, 256, Gb, 256, Mp, GM(2), nullptr, nullptr, DS(7), GM(1));
    wgrad(c, t, n, 6, H6, 256, Gb, 256, 256, 0, Mp, GM(2));
    dgrad(c, Gb, 256, 256, n.L[6].W, 256, 256, H6, 256, Ga, 256, Mp, GM(3), nullptr, nullptr, DS(6), GM(2));
    wgrad(c, t, n, 5, H5, 256, Ga, 256, 256, 0, Mp, GM(3));
    dgrad(c, Ga, 256, 256, n.L[5].W, 256, 256, H5, 256, Gb, 256, Mp, GM(4), nullptr, nullptr, DS(5), GM(3));
    wgrad(c, t, n, 4, C4, kLdC4, Gb, 256, 256, 0, Mp, GM(4));
    dgrad(c, Gb, 256, 256, n.L[4].W, 256, 256, C4, kLdC4, Ga, 256, Mp, GM(5), nullptr, nullptr, DS(4), GM(4));
    if (dx) dgrad_xyz(c, Gb, n.L[4].W + (size_t)256 * 256, dA0, Mp, false);     // skip connection's xyz rows
    wgrad(c, t, n, 3, H3, 256, Ga, 256, 256, 0, Mp, GM(5));
    dgrad(c, Ga, 256, 256, n.L[3].W, 256, 256, H3, 256, Gb, 256, Mp, GM(6), nullptr, nullptr, DS(3), GM(5));
    wgrad(c, t, n, 2, H2, 256, Gb, 256, 256, 0, Mp, GM(6));
    dgrad(c, Gb, 256, 256, n.L[2].W, 256, 256, H2, 256, Ga, 256, Mp, GM(7), nullptr, nullptr, DS(2), GM(6));
    wgrad(c, t, n, 1, H1, 256, Ga, 256, 256, 0, Mp, GM(7));
    dgrad(c, Ga, 256, 256, n.L[1].W, 256, 256, H1, 256, Gb, 256, Mp, GM(8), nullptr, nullptr, DS(1), GM(7));
    wgrad(c, t, n, 0, C4 + 256, kLdC4, Gb, 256, 256, 0, Mp, GM(8));
    if (dx) {
        dgrad_xyz(c, Gb, n.L[0].W, dA0, Mp, true);
        launch_pe_bwd(dA0, nullptr, o, dirs, (const float*)p.z.p, d.N, d.S, d_z, c->stream);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

int stage_in(nerf_ctx* c, DevBuf& b, const float* src, size_t bytes, int mem, const float** out) {
    if (!src) { *out = nullptr; return 0; }
    if (mem == NERF_MEM_DEVICE) { *out = src; return 0; }
    if (int r = h2d(c, b, src, bytes)) return r;
    *out = (const float*)b.p;
    return 0;
}

int gradients_impl(nerf_ctx* c, const float* rays_o, const float* rays_d, const float* target, int64_t N, int Sc,
                   int Sf, const float* u_c, const float* u_f, uint64_t seed, int mem) {
    TrainState* t = c->train;
    if (!t || !t->training) return fail("nerf_train_begin has not been called");
    if (!rays_o || !rays_d || !target) return fail("NULL argument");
    if (N <= 0) return fail("need at least one ray (got %lld)", (long long)N);
    if (Sc < 1 || Sc > 1024) return fail("bad coarse sample count %d", Sc);
    const bool fine = Sf > 0 && t->net[1].present;
    if (fine && Sc < 2) return fail("hierarchical sampling needs at least 2 coarse samples (got %d)", Sc);
    if (fine && Sf > 256) return fail("training supports at most 256 fine samples per ray (got %d)", Sf);
    // same budget as the render path's sampler (nerf_api.hip): the backward sampler was validated up to 64 KiB of LDS
    if (fine && (sample_pdf_lds_bytes(Sc, Sf) > 64 * 1024 || sample_pdf_bwd_lds_bytes(Sc, Sf) > 64 * 1024))
        return fail("Sc=%d Sf=%d exceeds the sampler's LDS budget (forward %zu B, backward %zu B, limit 65536 B)", Sc, Sf,
                    sample_pdf_lds_bytes(Sc, Sf), sample_pdf_bwd_lds_bytes(Sc, Sf));
    const size_t f = sizeof(float);
    const float *o, *d, *tg, *uc, *uf;
    if (int r = stage_in(c, t->o, rays_o, N * 4 * f, mem, &o)) return r;
    if (int r = stage_in(c, t->d, rays_d, N * 4 * f, mem, &d)) return r;
    if (int r = stage_in(c, t->tgt, target, N * 3 * f, mem, &tg)) return r;
    if (int r = stage_in(c, t->u_c, u_c, (size_t)N * Sc * f, mem, &uc)) return r;
    if (int r = stage_in(c, t->u_f, fine ? u_f : nullptr, (size_t)N * (fine ? Sf : 0) * f, mem, &uf)) return r;

    PassDims dc{N, Sc, N * Sc, (N * Sc + 127) / 128 * 128};
    PassDims df{N, Sf, N * (long long)Sf, (N * (long long)Sf + 127) / 128 * 128};
    const long long Mmax = fine && df.Mp > dc.Mp ? df.Mp : dc.Mp;
    int r = ensure_pass(c, t->pass[0], dc);
    if (fine) r |= ensure_pass(c, t->pass[1], df);
    r |= ensure(c, t->Ga, Mmax * 256 * f);
    r |= ensure(c, t->Gb, Mmax * 256 * f);
    r |= ensure(c, t->G9, Mmax * 128 * f);
    r |= ensure(c, t->Graw, Mmax * 4 * f);
    r |= ensure(c, t->dsig, Mmax * f);
    r |= ensure(c, t->dA0, Mmax * kXyzPad * f);
    r |= ensure(c, t->partial, (size_t)2 * kTrainSplitsWide * (kLdC4 + 1) * 256 * f);   // also holds a pass's batched slabs
    r |= ensure(c, t->d_rgb, N * 3 * f);
    r |= ensure(c, t->d_wext, dc.M * f);
    r |= ensure(c, t->d_zf, (fine ? df.M : 1) * f);
    r |= ensure(c, t->scal, 4 * f);
    if (!t->macc.p) {
        r |= ensure(c, t->macc, 4 * sizeof(double));
        if (!r) HIP_OK(hipMemsetAsync(t->macc.p, 0, 4 * sizeof(double), c->stream));
    }
    r |= ensure(c, t->gmax, 2 * 16 * 64 * sizeof(unsigned));
    if (r) return r;
    float* scal = (float*)t->scal.p;
    float* Graw = (float*)t->Graw.p;
    float* d_rgb = (float*)t->d_rgb.p;
    // the finiteness flag collects over ONE gradient computation: gradients that were computed and never applied
    // (nerf_train_gradients without nerf_train_apply) must not decide the next step's verdict
    if (t->mixed) launch_opt_begin((OptState*)t->opt.p, c->stream);
    if (!fine && t->net[1].present)   // a skipped fine pass must not move the fine network
        HIP_OK(hipMemsetAsync(t->net[1].grad, 0, t->nblob * f, c->stream));

    // coarse forward (src/NeRF.py:146-151)
    TPass& pc = t->pass[0];
    launch_z_values(c->cfg.near_boundary, c->cfg.far_boundary, N, Sc, uc, seed, 0, (float*)pc.z.p, c->stream);
    if (int q = forward_pass(c, t, 0, dc, o, d)) return q;
    const bool through_sampler = fine && t->cfg.sampler_gradient != 0;
    if (fine) {
        // fine forward on the Sf new samples only (src/NeRF.py:155-157)
        TPass& pf = t->pass[1];
        launch_sample_pdf((const float*)pc.w.p, (const float*)pc.z.p, N, Sc, Sf, uf, seed, 0, (float*)pf.z.p, nullptr,
                          c->stream);
        if (int q = forward_pass(c, t, 1, df, o, d)) return q;
        launch_mse((const float*)pf.rgb.p, tg, N, (const OptState*)t->opt.p, t->loss_w[1], d_rgb, scal + 1, c->stream);
        HIP_OK(hipMemsetAsync(Graw + df.M * 4, 0, (df.Mp - df.M) * 4 * f, c->stream));
        float* d_zf = through_sampler ? (float*)t->d_zf.p : nullptr;
        launch_composite_bwd((const float*)pf.raw.p, (const float*)pf.z.p, (const float*)pf.T.p, N, Sf, d_rgb, nullptr,
                             Graw, d_zf, c->stream);
        if (int q = backward_pass(c, t, 1, df, o, d, d_zf)) return q;
        if (through_sampler)
            launch_sample_pdf_bwd((const float*)pc.w.p, (const float*)pc.z.p, N, Sc, Sf, uf, seed, 0, d_zf,
                                  (float*)t->d_wext.p, c->stream);
    }
    launch_mse((const float*)pc.rgb.p, tg, N, (const OptState*)t->opt.p, t->loss_w[0], d_rgb, scal + 0, c->stream);
    HIP_OK(hipMemsetAsync(Graw + dc.M * 4, 0, (dc.Mp - dc.M) * 4 * f, c->stream));
    launch_composite_bwd((const float*)pc.raw.p, (const float*)pc.z.p, (const float*)pc.T.p, N, Sc, d_rgb,
                         through_sampler ? (const float*)t->d_wext.p : nullptr, Graw, nullptr, c->stream);
    if (int q = backward_pass(c, t, 0, dc, o, d, nullptr)) return q;
    if (int q = join_side(c, t)) return q;
    if (t->mixed) {
        // LossScaleOptimizer: unscale, test for Inf/NaN; the verdict is taken on the device (opt_verdict_kernel) and gates
        // this step's Adam update.  (The per-sample scaling of the backward chain makes the products themselves
        // scale-invariant; the loss scale still guards the compositing / sampler backward and gives the reference's
        // skip-step behaviour.)
        launch_unscale_check(t->net[0].grad, fine ? t->net[1].grad : nullptr, t->nblob, (OptState*)t->opt.p, c->stream);
    }
    launch_metrics_accum(scal, fine, t->loss_w[0], t->loss_w[1], (double*)t->macc.p, c->stream);
    HIP_OK(hipGetLastError());
    return 0;
}

// The verdict of the gradients just computed, on the device: loss-scale bookkeeping (mixed_float16 policy: Keras 2.7
// LossScaleOptimizer -- halve on a non-finite step, double after `growth` finite ones) and the flag that lets this step's
// Adam launches through.  No host round trip: nerf_train_step keeps enqueuing.
// The verdict is taken on the blobs that are about to be APPLIED -- after a gradient all-reduce (the library's own, or the
// caller's between nerf_train_gradients and nerf_train_apply) those are the reduced blobs, on which every rank reaches
// the same verdict by itself: a non-finite shard makes the sum non-finite everywhere.
void take_verdict(nerf_ctx* c, bool recheck) {
    TrainState* t = c->train;
    if (t->mixed && recheck)
        launch_unscale_check(t->net[0].grad, t->net[1].present ? t->net[1].grad : nullptr, t->nblob, (OptState*)t->opt.p,
                             c->stream, true);
    launch_opt_verdict((OptState*)t->opt.p, c->stream);
}

// Backward through NeRF.render() itself (src/NeRF.py:109-134), the graph DietNeRF's consistency loss differentiates
// (src/DietNeRF.py:204-222): coarse pass -> inverse-CDF samples -> fine pass on sort(concat(z_new, z_coarse)) -> rgb.
// Given d_rgb = dL/d(render()[0]) it leaves dL/d(weights) of both networks in (or adds it to) the gradient blobs:
// the fine network through its Sc+Sf merged samples, the coarse network only through the sampler (its own rgb is
// not an output of render() when a fine network exists).
int render_gradients_impl(nerf_ctx* c, const float* rays_o, const float* rays_d, const float* d_rgb_in, int64_t N, int Sc,
                          int Sf, const float* u_c, const float* u_f, uint64_t seed, int64_t ray_base, bool accumulate,
                          int mem) {
    TrainState* t = c->train;
    if (!t || !t->training) return fail("nerf_train_begin has not been called");
    if (!rays_o || !rays_d || !d_rgb_in) return fail("NULL argument");
    if (N <= 0) return fail("need at least one ray (got %lld)", (long long)N);
    if (Sc < 1 || Sc > 1024) return fail("bad coarse sample count %d", Sc);
    const bool fine = Sf > 0 && t->net[1].present;
    if (fine && Sc < 2) return fail("hierarchical sampling needs at least 2 coarse samples (got %d)", Sc);
    if (fine && Sf > 256) return fail("training supports at most 256 fine samples per ray (got %d)", Sf);
    if (fine && (sample_pdf_lds_bytes(Sc, Sf) > 64 * 1024 || sample_pdf_bwd_lds_bytes(Sc, Sf) > 64 * 1024))
        return fail("Sc=%d Sf=%d exceeds the sampler's LDS budget", Sc, Sf);
    const size_t f = sizeof(float);
    const float *o, *d, *dr, *uc, *uf;
    if (int r = stage_in(c, t->o, rays_o, N * 4 * f, mem, &o)) return r;
    if (int r = stage_in(c, t->d, rays_d, N * 4 * f, mem, &d)) return r;
    if (int r = stage_in(c, t->tgt, d_rgb_in, N * 3 * f, mem, &dr)) return r;
    if (int r = stage_in(c, t->u_c, u_c, (size_t)N * Sc * f, mem, &uc)) return r;
    if (int r = stage_in(c, t->u_f, fine ? u_f : nullptr, (size_t)N * (fine ? Sf : 0) * f, mem, &uf)) return r;
    const int Sm = Sc + Sf;                                          // the fine pass renders the merged samples
    PassDims dc{N, Sc, N * Sc, (N * Sc + 127) / 128 * 128};
    PassDims df{N, Sm, N * (long long)Sm, (N * (long long)Sm + 127) / 128 * 128};
    const long long Mmax = fine ? df.Mp : dc.Mp;
    int r = ensure_pass(c, t->pass[0], dc);
    if (fine) r |= ensure_pass(c, t->pass[1], df);
    r |= ensure(c, t->Ga, Mmax * 256 * f);
    r |= ensure(c, t->Gb, Mmax * 256 * f);
    r |= ensure(c, t->G9, Mmax * 128 * f);
    r |= ensure(c, t->Graw, Mmax * 4 * f);
    r |= ensure(c, t->dsig, Mmax * f);
    r |= ensure(c, t->dA0, Mmax * kXyzPad * f);
    r |= ensure(c, t->partial, (size_t)2 * kTrainSplitsWide * (kLdC4 + 1) * 256 * f);   // also holds a pass's batched slabs
    r |= ensure(c, t->d_wext, dc.M * f);
    r |= ensure(c, t->d_zf, (fine ? N * (long long)Sf : 1) * f);
    r |= ensure(c, t->z_new, (fine ? N * (long long)Sf : 1) * f);
    r |= ensure(c, t->d_zm, (fine ? df.M : 1) * f);
    r |= ensure(c, t->zero_rgb, N * 3 * f);
    r |= ensure(c, t->gmax, 2 * 16 * 64 * sizeof(unsigned));
    if (t->mixed) {
        r |= ensure(c, t->d_rgb, N * 3 * f);
        if (accumulate)
            for (int w = 0; w < 2; ++w)
                if (t->net[w].present) r |= ensure(c, t->gsave[w], t->nblob * f);
    }
    if (r) return r;
    float* Graw = (float*)t->Graw.p;
    const bool through_sampler = fine && t->cfg.sampler_gradient != 0;
    // the networks this call computes gradients for: the fine one through its merged pass, the coarse one through the
    // sampler (or, without a fine network, through its own rgb)
    const bool computes[2] = {!fine || through_sampler, fine};
    if (t->mixed) {
        // mixed_float16 (src/ExecutionRun.py:220-221; DietNeRF scales the SUM of ray loss and consistency loss and unscales
        // once, src/DietNeRF.py:142-153,192-202): the caller's d_rgb is multiplied by the current loss scale on the device,
        // the single-pass chain runs on it (fp16 gradient buffers carrying the scale), and the result is UNSCALED and tested
        // before it is stored or, with accumulate, added to the unscaled gradients nerf_train_gradients left -- like with like.
        // The finiteness flag is reset only when this call starts a new gradient computation (accumulate = 0), so with
        // accumulate = 1 it collects over both calls and nerf_train_apply takes the one verdict.
        OptState* st = (OptState*)t->opt.p;
        if (!accumulate) launch_opt_begin(st, c->stream);
        launch_scale_by_loss_scale(dr, N * 3, st, (float*)t->d_rgb.p, c->stream);
        dr = (const float*)t->d_rgb.p;
        if (accumulate)
            for (int w = 0; w < 2; ++w)
                if (computes[w] && t->net[w].present)
                    HIP_OK(hipMemcpyAsync(t->gsave[w].p, t->net[w].grad, t->nblob * f, hipMemcpyDeviceToDevice, c->stream));
    }
    t->acc_grads = accumulate && !t->mixed;      // (mixed: the scaled result overwrites, the addition happens after unscaling)

    TPass& pc = t->pass[0];
    launch_z_values(c->cfg.near_boundary, c->cfg.far_boundary, N, Sc, uc, seed, ray_base, (float*)pc.z.p, c->stream);
    if (int q = forward_pass(c, t, 0, dc, o, d)) { t->acc_grads = false; return q; }
    int q = 0;
    if (fine) {
        TPass& pf = t->pass[1];
        launch_sample_pdf((const float*)pc.w.p, (const float*)pc.z.p, N, Sc, Sf, uf, seed, ray_base, (float*)t->z_new.p,
                          (float*)pf.z.p, c->stream);                // z_new (sorted) and the merged, sorted depths
        q = forward_pass(c, t, 1, df, o, d);
        if (!q) {
            HIP_OK(hipMemsetAsync(Graw + df.M * 4, 0, (df.Mp - df.M) * 4 * f, c->stream));
            float* d_zm = through_sampler ? (float*)t->d_zm.p : nullptr;
            launch_composite_bwd((const float*)pf.raw.p, (const float*)pf.z.p, (const float*)pf.T.p, N, Sm, dr, nullptr,
                                 Graw, d_zm, c->stream);
            q = backward_pass(c, t, 1, df, o, d, d_zm);
            if (!q && through_sampler) {
                launch_unmerge_grad((const float*)t->z_new.p, (const float*)pc.z.p, d_zm, N, Sc, Sf, (float*)t->d_zf.p,
                                    c->stream);
                launch_sample_pdf_bwd((const float*)pc.w.p, (const float*)pc.z.p, N, Sc, Sf, uf, seed, ray_base,
                                      (const float*)t->d_zf.p, (float*)t->d_wext.p, c->stream);
            }
        }
        if (!q) {
            if (through_sampler) {
                // the coarse network: no direct rgb term, only dL/d(weights_coarse) from the sampler
                HIP_OK(hipMemsetAsync(t->zero_rgb.p, 0, N * 3 * f, c->stream));
                HIP_OK(hipMemsetAsync(Graw + dc.M * 4, 0, (dc.Mp - dc.M) * 4 * f, c->stream));
                launch_composite_bwd((const float*)pc.raw.p, (const float*)pc.z.p, (const float*)pc.T.p, N, Sc,
                                     (const float*)t->zero_rgb.p, (const float*)t->d_wext.p, Graw, nullptr, c->stream);
                q = backward_pass(c, t, 0, dc, o, d, nullptr);
            } else if (!accumulate) {
                HIP_OK(hipMemsetAsync(t->net[0].grad, 0, t->nblob * f, c->stream));   // render() does not depend on it
            }
        }
    } else {
        HIP_OK(hipMemsetAsync(Graw + dc.M * 4, 0, (dc.Mp - dc.M) * 4 * f, c->stream));
        launch_composite_bwd((const float*)pc.raw.p, (const float*)pc.z.p, (const float*)pc.T.p, N, Sc, dr, nullptr, Graw,
                             nullptr, c->stream);
        q = backward_pass(c, t, 0, dc, o, d, nullptr);
    }
    t->acc_grads = false;
    if (q) return q;
    if (int qj = join_side(c, t)) return qj;
    if (t->mixed) {
        float* g[2]; const float* add[2]; int n = 0;
        for (int w = 0; w < 2; ++w)
            if (computes[w] && t->net[w].present) {
                g[n] = t->net[w].grad;
                add[n] = accumulate ? (const float*)t->gsave[w].p : nullptr;
                ++n;
            }
        launch_unscale_check(g[0], n > 1 ? g[1] : nullptr, t->nblob, (OptState*)t->opt.p, c->stream, false, add[0],
                             n > 1 ? add[1] : nullptr);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

// ---- the same graph in two calls, with the activations kept in between (RenderSlot) -------------------------------------
static void swap_grad_members(TPass& a, TPass& b) {
    for (int l = 0; l < 10; ++l) std::swap(a.D[l], b.D[l]);
    std::swap(a.dxa, b.dxa);
    std::swap(a.dxb, b.dxb);
    std::swap(a.rs, b.rs);
}
// the slot's stash becomes TrainState::pass (which forward_pass / backward_pass work on), keeping the working gradient buffers
static void slot_swap_in(TrainState* t, RenderSlot& s) {
    for (int w = 0; w < 2; ++w) {
        std::swap(t->pass[w], s.pass[w]);
        swap_grad_members(t->pass[w], s.pass[w]);
    }
    std::swap(t->z_new, s.z_new);
}
static void slot_swap_out(TrainState* t, RenderSlot& s) {
    for (int w = 0; w < 2; ++w) {
        swap_grad_members(t->pass[w], s.pass[w]);
        std::swap(t->pass[w], s.pass[w]);
    }
    std::swap(t->z_new, s.z_new);
}

int render_forward_impl(nerf_ctx* c, int slot, const float* rays_o, const float* rays_d, int64_t N, int Sc, int Sf,
                        const float* u_c, const float* u_f, uint64_t seed, int64_t ray_base, int mem) {
    TrainState* t = c->train;
    if (!t || !t->training) return fail("nerf_train_begin has not been called");
    if (!rays_o || !rays_d) return fail("NULL argument");
    if (slot < 0 || slot >= kMaxRenderSlots) return fail("slot %d out of range (0..%d)", slot, kMaxRenderSlots - 1);
    if (N <= 0) return fail("need at least one ray (got %lld)", (long long)N);
    if (Sc < 1 || Sc > 1024) return fail("bad coarse sample count %d", Sc);
    const bool fine = Sf > 0 && t->net[1].present;
    if (fine && Sc < 2) return fail("hierarchical sampling needs at least 2 coarse samples (got %d)", Sc);
    if (fine && Sf > 256) return fail("training supports at most 256 fine samples per ray (got %d)", Sf);
    if (fine && (sample_pdf_lds_bytes(Sc, Sf) > 64 * 1024 || sample_pdf_bwd_lds_bytes(Sc, Sf) > 64 * 1024))
        return fail("Sc=%d Sf=%d exceeds the sampler's LDS budget", Sc, Sf);
    if ((size_t)slot >= t->slots.size()) t->slots.resize((size_t)slot + 1);
    RenderSlot& s = t->slots[slot];
    s.valid = false;
    const size_t f = sizeof(float);
    // the slot keeps its own copies of the rays and draws: the caller's buffers need not outlive this call
    auto keep = [&](DevBuf& b, const float* src, size_t bytes) -> int {
        if (mem == NERF_MEM_HOST) return h2d(c, b, src, bytes);
        if (int r = ensure(c, b, bytes)) return r;
        HIP_OK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyDeviceToDevice, c->stream));
        return 0;
    };
    if (int r = keep(s.o, rays_o, N * 4 * f)) return r;
    if (int r = keep(s.d, rays_d, N * 4 * f)) return r;
    s.has_uc = u_c != nullptr;
    s.has_uf = fine && u_f != nullptr;
    if (s.has_uc) if (int r = keep(s.u_c, u_c, (size_t)N * Sc * f)) return r;
    if (s.has_uf) if (int r = keep(s.u_f, u_f, (size_t)N * Sf * f)) return r;
    const float *o = (const float*)s.o.p, *d = (const float*)s.d.p;
    const float* uc = s.has_uc ? (const float*)s.u_c.p : nullptr;
    const float* uf = s.has_uf ? (const float*)s.u_f.p : nullptr;
    const int Sm = Sc + Sf;
    PassDims dc{N, Sc, N * Sc, (N * Sc + 127) / 128 * 128};
    PassDims df{N, Sm, N * (long long)Sm, (N * (long long)Sm + 127) / 128 * 128};
    slot_swap_in(t, s);
    int r = ensure_pass(c, t->pass[0], dc);
    if (fine) r |= ensure_pass(c, t->pass[1], df);
    r |= ensure(c, t->z_new, (fine ? N * (long long)Sf : 1) * f);
    if (!r) {
        TPass& pc = t->pass[0];
        launch_z_values(c->cfg.near_boundary, c->cfg.far_boundary, N, Sc, uc, seed, ray_base, (float*)pc.z.p, c->stream);
        r = forward_pass(c, t, 0, dc, o, d);
        if (!r && fine) {
            launch_sample_pdf((const float*)pc.w.p, (const float*)pc.z.p, N, Sc, Sf, uf, seed, ray_base, (float*)t->z_new.p,
                              (float*)t->pass[1].z.p, c->stream);
            r = forward_pass(c, t, 1, df, o, d);
        }
    }
    slot_swap_out(t, s);
    if (r) return r;
    HIP_OK(hipGetLastError());
    s.N = N; s.Sc = Sc; s.Sf = fine ? Sf : 0; s.seed = seed; s.ray_base = ray_base;
    s.valid = true;
    return 0;
}

int render_backward_impl(nerf_ctx* c, int slot, const float* d_rgb_in, bool accumulate, int mem) {
    TrainState* t = c->train;
    if (!t || !t->training) return fail("nerf_train_begin has not been called");
    if (!d_rgb_in) return fail("NULL argument");
    if (slot < 0 || (size_t)slot >= t->slots.size() || !t->slots[slot].valid)
        return fail("slot %d holds no forward pass (nerf_train_render_forward first; a backward pass consumes it, and so does "
                    "an optimizer step)", slot);
    RenderSlot& s = t->slots[slot];
    const long long N = s.N;
    const int Sc = s.Sc, Sf = s.Sf;
    const bool fine = Sf > 0;
    const size_t f = sizeof(float);
    const float* dr;
    if (int r = stage_in(c, t->tgt, d_rgb_in, N * 3 * f, mem, &dr)) return r;
    const int Sm = Sc + Sf;
    PassDims dc{N, Sc, N * Sc, (N * Sc + 127) / 128 * 128};
    PassDims df{N, Sm, N * (long long)Sm, (N * (long long)Sm + 127) / 128 * 128};
    const long long Mmax = fine ? df.Mp : dc.Mp;
    int r = ensure(c, t->Ga, Mmax * 256 * f);
    r |= ensure(c, t->Gb, Mmax * 256 * f);
    r |= ensure(c, t->G9, Mmax * 128 * f);
    r |= ensure(c, t->Graw, Mmax * 4 * f);
    r |= ensure(c, t->dsig, Mmax * f);
    r |= ensure(c, t->dA0, Mmax * kXyzPad * f);
    r |= ensure(c, t->partial, (size_t)2 * kTrainSplitsWide * (kLdC4 + 1) * 256 * f);
    r |= ensure(c, t->d_wext, dc.M * f);
    r |= ensure(c, t->d_zf, (fine ? N * (long long)Sf : 1) * f);
    r |= ensure(c, t->d_zm, (fine ? df.M : 1) * f);
    r |= ensure(c, t->zero_rgb, N * 3 * f);
    r |= ensure(c, t->gmax, 2 * 16 * 64 * sizeof(unsigned));
    if (t->mixed) {
        r |= ensure(c, t->d_rgb, N * 3 * f);
        if (accumulate)
            for (int w = 0; w < 2; ++w)
                if (t->net[w].present) r |= ensure(c, t->gsave[w], t->nblob * f);
    }
    if (r) return r;
    float* Graw = (float*)t->Graw.p;
    const bool through_sampler = fine && t->cfg.sampler_gradient != 0;
    const bool computes[2] = {!fine || through_sampler, fine};
    if (t->mixed) {                               // as render_gradients_impl: scale d_rgb on the device, unscale at the end
        OptState* st = (OptState*)t->opt.p;
        if (!accumulate) launch_opt_begin(st, c->stream);
        launch_scale_by_loss_scale(dr, N * 3, st, (float*)t->d_rgb.p, c->stream);
        dr = (const float*)t->d_rgb.p;
        if (accumulate)
            for (int w = 0; w < 2; ++w)
                if (computes[w] && t->net[w].present)
                    HIP_OK(hipMemcpyAsync(t->gsave[w].p, t->net[w].grad, t->nblob * f, hipMemcpyDeviceToDevice, c->stream));
    }
    t->acc_grads = accumulate && !t->mixed;
    const float *o = (const float*)s.o.p, *d = (const float*)s.d.p;
    const float* uf = s.has_uf ? (const float*)s.u_f.p : nullptr;
    const uint64_t seed = s.seed;
    const long long ray_base = s.ray_base;
    slot_swap_in(t, s);
    TPass& pc = t->pass[0];
    int q = 0;
    if (fine) {
        TPass& pf = t->pass[1];
        HIP_OK(hipMemsetAsync(Graw + df.M * 4, 0, (df.Mp - df.M) * 4 * f, c->stream));
        float* d_zm = through_sampler ? (float*)t->d_zm.p : nullptr;
        launch_composite_bwd((const float*)pf.raw.p, (const float*)pf.z.p, (const float*)pf.T.p, N, Sm, dr, nullptr, Graw, d_zm,
                             c->stream);
        q = backward_pass(c, t, 1, df, o, d, d_zm);
        if (!q && through_sampler) {
            launch_unmerge_grad((const float*)t->z_new.p, (const float*)pc.z.p, d_zm, N, Sc, Sf, (float*)t->d_zf.p, c->stream);
            launch_sample_pdf_bwd((const float*)pc.w.p, (const float*)pc.z.p, N, Sc, Sf, uf, seed, ray_base,
                                  (const float*)t->d_zf.p, (float*)t->d_wext.p, c->stream);
            HIP_OK(hipMemsetAsync(t->zero_rgb.p, 0, N * 3 * f, c->stream));
            HIP_OK(hipMemsetAsync(Graw + dc.M * 4, 0, (dc.Mp - dc.M) * 4 * f, c->stream));
            launch_composite_bwd((const float*)pc.raw.p, (const float*)pc.z.p, (const float*)pc.T.p, N, Sc,
                                 (const float*)t->zero_rgb.p, (const float*)t->d_wext.p, Graw, nullptr, c->stream);
            q = backward_pass(c, t, 0, dc, o, d, nullptr);
        } else if (!q && !accumulate) {
            HIP_OK(hipMemsetAsync(t->net[0].grad, 0, t->nblob * f, c->stream));   // render() does not depend on it
        }
    } else {
        HIP_OK(hipMemsetAsync(Graw + dc.M * 4, 0, (dc.Mp - dc.M) * 4 * f, c->stream));
        launch_composite_bwd((const float*)pc.raw.p, (const float*)pc.z.p, (const float*)pc.T.p, N, Sc, dr, nullptr, Graw, nullptr,
                             c->stream);
        q = backward_pass(c, t, 0, dc, o, d, nullptr);
    }
    t->acc_grads = false;
    int qj = join_side(c, t);                     // (before the stash leaves TrainState::pass: the side stream reads it)
    slot_swap_out(t, s);
    s.valid = false;
    if (q) return q;
    if (qj) return qj;
    if (t->mixed) {
        float* g[2]; const float* add[2]; int n = 0;
        for (int w = 0; w < 2; ++w)
            if (computes[w] && t->net[w].present) {
                g[n] = t->net[w].grad;
                add[n] = accumulate ? (const float*)t->gsave[w].p : nullptr;
                ++n;
            }
        launch_unscale_check(g[0], n > 1 ? g[1] : nullptr, t->nblob, (OptState*)t->opt.p, c->stream, false, add[0],
                             n > 1 ? add[1] : nullptr);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

int read_metrics(nerf_ctx* c, bool fine, float* metrics) {
    if (!metrics) return 0;
    float h[2] = {0.f, 0.f};
    HIP_OK(hipMemcpyAsync(h, c->train->scal.p, 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    const float* lw = c->train->loss_w;                                // (1, 1: the products are exact)
    metrics[0] = fine ? lw[0] * h[0] + lw[1] * h[1] : lw[0] * h[0];    // src/NeRF.py:151,157
    metrics[1] = (float)(-10.0 * log10((double)h[0]));                  // get_psnr, UtilsNeuralRadianceField.py:123-132
    metrics[2] = fine ? (float)(-10.0 * log10((double)h[1])) : 0.f;
    return 0;
}

int apply_impl(nerf_ctx* c) {
    // Adam (gated on the device by the latest verdict: a dropped step leaves weights, moments and the iteration count
    // alone), then the training matrices are re-laid out from the blob either way
    TrainState* t = c->train;
    OptState* st = (OptState*)t->opt.p;
    for (RenderSlot& sl : t->slots) sl.valid = false;      // kept activations belong to the weights that made them
    for (int w = 0; w < 2; ++w) {
        TNet& n = t->net[w];
        if (!n.present) continue;
        launch_adam(n.blob, n.m, n.v, n.grad, t->nblob, t->cfg.learning_rate, t->cfg.beta_1, t->cfg.beta_2, t->cfg.epsilon,
                    st, c->stream);
        if (int r = relayout_net(c, n)) return r;
        n.render_dirty = true;
    }
    launch_opt_tick(st, t->cfg.beta_1, t->cfg.beta_2, c->stream);
    HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace

namespace nerf {

void train_free(nerf_ctx* c) {
    TrainState* t = c->train;
    if (!t) return;
    for (auto& n : t->net) {
        if (n.blob) (void)hipFree(n.blob);
        if (n.grad) (void)hipFree(n.grad);
        if (n.m) (void)hipFree(n.m);
        if (n.v) (void)hipFree(n.v);
        if (n.mats) (void)hipFree(n.mats);
        if (n.fstream) (void)hipFree(n.fstream);
        if (n.fcst) (void)hipFree(n.fcst);
        if (n.bstream) (void)hipFree(n.bstream);
    }
    for (int32_t* p : {t->rt_h, t->rt_h1, t->rt_ch, t->rt_f, t->rt_cf})
        if (p) (void)hipFree(p);
    if (t->sidx) (void)hipFree(t->sidx);
    if (t->cidx) (void)hipFree(t->cidx);
    for (int32_t* bi : t->bidx) if (bi) (void)hipFree(bi);
    std::vector<TPass*> passes = {&t->pass[0], &t->pass[1]};
    for (RenderSlot& sl : t->slots) {
        passes.push_back(&sl.pass[0]);
        passes.push_back(&sl.pass[1]);
        for (DevBuf* b : {&sl.o, &sl.d, &sl.u_c, &sl.u_f, &sl.z_new}) free_buf(*b);
    }
    for (TPass* pp : passes) {
        TPass& p = *pp;
        DevBuf* bs[] = {&p.C4, &p.C8, &p.H1, &p.H2, &p.H3, &p.H5, &p.H6, &p.H7, &p.H8b, &p.H9, &p.raw, &p.T, &p.w,
                        &p.rgb, &p.z};
        for (DevBuf* b : bs) free_buf(*b);
        free_buf(p.masks); free_buf(p.dxa); free_buf(p.dxb); free_buf(p.rs);
        for (DevBuf& b : p.D) free_buf(b);
    }
    if (t->side) { (void)hipStreamSynchronize(t->side); (void)hipStreamDestroy(t->side); }
    if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
    if (t->ev_join) (void)hipEventDestroy(t->ev_join);
    free_buf(t->partial_side);
    DevBuf* bs[] = {&t->Ga, &t->Gb, &t->G9, &t->Graw, &t->dsig, &t->dA0, &t->partial, &t->d_rgb, &t->d_wext, &t->d_zf, &t->tgt,
                    &t->o, &t->d, &t->u_c, &t->u_f, &t->scal, &t->gmax, &t->z_new, &t->d_zm, &t->zero_rgb, &t->opt,
                    &t->gsave[0], &t->gsave[1], &t->macc};
    for (DevBuf* b : bs) free_buf(*b);
    delete t;
    c->train = nullptr;
}

int train_on_load(nerf_ctx* c, int which) {
    TrainState* t = c->train;
    if (!t) return 0;
    TNet& n = t->net[which];
    if (!n.present) return init_net(c, t, which);
    HIP_OK(hipMemcpyAsync(n.blob, c->net[which].host_blob.data(), t->nblob * sizeof(float), hipMemcpyHostToDevice,
                          c->stream));
    n.render_dirty = n.host_stale = false;
    return relayout_net(c, n);
}

// gather tables of the render path's operand streams (see TrainState::rt_*)
static int ensure_render_tables(nerf_ctx* c, TrainState* t) {
    if (t->rt_h) return 0;
    const int na = c->cfg.n_angles;
    auto up = [&](const std::vector<int32_t>& v, int32_t** dst) -> int {
        HIP_OK(hipMalloc((void**)dst, v.size() * sizeof(int32_t)));
        HIP_OK(hipMemcpy(*dst, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        return 0;
    };
    std::vector<int32_t> h(f16_stream_bytes(na, false) / 2), h1(f16_stream_bytes(na, true) / 2), ch(kConstFloats),
        ch1(kConstFloats);
    build_f16x3_gather(na, false, h.data(), ch.data());
    build_f16x3_gather(na, true, h1.data(), ch1.data());
    if (ch != ch1) return fail("internal: the two fp16 streams disagree about their constants");
    // fp32 stream: pack a blob whose entry i holds i + 1 (exact in fp32: the blob has 5e5 entries) -- what lands in a slot
    // is the 1-based index of the weight that belongs there, 0 where the packer pads
    const size_t nf = (na == 0 ? kStreamBytesXyzF32 : kStreamBytes) / 4;
    std::vector<float> idx(t->nblob), sf(nf), cf(kConstFloats);
    for (size_t i = 0; i < t->nblob; ++i) idx[i] = (float)(i + 1);
    pack_weights_fp32(idx.data(), na, sf.data(), cf.data());
    std::vector<int32_t> f(nf), cfi(kConstFloats);
    for (size_t i = 0; i < nf; ++i) f[i] = (int32_t)sf[i];
    for (size_t i = 0; i < (size_t)kConstFloats; ++i) cfi[i] = (int32_t)cf[i];
    if (int r = up(h1, &t->rt_h1)) return r;
    if (int r = up(ch, &t->rt_ch)) return r;
    if (int r = up(f, &t->rt_f)) return r;
    if (int r = up(cfi, &t->rt_cf)) return r;
    return up(h, &t->rt_h);                  // last: rt_h != nullptr means all five exist
}

// The render path's view of a network that is being trained.  After optimizer steps its three operand streams are
// re-packed from the trained blob on the device (enqueued on the ctx stream: a render between steps does not synchronise);
// to_host additionally brings NetWeights::host_blob up to date (nerf_train_end, a restarting nerf_train_begin: the next
// trainer starts from it).
int train_flush_weights(nerf_ctx* c, int which, bool to_host) {
    TrainState* t = c->train;
    if (!t || !t->net[which].present) return 0;
    TNet& n = t->net[which];
    NetWeights& nw = c->net[which];
    if (n.render_dirty) {
        if (int r = ensure_render_tables(c, t)) return r;
        const int na = c->cfg.n_angles;
        launch_repack_f16x3(n.blob, t->rt_h, nw.stream_h, t->rt_ch, nw.cst_h, f16_stream_bytes(na, false), c->stream);
        launch_repack_f16x3(n.blob, t->rt_h1, nw.stream_h1, t->rt_ch, nw.cst_h, f16_stream_bytes(na, true), c->stream);
        launch_gather_blob(n.blob, t->rt_f, nw.stream, (na == 0 ? kStreamBytesXyzF32 : kStreamBytes) / 4, c->stream);
        launch_gather_blob(n.blob, t->rt_cf, nw.cst, kConstFloats, c->stream);
        HIP_OK(hipGetLastError());
        n.render_dirty = false;
        n.host_stale = true;
    }
    if (to_host && n.host_stale) {
        HIP_OK(hipMemcpyAsync(nw.host_blob.data(), n.blob, t->nblob * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        n.host_stale = false;
    }
    return 0;
}

}  // namespace nerf

extern "C" {

int nerf_train_begin(nerf_ctx* c, const nerf_train_config* cfg) {
    ENTER(c);
    if (!cfg) return fail("nerf_train_config is NULL");
    if (!(cfg->learning_rate > 0.f)) return fail("learning_rate must be positive");
    if (!c->net[0].loaded) return fail("load the coarse network's weights before nerf_train_begin");
    if (c->train && c->train->training) {
        // a trainer restarted without nerf_train_end goes on from the TRAINED weights ("the weights currently loaded")
        for (int w = 0; w < 2; ++w)
            if (int r = train_flush_weights(c, w, true)) return r;
        train_free(c);
    }
    TrainState* t = c->train;           // an inference-only state (xyz-only network) is promoted in place
    if (!t) {
        t = new TrainState();
        t->nblob = nerf_blob_size(&c->cfg);
        c->train = t;
    }
    t->cfg = *cfg;
    t->training = true;
    t->loss_w[0] = t->loss_w[1] = 1.f;
    t->mixed = cfg->mixed_float16 != 0;
    {
        OptState h{};
        h.scale = t->mixed ? (cfg->initial_loss_scale > 0.f ? cfg->initial_loss_scale : 32768.f) : 1.f;
        h.inv_scale = 1.0f / h.scale;
        h.adam_corr = (float)(sqrt(1.0 - (double)cfg->beta_2) / (1.0 - (double)cfg->beta_1));      // t = 1
        h.finite = 1; h.apply_ok = 1; h.good = 0;
        h.growth = cfg->dynamic_growth_steps > 0 ? cfg->dynamic_growth_steps : 2000;
        h.dynamic = t->mixed ? 1 : 0;
        h.iterations = 0; h.skipped = 0;
        if (int r = ensure(c, t->opt, sizeof(OptState))) { train_free(c); return r; }
        HIP_OK(hipMemcpy(t->opt.p, &h, sizeof(OptState), hipMemcpyHostToDevice));
    }
    // forward on the fused kernel unless the network has no fused kernel (xyz-only) or NERF_TRAIN_FORWARD=gemm asks
    // for the layer-wise fp32 GEMM forward (exact fp32 products instead of the 3-pass split)
    const char* fw = getenv("NERF_TRAIN_FORWARD");
    t->fused_forward = !(fw && strcmp(fw, "gemm") == 0);
    // weight gradients on the fp16 matrix cores (split operands, fp32-class) unless NERF_TRAIN_WGRAD=fp32
    const char* wg = getenv("NERF_TRAIN_WGRAD");
    t->wgrad_f16 = !(wg && strcmp(wg, "fp32") == 0);
    const char* ww = getenv("NERF_TRAIN_WGRAD_TILE");
    t->wgrad_wide = !(ww && strcmp(ww, "128") == 0);
    const char* dg = getenv("NERF_TRAIN_DGRAD");
    t->dgrad_f16 = t->wgrad_f16 && !(dg && strcmp(dg, "fp32") == 0);     // needs the max tracking of the f16 path
    // data gradients by the fused chain kernel (the stash forward's counterpart) unless NERF_TRAIN_BACKWARD=layers asks
    // for the layer-by-layer GEMMs; it reads the forward's mask records, so it needs the fused forward
    const char* bw = getenv("NERF_TRAIN_BACKWARD");
    t->fused_backward = t->fused_forward && t->dgrad_f16 && !(bw && strcmp(bw, "layers") == 0);
    // the fine pass's batched weight-gradient launch on a second stream (see TrainState::overlap) unless NERF_TRAIN_OVERLAP=0
    const char* ov = getenv("NERF_TRAIN_OVERLAP");
    t->overlap = !(ov && strcmp(ov, "0") == 0);
    t->ldh = 256; t->ldh9 = 128;
    // the fused forward writes fragment-major buffers that only the fused backward and the weight-gradient kernels read:
    // both fused or neither (NERF_TRAIN_BACKWARD=layers, NERF_TRAIN_WGRAD / NERF_TRAIN_DGRAD=fp32 select the layer-wise
    // GEMM forward as well)
    if (!t->fused_backward) t->fused_forward = false;
    t->frag = t->fused_forward && t->fused_backward;
    t->pair16 = kPair16 && t->frag && !t->mixed;
    if (t->mixed && !(t->fused_forward && t->fused_backward)) {
        train_free(c);
        return fail("mixed_float16 training runs on the fused forward / backward kernels: unset NERF_TRAIN_FORWARD / "
                    "NERF_TRAIN_BACKWARD / NERF_TRAIN_DGRAD / NERF_TRAIN_WGRAD");
    }
    for (int w = 0; w < 2; ++w) {
        if (!c->net[w].loaded) continue;
        if (!t->net[w].present) {
            if (int r = init_net(c, t, w)) { train_free(c); return r; }
        } else {
            if (int r = alloc_optimizer(c, t, t->net[w])) { train_free(c); return r; }
            if (int r = ensure_fused(c, t, t->net[w])) { train_free(c); return r; }
            if (int r = relayout_net(c, t->net[w])) { train_free(c); return r; }
        }
    }
    return 0;
}

int nerf_train_end(nerf_ctx* c) {
    ENTER(c);
    for (int w = 0; w < 2; ++w)
        if (int r = train_flush_weights(c, w, true)) return r;
    HIP_OK(hipStreamSynchronize(c->stream));
    train_free(c);
    return 0;
}

int nerf_train_loss_scale(nerf_ctx* c, float* loss_scale, int64_t* steps_applied, int64_t* steps_skipped) {
    if (!c || !c->train || !c->train->training) return fail("nerf_train_begin has not been called");
    OptState h;
    HIP_OK(hipMemcpyAsync(&h, c->train->opt.p, sizeof(OptState), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    if (loss_scale) *loss_scale = h.scale;
    if (steps_applied) *steps_applied = h.iterations;
    if (steps_skipped) *steps_skipped = h.skipped;
    return 0;
}

int nerf_train_read_metric_sums(nerf_ctx* c, double* sums, int64_t* steps) {
    ENTER(c);
    TrainState* t = c->train;
    if (!t || !t->training) return fail("nerf_train_begin has not been called");
    double h[4] = {0, 0, 0, 0};
    if (t->macc.p) {
        HIP_OK(hipMemcpyAsync(h, t->macc.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipMemsetAsync(t->macc.p, 0, sizeof h, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    if (sums) { sums[0] = h[0]; sums[1] = h[1]; sums[2] = h[2]; }
    if (steps) *steps = (int64_t)h[3];
    return 0;
}

int nerf_train_set_learning_rate(nerf_ctx* c, float lr) {
    if (!c || !c->train || !c->train->training) return fail("nerf_train_begin has not been called");
    if (!(lr > 0.f)) return fail("learning_rate must be positive");
    c->train->cfg.learning_rate = lr;
    return 0;
}

int nerf_train_set_loss_weights(nerf_ctx* c, float coarse_mse_weight, float fine_mse_weight) {
    if (!c || !c->train || !c->train->training) return fail("nerf_train_begin has not been called");
    if (!(coarse_mse_weight >= 0.f) || !(fine_mse_weight >= 0.f) || !(coarse_mse_weight <= 3.0e38f) ||
        !(fine_mse_weight <= 3.0e38f))
        return fail("loss weights must be finite and non-negative (got %g, %g)", coarse_mse_weight, fine_mse_weight);
    c->train->loss_w[0] = coarse_mse_weight;
    c->train->loss_w[1] = fine_mse_weight;
    return 0;
}

int nerf_train_gradients(nerf_ctx* c, const float* rays_orig, const float* rays_dirs, const float* target_rgb, int64_t N,
                         int32_t Sc, int32_t Sf, const float* u_coarse, const float* u_fine, uint64_t seed,
                         float* grad_coarse, float* grad_fine, float* metrics, int mem) {
    ENTER(c);
    if (int r = gradients_impl(c, rays_orig, rays_dirs, target_rgb, N, Sc, Sf, u_coarse, u_fine, seed, mem)) return r;
    TrainState* t = c->train;
    const bool fine = Sf > 0 && t->net[1].present;
    // (mixed_float16: the gradients come back unscaled; the skip-or-apply verdict and the loss-scale move belong to
    // nerf_train_apply, which tests the blobs it is given)
    const hipMemcpyKind kind = mem == NERF_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (grad_coarse) HIP_OK(hipMemcpyAsync(grad_coarse, t->net[0].grad, t->nblob * sizeof(float), kind, c->stream));
    if (grad_fine) {
        if (!fine) return fail("grad_fine requested but no fine pass ran (Sf = %d)", Sf);
        HIP_OK(hipMemcpyAsync(grad_fine, t->net[1].grad, t->nblob * sizeof(float), kind, c->stream));
    }
    if (mem == NERF_MEM_HOST) HIP_OK(hipStreamSynchronize(c->stream));
    return read_metrics(c, fine, metrics);
}

int nerf_train_render_gradients(nerf_ctx* c, const float* rays_orig, const float* rays_dirs, const float* d_rgb, int64_t N,
                                int32_t Sc, int32_t Sf, const float* u_coarse, const float* u_fine, uint64_t seed,
                                int64_t ray_base, int32_t accumulate, float* rgb_out, float* grad_coarse,
                                float* grad_fine, int mem) {
    ENTER(c);
    if (int r = render_gradients_impl(c, rays_orig, rays_dirs, d_rgb, N, Sc, Sf, u_coarse, u_fine, seed, ray_base,
                                      accumulate != 0, mem)) return r;
    TrainState* t = c->train;
    const bool fine = Sf > 0 && t->net[1].present;
    const hipMemcpyKind kind = mem == NERF_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (rgb_out) HIP_OK(hipMemcpyAsync(rgb_out, t->pass[fine ? 1 : 0].rgb.p, N * 3 * sizeof(float), kind, c->stream));
    if (grad_coarse) HIP_OK(hipMemcpyAsync(grad_coarse, t->net[0].grad, t->nblob * sizeof(float), kind, c->stream));
    if (grad_fine) {
        if (!fine) return fail("grad_fine requested but no fine pass ran (Sf = %d)", Sf);
        HIP_OK(hipMemcpyAsync(grad_fine, t->net[1].grad, t->nblob * sizeof(float), kind, c->stream));
    }
    if (mem == NERF_MEM_HOST) HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int nerf_train_render_forward(nerf_ctx* c, int32_t slot, const float* rays_orig, const float* rays_dirs, int64_t N, int32_t Sc,
                              int32_t Sf, const float* u_coarse, const float* u_fine, uint64_t seed, int64_t ray_base,
                              float* rgb_out, int mem) {
    ENTER(c);
    if (int r = render_forward_impl(c, slot, rays_orig, rays_dirs, N, Sc, Sf, u_coarse, u_fine, seed, ray_base, mem)) return r;
    const RenderSlot& s = c->train->slots[slot];
    if (rgb_out) {
        const hipMemcpyKind kind = mem == NERF_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        HIP_OK(hipMemcpyAsync(rgb_out, s.pass[s.Sf > 0 ? 1 : 0].rgb.p, N * 3 * sizeof(float), kind, c->stream));
    }
    if (mem == NERF_MEM_HOST) HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int nerf_train_render_backward(nerf_ctx* c, int32_t slot, const float* d_rgb, int32_t accumulate, float* grad_coarse,
                               float* grad_fine, int mem) {
    ENTER(c);
    const bool fine = c->train && slot >= 0 && (size_t)slot < c->train->slots.size() && c->train->slots[slot].Sf > 0;
    if (grad_fine && !fine) return fail("grad_fine requested but slot %d ran no fine pass", slot);   // (before it is consumed)
    if (int r = render_backward_impl(c, slot, d_rgb, accumulate != 0, mem)) return r;
    TrainState* t = c->train;
    const hipMemcpyKind kind = mem == NERF_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (grad_coarse) HIP_OK(hipMemcpyAsync(grad_coarse, t->net[0].grad, t->nblob * sizeof(float), kind, c->stream));
    if (grad_fine) HIP_OK(hipMemcpyAsync(grad_fine, t->net[1].grad, t->nblob * sizeof(float), kind, c->stream));
    if (mem == NERF_MEM_HOST) HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int nerf_train_render_release(nerf_ctx* c) {
    ENTER(c);
    TrainState* t = c->train;
    if (!t) return 0;
    HIP_OK(hipStreamSynchronize(c->stream));
    if (t->side) HIP_OK(hipStreamSynchronize(t->side));
    for (RenderSlot& sl : t->slots) {
        for (TPass& p : sl.pass) {
            DevBuf* bs[] = {&p.C4, &p.C8, &p.H1, &p.H2, &p.H3, &p.H5, &p.H6, &p.H7, &p.H8b, &p.H9, &p.raw, &p.T, &p.w,
                            &p.rgb, &p.z, &p.masks, &p.dxa, &p.dxb, &p.rs};
            for (DevBuf* b : bs) free_buf(*b);
            for (DevBuf& b : p.D) free_buf(b);
        }
        for (DevBuf* b : {&sl.o, &sl.d, &sl.u_c, &sl.u_f, &sl.z_new}) free_buf(*b);
    }
    t->slots.clear();
    (void)hipGetLastError();      // (a failed allocation while filling slots is what usually brings a caller here: start clean)
    return 0;
}

int nerf_train_apply(nerf_ctx* c, const float* grad_coarse, const float* grad_fine, int mem) {
    ENTER(c);
    TrainState* t = c->train;
    if (!t || !t->training) return fail("nerf_train_begin has not been called");
    const hipMemcpyKind kind = mem == NERF_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (grad_coarse) HIP_OK(hipMemcpyAsync(t->net[0].grad, grad_coarse, t->nblob * sizeof(float), kind, c->stream));
    if (grad_fine) {
        if (!t->net[1].present) return fail("grad_fine given but no fine network is loaded");
        HIP_OK(hipMemcpyAsync(t->net[1].grad, grad_fine, t->nblob * sizeof(float), kind, c->stream));
    }
    if (mem == NERF_MEM_HOST && (grad_coarse || grad_fine)) HIP_OK(hipStreamSynchronize(c->stream));
    // LossScaleOptimizer.apply_gradients: test what is about to be applied (the caller's all-reduced blobs, or the ctx's
    // own after nerf_train_gradients [+ nerf_train_render_gradients]); a non-finite step is skipped on the device
    take_verdict(c, true);
    return apply_impl(c);
}

int nerf_train_step(nerf_ctx* c, const float* rays_orig, const float* rays_dirs, const float* target_rgb, int64_t N,
                    int32_t Sc, int32_t Sf, const float* u_coarse, const float* u_fine, uint64_t seed, float* metrics,
                    int mem) {
    ENTER(c);
    if (int r = gradients_impl(c, rays_orig, rays_dirs, target_rgb, N, Sc, Sf, u_coarse, u_fine, seed, mem)) return r;
    const bool fine = Sf > 0 && c->train->net[1].present;
    // data-parallel: with a communicator (nerf_comm_init) every rank passes its shard of the batch and the gradient
    // blobs are averaged here, one all-reduce each, before the identical Adam update
    for (int w = 0; w < 2; ++w)
        if (c->train->net[w].present)
            if (int r = comm_allreduce_mean(c, c->train->net[w].grad, c->train->nblob)) return r;
    // a non-finite shard gradient makes the all-reduced blob non-finite on EVERY rank: the finiteness test is repeated on
    // the reduced blobs so that all ranks reach the same verdict (drop the step, halve the scale) by themselves
    // (repeated whenever the ctx has a communicator, one rank included: one small kernel)
    take_verdict(c, c->comm != nullptr);
    if (int r = apply_impl(c)) return r;                 // gated on the device by the verdict
    return read_metrics(c, fine, metrics);
}

int nerf_train_get_gradients(nerf_ctx* c, int which, float* blob, size_t n_floats, int mem) {
    ENTER(c);
    TrainState* t = c->train;
    if (!t || !t->training) return fail("nerf_train_begin has not been called");
    if (!blob) return fail("blob is NULL");
    if (which != NERF_NET_COARSE && which != NERF_NET_FINE) return fail("which must be 0 (coarse) or 1 (fine)");
    if (!t->net[which].present) return fail("network %d has no weights loaded", which);
    if (n_floats != t->nblob) return fail("gradient blob has %zu floats, expected %zu", n_floats, t->nblob);
    const hipMemcpyKind kind = mem == NERF_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    HIP_OK(hipMemcpyAsync(blob, t->net[which].grad, t->nblob * sizeof(float), kind, c->stream));
    if (mem == NERF_MEM_HOST) HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int nerf_get_weights(nerf_ctx* c, int which, float* blob, size_t n_floats, int mem) {
    ENTER(c);
    if (!blob) return fail("blob is NULL");
    if (which != NERF_NET_COARSE && which != NERF_NET_FINE) return fail("which must be 0 (coarse) or 1 (fine)");
    if (!c->net[which].loaded) return fail("network %d has no weights loaded", which);
    const size_t want = nerf_blob_size(&c->cfg);
    if (n_floats != want) return fail("weight blob has %zu floats, expected %zu", n_floats, want);
    TrainState* t = c->train;
    if (t && t->net[which].present) {
        const hipMemcpyKind kind = mem == NERF_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        HIP_OK(hipMemcpyAsync(blob, t->net[which].blob, want * sizeof(float), kind, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        return 0;
    }
    const hipMemcpyKind kind = mem == NERF_MEM_DEVICE ? hipMemcpyHostToDevice : hipMemcpyHostToHost;
    HIP_OK(hipMemcpy(blob, c->net[which].host_blob.data(), want * sizeof(float), kind));
    return 0;
}

}  // extern "C"

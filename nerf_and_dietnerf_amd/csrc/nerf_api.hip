// nerf_api.hip -- the C ABI of libnerf_mi355.so (include/nerf_mi355.h): context, weight upload,
// scratch arena and the orchestration of the render path
//   NeRF.render        src/NeRF.py:109-134
//   NeRF.render_image  src/NeRF.py:190-246
//   render_rays        src/UtilsNeuralRadianceField.py:181-211
// on one HIP stream.  No CPU compute path exists here: without a gfx950 device every call fails.
#include "../../include/nerf_mi355.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "nerf_ctx.h"

using namespace nerf;

thread_local std::string g_err;

namespace nerf {

int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

int ensure(nerf_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return 0;
    // a grow may free memory a still-running kernel uses: drain the stream first
    HIP_OK(hipStreamSynchronize(c->stream));
    if (b.p) HIP_OK(hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    size_t want = bytes + bytes / 8;
    HIP_OK(hipMalloc(&b.p, want));
    b.cap = want;
    return 0;
}

int h2d(nerf_ctx* c, DevBuf& b, const void* src, size_t bytes) {
    if (int r = ensure(c, b, bytes)) return r;
    HIP_OK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}

int enter(nerf_ctx* c) {
    if (!c) return fail("ctx is NULL");
    HIP_OK(hipSetDevice(c->cfg.device));
    return 0;
}

int upload_packed_weights(nerf_ctx* c, int which, const float* blob) {
    HIP_OK(hipSetDevice(c->cfg.device));
    NetWeights& n = c->net[which];
    if (c->cfg.n_angles == 0) {
        // xyz-only network: all three arithmetic modes run on the fused kernels' xyz-only variants, each with its own
        // stream and constant layout
        std::vector<uint16_t> sx(kStreamBytesF16Xyz / 2), sx1(kStreamBytesF16HiXyz / 2);
        std::vector<float> cx(kConstFloats), sf(kStreamBytesXyzF32 / 4), cf(kConstFloats);
        pack_weights_f16x3(blob, 0, sx.data(), cx.data());
        pack_weights_f16(blob, 0, sx1.data(), cx.data());
        pack_weights_fp32(blob, 0, sf.data(), cf.data());
        if (!n.stream_h) HIP_OK(hipMalloc((void**)&n.stream_h, kStreamBytesF16Xyz));
        if (!n.stream_h1) HIP_OK(hipMalloc((void**)&n.stream_h1, kStreamBytesF16HiXyz));
        if (!n.cst_h) HIP_OK(hipMalloc((void**)&n.cst_h, kConstBytes));
        if (!n.stream) HIP_OK(hipMalloc((void**)&n.stream, kStreamBytesXyzF32));
        if (!n.cst) HIP_OK(hipMalloc((void**)&n.cst, kConstBytes));
        HIP_OK(hipStreamSynchronize(c->stream));
        HIP_OK(hipMemcpy(n.stream, sf.data(), kStreamBytesXyzF32, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(n.cst, cf.data(), kConstBytes, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(n.stream_h, sx.data(), kStreamBytesF16Xyz, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(n.stream_h1, sx1.data(), kStreamBytesF16HiXyz, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(n.cst_h, cx.data(), kConstBytes, hipMemcpyHostToDevice));
        const size_t nf0 = nerf_blob_size(&c->cfg);
        if (n.host_blob.data() != blob) n.host_blob.assign(blob, blob + nf0);
        n.loaded = true;
        return 0;
    }
    std::vector<float> st(kStreamBytes / 4), cs(kConstFloats);
    pack_weights_fp32(blob, c->cfg.n_angles, st.data(), cs.data());
    if (!n.stream) HIP_OK(hipMalloc((void**)&n.stream, kStreamBytes));
    if (!n.cst) HIP_OK(hipMalloc((void**)&n.cst, kConstBytes));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpy(n.stream, st.data(), kStreamBytes, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(n.cst, cs.data(), kConstBytes, hipMemcpyHostToDevice));
    // both operand formats are kept resident (2 x 2.1 MB per network) so precision can be switched per call
    std::vector<uint16_t> sth(kStreamBytesF16 / 2);
    std::vector<float> csh(kConstFloats);
    pack_weights_f16x3(blob, c->cfg.n_angles, sth.data(), csh.data());
    if (!n.stream_h) HIP_OK(hipMalloc((void**)&n.stream_h, kStreamBytesF16));
    if (!n.cst_h) HIP_OK(hipMalloc((void**)&n.cst_h, kConstBytes));
    HIP_OK(hipMemcpy(n.stream_h, sth.data(), kStreamBytesF16, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(n.cst_h, csh.data(), kConstBytes, hipMemcpyHostToDevice));
    std::vector<uint16_t> sth1(kStreamBytesF16Hi / 2);
    pack_weights_f16(blob, c->cfg.n_angles, sth1.data(), csh.data());       // same constants as the 3-pass stream
    if (!n.stream_h1) HIP_OK(hipMalloc((void**)&n.stream_h1, kStreamBytesF16Hi));
    HIP_OK(hipMemcpy(n.stream_h1, sth1.data(), kStreamBytesF16Hi, hipMemcpyHostToDevice));
    const size_t nf = nerf_blob_size(&c->cfg);
    if (n.host_blob.data() != blob) n.host_blob.assign(blob, blob + nf);
    n.loaded = true;
    return 0;
}

}  // namespace nerf

namespace {

int check_cfg(const nerf_config* cfg) {
    if (!cfg) return fail("nerf_config is NULL");
    if (cfg->n_angles != 2 && cfg->n_angles != 1 && cfg->n_angles != 0)
        return fail("n_angles_for_model should be 1 or 2.");   // message of src/UtilsCV.py:138 (0 = xyz-only network)
    if (cfg->n_pos_enc_xyz != kLx || cfg->n_pos_enc_dir != kLd || cfg->hidden_dim != kHidden ||
        cfg->last_hidden_dim != kLast)
        return fail("fused kernel is specialised for Lx=%d Ld=%d hidden=%d last=%d (got %d %d %d %d)", kLx, kLd,
                    kHidden, kLast, cfg->n_pos_enc_xyz, cfg->n_pos_enc_dir, cfg->hidden_dim, cfg->last_hidden_dim);
    if (cfg->precision != NERF_PRECISION_FP32 && cfg->precision != NERF_PRECISION_F16X3 &&
        cfg->precision != NERF_PRECISION_F16)
        return fail("unknown precision %d (NERF_PRECISION_FP32 = 0, NERF_PRECISION_F16X3 = 1, NERF_PRECISION_F16 = 2)",
                    cfg->precision);
    return 0;
}

// record the MLP launch between two events when timing is on
int run_mlp(nerf_ctx* c, int which, const float* in_a, const float* in_b, const float* z, float* raw, long long M,
            int S, int mode) {
    if (!c->net[which].loaded) return fail("network %d has no weights loaded", which);
    if (int r = train_flush_weights(c, which)) return r;   // re-pack the operand streams after optimizer steps
    const bool f16 = c->cfg.precision == NERF_PRECISION_F16X3 || c->cfg.precision == NERF_PRECISION_F16;
    MlpArgs a{};
    a.wstream = c->cfg.precision == NERF_PRECISION_F16 ? (const float*)c->net[which].stream_h1
                : f16 ? (const float*)c->net[which].stream_h : c->net[which].stream;
    a.wconst = f16 ? c->net[which].cst_h : c->net[which].cst;
    a.in_a = in_a; a.in_b = in_b; a.z = z; a.raw = raw; a.M = M; a.S = S; a.mode = mode;
    a.nonfinite = c->nonfinite;
    a.alpha = c->cfg.leaky_relu_alpha;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t x, y;
            HIP_OK(hipEventCreate(&x));
            HIP_OK(hipEventCreate(&y));
            c->ev_pool.emplace_back(x, y);
        }
        e0 = c->ev_pool[c->ev_used].first; e1 = c->ev_pool[c->ev_used].second;
        c->ev_used++;
        c->timed_rows += M;
        HIP_OK(hipEventRecord(e0, c->stream));
    }
    // single-pass mode: two sample tiles per wave (half the weight stream per row) unless NERF_F16_TILES=1 asks for the
    // one-tile kernel; the xyz-only network has the one-tile variant only
    static const bool one_tile = [] { const char* e = getenv("NERF_F16_TILES"); return e && e[0] == '1'; }();
    if (c->cfg.precision == NERF_PRECISION_F16 && c->cfg.n_angles != 0 && !one_tile) launch_mlp_f16_2t(a, c->num_cus, c->stream);
    else if (f16) launch_mlp_f16x3(a, c->num_cus, c->stream, c->cfg.precision == NERF_PRECISION_F16, c->cfg.n_angles == 0);
    else launch_mlp_fp32(a, c->num_cus, c->stream, c->cfg.n_angles == 0);
    if (c->timing) HIP_OK(hipEventRecord(e1, c->stream));
    HIP_OK(hipGetLastError());
    return 0;
}

struct OutSizes { size_t per_ray[7]; };
OutSizes out_sizes(int S) {
    OutSizes o;
    o.per_ray[0] = 3; o.per_ray[1] = S; o.per_ray[2] = S; o.per_ray[3] = S; o.per_ray[4] = 3 * (size_t)S;
    o.per_ray[5] = S; o.per_ray[6] = 1;
    return o;
}
float** out_ptrs(nerf_outputs& o, int i) {
    switch (i) {
        case 0: return &o.rgb; case 1: return &o.weights; case 2: return &o.cumprod; case 3: return &o.alpha;
        case 4: return &o.rgb_samples; case 5: return &o.z; default: return &o.depth;
    }
}

// For host-memory calls: device twins of the requested outputs.
int make_dev_outputs(nerf_ctx* c, const nerf_outputs* host, long long N, int S, nerf_outputs* dev) {
    const OutSizes sz = out_sizes(S);
    nerf_outputs h = host ? *host : nerf_outputs{};
    *dev = nerf_outputs{};
    for (int i = 0; i < 7; ++i) {
        if (!*out_ptrs(h, i)) continue;
        if (int r = ensure(c, c->b_out[i], sz.per_ray[i] * N * sizeof(float))) return r;
        *out_ptrs(*dev, i) = (float*)c->b_out[i].p;
    }
    return 0;
}
int copy_back_outputs(nerf_ctx* c, const nerf_outputs* host, const nerf_outputs* dev, long long N, int S) {
    const OutSizes sz = out_sizes(S);
    nerf_outputs h = host ? *host : nerf_outputs{};
    nerf_outputs d = *dev;
    for (int i = 0; i < 7; ++i) {
        float* hp = *out_ptrs(h, i);
        if (!hp) continue;
        HIP_OK(hipMemcpyAsync(hp, *out_ptrs(d, i), sz.per_ray[i] * N * sizeof(float), hipMemcpyDeviceToHost,
                              c->stream));
    }
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

// Host-memory render_image: the rows [off, off + n) of every requested output leave for the host on the ctx's COPY stream
// as soon as the batch that produced them is done (event), while the next batch computes on the ctx stream.  With
// page-locked destinations (nerf_host_alloc) the copies are DMA transfers that run beside the kernels; pageable
// destinations work too (the runtime stages them) but serialise.
int copy_back_batch(nerf_ctx* c, const nerf_outputs* host, const nerf_outputs* dev, long long off, long long n, int S,
                    size_t batch_index) {
    if (!c->copy_stream) HIP_OK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    constexpr size_t kRing = 64;
    if (c->copy_ev.size() < kRing) {
        hipEvent_t e;
        HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->copy_ev.push_back(e);
    }
    hipEvent_t ev = c->copy_ev[batch_index % c->copy_ev.size()];
    HIP_OK(hipEventRecord(ev, c->stream));
    HIP_OK(hipStreamWaitEvent(c->copy_stream, ev, 0));
    const OutSizes sz = out_sizes(S);
    nerf_outputs h = host ? *host : nerf_outputs{};
    nerf_outputs d = *dev;
    for (int i = 0; i < 7; ++i) {
        float* hp = *out_ptrs(h, i);
        if (!hp) continue;
        HIP_OK(hipMemcpyAsync(hp + sz.per_ray[i] * off, *out_ptrs(d, i) + sz.per_ray[i] * off,
                              sz.per_ray[i] * n * sizeof(float), hipMemcpyDeviceToHost, c->copy_stream));
    }
    return 0;
}

// render_rays on device pointers
int dev_render_rays(nerf_ctx* c, int which, const float* o, const float* d, const float* z, long long N, int S,
                    const nerf_outputs& outs) {
    if (int r = ensure(c, c->b_raw, (size_t)N * S * 4 * sizeof(float))) return r;
    float* raw = (float*)c->b_raw.p;
    if (int r = run_mlp(c, which, o, d, z, raw, N * S, S, 0)) return r;
    launch_composite(raw, z, N, S, outs.rgb, outs.weights, outs.cumprod, outs.alpha, outs.rgb_samples, outs.depth,
                     c->stream);
    if (outs.z && outs.z != z)
        HIP_OK(hipMemcpyAsync(outs.z, z, (size_t)N * S * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIP_OK(hipGetLastError());
    return 0;
}

// NeRF.render on device pointers (u_* may be NULL -> Philox)
int dev_render(nerf_ctx* c, const float* o, const float* d, long long N, int Sc, int Sf, const float* u_c,
               const float* u_f, uint64_t seed, long long ray_base, const nerf_outputs& outs) {
    const bool fine = Sf > 0 && c->net[NERF_NET_FINE].loaded;
    if (int r = ensure(c, c->b_zc, (size_t)N * Sc * sizeof(float))) return r;
    float* zc = (float*)c->b_zc.p;
    launch_z_values(c->cfg.near_boundary, c->cfg.far_boundary, N, Sc, u_c, seed, ray_base, zc, c->stream);
    if (!fine) return dev_render_rays(c, NERF_NET_COARSE, o, d, zc, N, Sc, outs);
    if (Sc < 2) return fail("hierarchical sampling needs at least 2 coarse samples (got %d)", Sc);
    if (int r = ensure(c, c->b_wc, (size_t)N * Sc * sizeof(float))) return r;
    nerf_outputs co{};
    co.weights = (float*)c->b_wc.p;
    if (int r = dev_render_rays(c, NERF_NET_COARSE, o, d, zc, N, Sc, co)) return r;
    const int St = Sc + Sf;
    if (int r = ensure(c, c->b_zf, (size_t)N * St * sizeof(float))) return r;
    float* zf = (float*)c->b_zf.p;
    launch_sample_pdf(co.weights, zc, N, Sc, Sf, u_f, seed, ray_base, nullptr, zf, c->stream);
    return dev_render_rays(c, NERF_NET_FINE, o, d, zf, N, St, outs);
}

}  // namespace

extern "C" {

int nerf_abi_version(void) { return NERF_ABI_VERSION; }
const char* nerf_last_error(void) { return g_err.c_str(); }

size_t nerf_blob_size(const nerf_config* cfg) {
    if (check_cfg(cfg)) return 0;
    if (cfg->n_angles == 0)   // get_network_only_xyz, src/NeRF.py:248-288: 12 Dense layers
        return 33 * 256 + 256 + 3 * (256 * 256 + 256) + 289 * 256 + 256 + 3 * (256 * 256 + 256) + (256 * 256 + 256) +
               256 * 128 + 128 + 128 * 3 + 3 + 256 + 1;
    const size_t kd = 256 + 8 * (size_t)(cfg->n_angles + 1);   // [hidden, dir_enc]: 280 or 272
    return 33 * 256 + 256 + 3 * (256 * 256 + 256) + 289 * 256 + 256 + 3 * (256 * 256 + 256) + kd * 128 + 128 +
           128 * 3 + 3 + kd + 1;
}

int nerf_ctx_create(const nerf_config* cfg, nerf_ctx** out) {
    if (!out) return fail("out is NULL");
    *out = nullptr;
    if (int r = check_cfg(cfg)) return r;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail("no HIP device available (%s): libnerf_mi355 has no CPU path", hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev) return fail("device %d out of range (%d devices)", cfg->device, ndev);
    HIP_OK(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("device %d is %s; this library is built for gfx950 (MI355X) only", cfg->device, prop.gcnArchName);
    nerf_ctx* c = new nerf_ctx();
    c->cfg = *cfg;
    c->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail("hipStreamCreate failed");
    }
    c->stream = c->own_stream;
    if (hipMalloc((void**)&c->nonfinite, sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(c->nonfinite, 0, sizeof(unsigned long long)) != hipSuccess) {
        delete c;
        return fail("hipMalloc of the status counter failed");
    }
    mlp_fp32_set_attributes();
    mlp_f16x3_set_attributes();
    mlp_bwd_f16x3_set_attributes();
    mlp_f16_2t_set_attributes();
    *out = c;
    return 0;
}

void nerf_ctx_destroy(nerf_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->b_orig, &c->b_dirs, &c->b_zc, &c->b_zf, &c->b_raw, &c->b_wc, &c->b_u0, &c->b_u1,
                      &c->b_in0, &c->b_in1, &c->b_in2};
    for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
    for (auto& b : c->b_out) if (b.p) (void)hipFree(b.p);
    for (auto& n : c->net) {
        if (n.stream) (void)hipFree(n.stream);
        if (n.cst) (void)hipFree(n.cst);
        if (n.stream_h) (void)hipFree(n.stream_h);
        if (n.stream_h1) (void)hipFree(n.stream_h1);
        if (n.cst_h) (void)hipFree(n.cst_h);
    }
    for (auto& ev : c->ev_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (hipEvent_t e : c->copy_ev) (void)hipEventDestroy(e);
    comm_free(c);
    train_free(c);
    if (c->nonfinite) (void)hipFree(c->nonfinite);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int nerf_ctx_synchronize(nerf_ctx* c) {
    if (!c) return fail("ctx is NULL");
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int nerf_ctx_set_stream(nerf_ctx* c, void* s) {
    if (!c) return fail("ctx is NULL");
    HIP_OK(hipStreamSynchronize(c->stream));
    c->stream = (s == NERF_STREAM_OWN) ? c->own_stream : (hipStream_t)s;
    return 0;
}

int nerf_ctx_set_bounds(nerf_ctx* c, float near_b, float far_b) {
    if (!c) return fail("ctx is NULL");
    c->cfg.near_boundary = near_b; c->cfg.far_boundary = far_b;
    return 0;
}

int nerf_ctx_set_precision(nerf_ctx* c, int precision) {
    if (!c) return fail("ctx is NULL");
    if (precision != NERF_PRECISION_FP32 && precision != NERF_PRECISION_F16X3 && precision != NERF_PRECISION_F16)
        return fail("unknown precision %d", precision);
    HIP_OK(hipStreamSynchronize(c->stream));
    c->cfg.precision = precision;
    return 0;
}

int nerf_load_weights(nerf_ctx* c, int which, const float* blob, size_t n_floats) {
    if (!c || !blob) return fail("ctx/blob is NULL");
    if (which != NERF_NET_COARSE && which != NERF_NET_FINE) return fail("which must be 0 (coarse) or 1 (fine)");
    const size_t want = nerf_blob_size(&c->cfg);
    if (n_floats != want) return fail("weight blob has %zu floats, expected %zu", n_floats, want);
    if (int r = upload_packed_weights(c, which, blob)) return r;
    return train_on_load(c, which);      // a running trainer restarts from the new weights
}

int nerf_get_rays_directions(nerf_ctx* c, const float* c2w, float fov, int32_t H, int32_t W, float* dirs, int mem) {
    ENTER(c);
    if (!c || !c2w || !dirs) return fail("NULL argument");
    if (H <= 0 || W <= 0) return fail("bad image size %dx%d", H, W);
    const long long N = (long long)H * W;
    float* d = dirs;
    if (mem == NERF_MEM_HOST) {
        if (int r = ensure(c, c->b_dirs, N * 16)) return r;
        d = (float*)c->b_dirs.p;
    }
    launch_raygen(c2w, fov, H, W, 0, N, nullptr, d, c->stream);
    HIP_OK(hipGetLastError());
    if (mem == NERF_MEM_HOST) {
        HIP_OK(hipMemcpyAsync(dirs, d, N * 16, hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int nerf_get_z_values(nerf_ctx* c, int64_t N, int32_t S, const float* u, uint64_t seed, int64_t ray_base, float* z,
                      int mem) {
    ENTER(c);
    if (!c || !z) return fail("NULL argument");
    if (N < 0 || S <= 0) return fail("bad shape N=%lld S=%d", (long long)N, S);
    const float* du = u;
    float* dz = z;
    if (mem == NERF_MEM_HOST) {
        if (u) { if (int r = h2d(c, c->b_u0, u, (size_t)N * S * 4)) return r; du = (const float*)c->b_u0.p; }
        if (int r = ensure(c, c->b_zc, (size_t)N * S * 4)) return r;
        dz = (float*)c->b_zc.p;
    }
    launch_z_values(c->cfg.near_boundary, c->cfg.far_boundary, N, S, du, seed, ray_base, dz, c->stream);
    HIP_OK(hipGetLastError());
    if (mem == NERF_MEM_HOST) {
        HIP_OK(hipMemcpyAsync(z, dz, (size_t)N * S * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int nerf_sample_pdf(nerf_ctx* c, const float* weights, const float* z, int64_t N, int32_t S, int32_t Sf,
                    const float* u, uint64_t seed, int64_t ray_base, float* z_new, float* z_merged, int mem) {
    ENTER(c);
    if (!c || !weights || !z) return fail("NULL argument");
    if (N < 0 || S < 2 || Sf <= 0) return fail("bad shape N=%lld S=%d Sf=%d (need S>=2, Sf>=1)", (long long)N, S, Sf);
    if (sample_pdf_lds_bytes(S, Sf) > 64 * 1024) return fail("S=%d Sf=%d exceeds the sampler's LDS budget", S, Sf);
    const float *dw = weights, *dz = z, *du = u;
    float *dn = z_new, *dm = z_merged;
    if (mem == NERF_MEM_HOST) {
        if (int r = h2d(c, c->b_in0, weights, (size_t)N * S * 4)) return r;
        if (int r = h2d(c, c->b_in1, z, (size_t)N * S * 4)) return r;
        dw = (const float*)c->b_in0.p; dz = (const float*)c->b_in1.p;
        if (u) { if (int r = h2d(c, c->b_u1, u, (size_t)N * Sf * 4)) return r; du = (const float*)c->b_u1.p; }
        if (z_new) { if (int r = ensure(c, c->b_in2, (size_t)N * Sf * 4)) return r; dn = (float*)c->b_in2.p; }
        if (z_merged) { if (int r = ensure(c, c->b_zf, (size_t)N * (S + Sf) * 4)) return r; dm = (float*)c->b_zf.p; }
    }
    launch_sample_pdf(dw, dz, N, S, Sf, du, seed, ray_base, dn, dm, c->stream);
    HIP_OK(hipGetLastError());
    if (mem == NERF_MEM_HOST) {
        if (z_new) HIP_OK(hipMemcpyAsync(z_new, dn, (size_t)N * Sf * 4, hipMemcpyDeviceToHost, c->stream));
        if (z_merged) HIP_OK(hipMemcpyAsync(z_merged, dm, (size_t)N * (S + Sf) * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int nerf_positional_encoding(nerf_ctx* c, const float* x, int64_t M, int32_t n_enc, int32_t passthrough, float* out,
                             int mem) {
    ENTER(c);
    if (!c || !x || !out) return fail("NULL argument");
    if (M < 0 || n_enc < 0 || n_enc > 16) return fail("bad shape M=%lld n_enc=%d", (long long)M, n_enc);
    const size_t per = 3 * ((passthrough ? 1 : 0) + 2 * (size_t)n_enc);
    const float* dx = x;
    float* dout = out;
    if (mem == NERF_MEM_HOST) {
        if (int r = h2d(c, c->b_in0, x, (size_t)M * 12)) return r;
        if (int r = ensure(c, c->b_in1, (size_t)M * per * 4)) return r;
        dx = (const float*)c->b_in0.p; dout = (float*)c->b_in1.p;
    }
    launch_posenc(dx, M, n_enc, passthrough, dout, c->stream);
    HIP_OK(hipGetLastError());
    if (mem == NERF_MEM_HOST) {
        HIP_OK(hipMemcpyAsync(out, dout, (size_t)M * per * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int nerf_model_predict(nerf_ctx* c, int which, const float* xyz, const float* view_dirs, int64_t M, float* raw,
                       int mem) {
    ENTER(c);
    // view_dirs may be NULL for the xyz-only network (model_predict(..., view_dirs=None), UtilsNRF.py:229-234)
    if (!c || !xyz || !raw || (!view_dirs && c->cfg.n_angles != 0)) return fail("NULL argument");
    if (which != 0 && which != 1) return fail("which must be 0 (coarse) or 1 (fine)");
    if (M < 0) return fail("bad M");
    const float *dx = xyz, *dv = view_dirs;
    float* dr = raw;
    if (mem == NERF_MEM_HOST) {
        if (int r = h2d(c, c->b_in0, xyz, (size_t)M * 12)) return r;
        if (view_dirs)
            if (int r = h2d(c, c->b_in1, view_dirs, (size_t)M * 12)) return r;
        if (int r = ensure(c, c->b_raw, (size_t)M * 16)) return r;
        dx = (const float*)c->b_in0.p; dv = view_dirs ? (const float*)c->b_in1.p : nullptr; dr = (float*)c->b_raw.p;
    }
    if (int r = run_mlp(c, which, dx, dv, nullptr, dr, M, 1, 1)) return r;
    if (mem == NERF_MEM_HOST) {
        HIP_OK(hipMemcpyAsync(raw, dr, (size_t)M * 16, hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int nerf_ray_marching(nerf_ctx* c, const float* raw, const float* z, int64_t N, int32_t S, const nerf_outputs* outs,
                      int mem) {
    ENTER(c);
    if (!c || !raw || !z || !outs) return fail("NULL argument");
    if (N < 0 || S <= 0) return fail("bad shape");
    if (mem == NERF_MEM_DEVICE) {
        launch_composite(raw, z, N, S, outs->rgb, outs->weights, outs->cumprod, outs->alpha, outs->rgb_samples,
                         outs->depth, c->stream);
        if (outs->z && outs->z != z)
            HIP_OK(hipMemcpyAsync(outs->z, z, (size_t)N * S * 4, hipMemcpyDeviceToDevice, c->stream));
        HIP_OK(hipGetLastError());
        return 0;
    }
    if (int r = h2d(c, c->b_raw, raw, (size_t)N * S * 16)) return r;
    if (int r = h2d(c, c->b_zc, z, (size_t)N * S * 4)) return r;
    nerf_outputs dev;
    if (int r = make_dev_outputs(c, outs, N, S, &dev)) return r;
    launch_composite((const float*)c->b_raw.p, (const float*)c->b_zc.p, N, S, dev.rgb, dev.weights, dev.cumprod,
                     dev.alpha, dev.rgb_samples, dev.depth, c->stream);
    if (dev.z) HIP_OK(hipMemcpyAsync(dev.z, c->b_zc.p, (size_t)N * S * 4, hipMemcpyDeviceToDevice, c->stream));
    HIP_OK(hipGetLastError());
    return copy_back_outputs(c, outs, &dev, N, S);
}

int nerf_render_rays(nerf_ctx* c, int which, const float* o, const float* d, const float* z, int64_t N, int32_t S,
                     const nerf_outputs* outs, int mem) {
    ENTER(c);
    if (!c || !o || !d || !z || !outs) return fail("NULL argument");
    if (which != 0 && which != 1) return fail("which must be 0 (coarse) or 1 (fine)");
    if (N < 0 || S <= 0) return fail("bad shape");
    if (mem == NERF_MEM_DEVICE) return dev_render_rays(c, which, o, d, z, N, S, *outs);
    if (int r = h2d(c, c->b_orig, o, (size_t)N * 16)) return r;
    if (int r = h2d(c, c->b_dirs, d, (size_t)N * 16)) return r;
    if (int r = h2d(c, c->b_zc, z, (size_t)N * S * 4)) return r;
    nerf_outputs dev;
    if (int r = make_dev_outputs(c, outs, N, S, &dev)) return r;
    if (int r = dev_render_rays(c, which, (const float*)c->b_orig.p, (const float*)c->b_dirs.p,
                                (const float*)c->b_zc.p, N, S, dev))
        return r;
    return copy_back_outputs(c, outs, &dev, N, S);
}

int nerf_render(nerf_ctx* c, const float* o, const float* d, int64_t N, int32_t Sc, int32_t Sf, const float* u_c,
                const float* u_f, uint64_t seed, int64_t ray_base, const nerf_outputs* outs, int mem) {
    ENTER(c);
    if (!c || !o || !d || !outs) return fail("NULL argument");
    if (N < 0 || Sc <= 0 || Sf < 0) return fail("bad shape N=%lld Sc=%d Sf=%d", (long long)N, Sc, Sf);
    const bool fine = Sf > 0 && c->net[NERF_NET_FINE].loaded;
    const int S = fine ? Sc + Sf : Sc;
    if (fine && sample_pdf_lds_bytes(Sc, Sf) > 64 * 1024) return fail("Sc=%d Sf=%d exceeds the sampler's LDS budget", Sc, Sf);
    if (mem == NERF_MEM_DEVICE) return dev_render(c, o, d, N, Sc, Sf, u_c, u_f, seed, ray_base, *outs);
    if (int r = h2d(c, c->b_orig, o, (size_t)N * 16)) return r;
    if (int r = h2d(c, c->b_dirs, d, (size_t)N * 16)) return r;
    const float *duc = nullptr, *duf = nullptr;
    if (u_c) { if (int r = h2d(c, c->b_u0, u_c, (size_t)N * Sc * 4)) return r; duc = (const float*)c->b_u0.p; }
    if (u_f && fine) { if (int r = h2d(c, c->b_u1, u_f, (size_t)N * Sf * 4)) return r; duf = (const float*)c->b_u1.p; }
    nerf_outputs dev;
    if (int r = make_dev_outputs(c, outs, N, S, &dev)) return r;
    if (int r = dev_render(c, (const float*)c->b_orig.p, (const float*)c->b_dirs.p, N, Sc, Sf, duc, duf, seed,
                           ray_base, dev))
        return r;
    return copy_back_outputs(c, outs, &dev, N, S);
}

int nerf_render_image(nerf_ctx* c, const float* c2w, float fov, int32_t H, int32_t W, int64_t ray_begin,
                      int64_t ray_count, int64_t batch, int32_t Sc, int32_t Sf, const float* u_c, const float* u_f,
                      uint64_t seed, const nerf_outputs* outs, int mem) {
    ENTER(c);
    if (!c || !c2w || !outs) return fail("NULL argument");
    if (H <= 0 || W <= 0 || Sc <= 0 || Sf < 0) return fail("bad shape");
    const long long total = (long long)H * W;
    if (ray_count <= 0) { ray_begin = 0; ray_count = total; }
    if (ray_begin < 0 || ray_begin + ray_count > total) return fail("ray slab [%lld,+%lld) outside %lld rays",
                                                                    (long long)ray_begin, (long long)ray_count, total);
    if (batch < 0) return fail("batch_size must be > 0");   // assert of src/UtilsNRF.py:25
    const bool fine = Sf > 0 && c->net[NERF_NET_FINE].loaded;
    const int S = fine ? Sc + Sf : Sc;
    if (fine && sample_pdf_lds_bytes(Sc, Sf) > 64 * 1024) return fail("Sc=%d Sf=%d exceeds the sampler's LDS budget", Sc, Sf);
    const long long N = ray_count;
    if (batch == 0) {
        batch = 1 << 18;
        // host destinations with per-sample outputs (up to 5.4 KB per ray): four batches per slab so that the device-to-host
        // copies of one batch run under the next batch's kernels (results do not depend on the batch)
        const bool per_sample = outs->weights || outs->cumprod || outs->alpha || outs->rgb_samples || outs->z;
        if (mem == NERF_MEM_HOST && per_sample && N >= 32768) batch = ((N + 3) / 4 + 127) / 128 * 128;
    }
    // rays of the slab (origins are the broadcast translation column, src/NeRF.py:209)
    if (int r = ensure(c, c->b_orig, (size_t)N * 16)) return r;
    if (int r = ensure(c, c->b_dirs, (size_t)N * 16)) return r;
    launch_raygen(c2w, fov, H, W, ray_begin, N, (float*)c->b_orig.p, (float*)c->b_dirs.p, c->stream);
    const float *duc = u_c, *duf = u_f;
    nerf_outputs dev = *outs;
    if (mem == NERF_MEM_HOST) {
        if (u_c) { if (int r = h2d(c, c->b_u0, u_c + ray_begin * Sc, (size_t)N * Sc * 4)) return r; duc = (const float*)c->b_u0.p; }
        if (u_f && fine) { if (int r = h2d(c, c->b_u1, u_f + ray_begin * Sf, (size_t)N * Sf * 4)) return r; duf = (const float*)c->b_u1.p; }
        if (int r = make_dev_outputs(c, outs, N, S, &dev)) return r;
    } else {
        if (u_c) duc = u_c + ray_begin * Sc;
        if (u_f) duf = u_f + ray_begin * Sf;
    }
    const OutSizes sz = out_sizes(S);
    size_t k = 0;
    for (long long off = 0; off < N; off += batch, ++k) {
        const long long n = std::min<long long>(batch, N - off);
        nerf_outputs part = dev;
        for (int i = 0; i < 7; ++i) {
            float** p = out_ptrs(part, i);
            if (*p) *p += sz.per_ray[i] * off;
        }
        if (int r = dev_render(c, (const float*)c->b_orig.p + off * 4, (const float*)c->b_dirs.p + off * 4, n, Sc, Sf,
                               duc ? duc + off * Sc : nullptr, (duf && fine) ? duf + off * Sf : nullptr, seed,
                               ray_begin + off, part)) {
            if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
            return r;
        }
        if (mem == NERF_MEM_HOST)
            if (int r = copy_back_batch(c, outs, &dev, off, n, S, k)) { (void)hipStreamSynchronize(c->copy_stream); return r; }
    }
    // the last copy waits for the last batch: an idle copy stream means the whole call is done
    if (mem == NERF_MEM_HOST && c->copy_stream) HIP_OK(hipStreamSynchronize(c->copy_stream));
    return 0;
}

// Page-locked host memory: destinations the device writes by DMA at link speed, beside running kernels.
int nerf_host_alloc(size_t bytes, void** out) {
    if (!out) return fail("out is NULL");
    *out = nullptr;
    if (bytes == 0) return fail("nerf_host_alloc of 0 bytes");
    HIP_OK(hipHostMalloc(out, bytes, hipHostMallocPortable));
    return 0;
}

int nerf_host_free(void* p) {
    if (!p) return 0;
    HIP_OK(hipHostFree(p));
    return 0;
}

int nerf_ctx_read_nonfinite(nerf_ctx* c, int64_t* rows) {
    if (!c || !rows) return fail("NULL argument");
    ENTER(c);
    HIP_OK(hipStreamSynchronize(c->stream));
    unsigned long long v = 0;
    HIP_OK(hipMemcpy(&v, c->nonfinite, sizeof v, hipMemcpyDeviceToHost));
    HIP_OK(hipMemset(c->nonfinite, 0, sizeof v));
    *rows = (int64_t)v;
    return 0;
}

int nerf_ctx_enable_timing(nerf_ctx* c, int on) {
    if (!c) return fail("ctx is NULL");
    HIP_OK(hipStreamSynchronize(c->stream));
    c->timing = on != 0;
    c->ev_used = 0;
    c->timed_rows = 0;
    return 0;
}

int nerf_ctx_read_timing(nerf_ctx* c, double* mlp_ms, int64_t* n_launches, int64_t* n_rows) {
    if (!c) return fail("ctx is NULL");
    HIP_OK(hipStreamSynchronize(c->stream));
    double tot = 0;
    for (size_t i = 0; i < c->ev_used; ++i) {
        float ms = 0;
        HIP_OK(hipEventElapsedTime(&ms, c->ev_pool[i].first, c->ev_pool[i].second));
        tot += ms;
    }
    if (mlp_ms) *mlp_ms = tot;
    if (n_launches) *n_launches = (int64_t)c->ev_used;
    if (n_rows) *n_rows = c->timed_rows;
    c->ev_used = 0;
    c->timed_rows = 0;
    return 0;
}

}  // extern "C"

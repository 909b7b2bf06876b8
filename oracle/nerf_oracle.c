/*
 * nerf_oracle.c -- plain-C CPU restatement of the reference render path.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call this.  Second, independent restatement next to
 * oracle/nerf_oracle.py (numpy); tests/test_oracle_c.py checks the two against each other and against
 * the committed golden vectors.  Pinning status: see the header of nerf_oracle.py ("pinned by the
 * reference's shipped artifacts; elementwise agreement with TensorFlow itself is parity unpinned").
 *
 * Scalar fp32, canonical left-to-right evaluation order, no FMA contraction (build with
 * -ffp-contract=off).  Citations are into /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define XYZ_DIM 33
#define DIR_DIM 24
#define HID 256
#define LAST 128

/* ---- Philox4x32-10 (the build's counter RNG; mirrors nerf_oracle.py:philox_uniform) ---------- */
static void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)c[0] * 0xD2511F53u, p1 = (uint64_t)c[2] * 0xCD9E8D57u;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

float oracle_philox_uniform(uint64_t seed, uint64_t ray, int sample, uint32_t stream) {
    uint32_t c[4] = {(uint32_t)ray, (uint32_t)(ray >> 32), (uint32_t)(sample >> 2), stream};
    philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t bits = (c[sample & 3] >> 9) | 0x3F800000u;
    float f;
    memcpy(&f, &bits, 4);
    return f - 1.0f;
}

/* ---- get_rays_directions, src/UtilsCV.py:467-499 ------------------------------------------------ */
void oracle_get_rays_directions(int H, int W, float fov, const float* c2w, float* dirs /* (H,W,4) */) {
    const float tan_half = (float)tan((double)(fov * 0.5f));   /* :488, one rounding (see .py) */
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            float x_ndc = ((float)j + 0.5f) / (float)W;         /* :477-483 */
            float y_ndc = ((float)i + 0.5f) / (float)H;
            float xs = 2.0f * x_ndc - 1.0f, ys = 1.0f - 2.0f * y_ndc;   /* :485-486 */
            float xc = xs * tan_half, yc = ys * tan_half;       /* :489-490 */
            for (int r = 0; r < 4; ++r) {                       /* :498 einsum('ij,...j') */
                float s = c2w[r * 4 + 0] * xc + c2w[r * 4 + 1] * yc;
                s = s + c2w[r * 4 + 2] * -1.0f;
                s = s + c2w[r * 4 + 3] * 0.0f;
                dirs[((size_t)i * W + j) * 4 + r] = s;
            }
        }
}

/* ---- get_z_values, src/UtilsCV.py:565-581 (u explicit) ----------------------------------------- */
void oracle_get_z_values(float z_start, float z_end, int64_t N, int S, const float* u, float* z) {
    const float delta = S > 1 ? (z_end - z_start) / (float)(S - 1) : 0.f;
    const float span = (float)((double)z_end - (double)z_start);
    for (int64_t r = 0; r < N; ++r)
        for (int s = 0; s < S; ++s) {
            float lin = z_start + delta * (float)s;             /* tf.linspace */
            if (s == 0) lin = z_start;
            if (s == S - 1 && S > 1) lin = z_end;
            z[r * S + s] = lin + (u[r * S + s] * span) / (float)S;   /* :580 */
        }
}

/* ---- get_z_vals_from_prob_dist_func, src/UtilsCV.py:502-539 (u explicit) ------------------------ */
static int cmp_float(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

void oracle_sample_pdf(const float* w, const float* z, int64_t N, int S, int Sf, const float* u, float* z_new) {
    float* cdf = (float*)malloc(sizeof(float) * S);
    for (int64_t r = 0; r < N; ++r) {
        const float *wr = w + r * S, *zr = z + r * S;
        float sum = 0.f;
        for (int s = 0; s < S; ++s) sum = sum + wr[s];
        const float den0 = sum + 1e-7f;                          /* :514 */
        float acc = 0.f;
        for (int s = 0; s < S; ++s) { acc = acc + wr[s] / den0; cdf[s] = acc; }   /* :515 */
        for (int k = 0; k < Sf; ++k) {
            const float uu = u[r * Sf + k];
            int lo = 0, hi = S;                                  /* :517 searchsorted, side='left' */
            while (lo < hi) { int mid = (lo + hi) >> 1; if (cdf[mid] < uu) lo = mid + 1; else hi = mid; }
            int b = lo - 1 > 0 ? lo - 1 : 0;                     /* :519 */
            int t = lo < S - 1 ? lo : S - 1;                     /* :520-522 */
            float c_lo = cdf[b], c_hi = cdf[t];                  /* :525 */
            int bz = b < S - 2 ? b : S - 2, tz = t < S - 2 ? t : S - 2;   /* :528 clip to [0, S-2] */
            float z_lo = 0.5f * (zr[bz + 1] + zr[bz]);           /* :527 */
            float z_hi = 0.5f * (zr[tz + 1] + zr[tz]);
            float den = c_hi - c_lo;                             /* :532 */
            if (den < 1e-5f) den = 1e-5f;                        /* :533 */
            float tt = (uu - c_lo) / den;                        /* :535 */
            z_new[r * Sf + k] = z_lo + tt * (z_hi - z_lo);       /* :536 */
        }
        qsort(z_new + r * Sf, Sf, sizeof(float), cmp_float);     /* :537 */
    }
    free(cdf);
}

/* ---- positional encodings, src/UtilsNeuralRadianceField.py:52-85 -------------------------------- */
static void posenc(const float* x3, int n_enc, int passthrough, float* out) {
    const float pi = 3.14159274101257324f;   /* fp32(pi) */
    for (int c = 0; c < 3; ++c) {
        if (passthrough) *out++ = x3[c];
        for (int k = 0; k < n_enc; ++k) {
            float th = ((float)(1 << k) * pi) * x3[c];           /* (2^k * pi) * x, fp32 */
            *out++ = sinf(th);
            *out++ = cosf(th);
        }
    }
}

void oracle_positional_encoding(const float* x, int64_t M, int n_enc, int passthrough, float* out) {
    const int per = 3 * ((passthrough ? 1 : 0) + 2 * n_enc);
    for (int64_t m = 0; m < M; ++m) posenc(x + m * 3, n_enc, passthrough, out + m * per);
}

/* ---- the network, src/NeRF.py:316-339; blob = Keras get_weights() order ------------------------- */
static const int kShapes[11][2] = {{33, 256}, {256, 256}, {256, 256}, {256, 256}, {289, 256}, {256, 256},
                                   {256, 256}, {256, 256}, {280, 128}, {128, 3}, {280, 1}};

static void dense(const float* x, int in, const float* k, const float* b, int out, float alpha, int act, float* y) {
    for (int o = 0; o < out; ++o) {
        float acc = 0.f;
        for (int i = 0; i < in; ++i) acc = acc + x[i] * k[(size_t)i * out + o];
        acc = acc + b[o];
        if (act) { float a = alpha * acc; acc = acc > a ? acc : a; }   /* LeakyReLU = max(x, alpha x) */
        y[o] = acc;
    }
}

void oracle_model_predict(const float* blob, const float* xyz, const float* view, int64_t M, float alpha, float* raw) {
    const float *K[11], *B[11];
    size_t off = 0;
    for (int i = 0; i < 11; ++i) {
        K[i] = blob + off; off += (size_t)kShapes[i][0] * kShapes[i][1];
        B[i] = blob + off; off += kShapes[i][1];
    }
    float xe[XYZ_DIM], de[DIR_DIM], h[HID], h2[HID], cat[XYZ_DIM + HID], cat2[HID + DIR_DIM], h8[LAST];
    for (int64_t m = 0; m < M; ++m) {
        posenc(xyz + m * 3, 5, 1, xe);
        posenc(view + m * 3, 4, 0, de);
        dense(xe, 33, K[0], B[0], 256, alpha, 1, h);                       /* :319 */
        for (int l = 1; l <= 3; ++l) { dense(h, 256, K[l], B[l], 256, alpha, 1, h2); memcpy(h, h2, sizeof h); }
        memcpy(cat, xe, sizeof xe); memcpy(cat + XYZ_DIM, h, sizeof h);    /* :324 [xyz, hidden] */
        dense(cat, 289, K[4], B[4], 256, alpha, 1, h2); memcpy(h, h2, sizeof h);
        for (int l = 5; l <= 7; ++l) { dense(h, 256, K[l], B[l], 256, alpha, 1, h2); memcpy(h, h2, sizeof h); }
        memcpy(cat2, h, sizeof h); memcpy(cat2 + HID, de, sizeof de);      /* :330 [hidden, dirs] */
        dense(cat2, 280, K[8], B[8], 128, alpha, 1, h8);                   /* :331 */
        dense(h8, 128, K[9], B[9], 3, alpha, 0, raw + m * 4);              /* :333 */
        dense(cat2, 280, K[10], B[10], 1, alpha, 0, raw + m * 4 + 3);      /* :336 */
    }
}

/* ---- ray_marching, src/UtilsNeuralRadianceField.py:88-115 (+ depth, src/ExecutionRun.py:346) ---- */
void oracle_ray_marching(const float* raw, const float* z, int64_t N, int S, float* rgb, float* weights,
                         float* cumprod, float* alpha_out, float* rgb_samples, float* depth) {
    for (int64_t r = 0; r < N; ++r) {
        float T = 1.0f, c[3] = {0.f, 0.f, 0.f}, dep = 0.f;
        for (int s = 0; s < S; ++s) {
            const float* o = raw + (r * S + s) * 4;
            float delta = s + 1 < S ? z[r * S + s + 1] - z[r * S + s] : 1e9f;   /* :104-106 */
            float sigma = o[3] > 0.f ? o[3] : 0.f;                               /* :100 */
            float a = 1.0f - expf(-(sigma * delta));                             /* :111 */
            float w = a * T;                                                     /* :113 */
            for (int k = 0; k < 3; ++k) {
                float col = 1.0f / (1.0f + expf(-o[k]));                         /* :101 */
                c[k] = c[k] + w * col;                                           /* :114 */
                if (rgb_samples) rgb_samples[(r * S + s) * 3 + k] = col;
            }
            dep = dep + w * z[r * S + s];
            if (weights) weights[r * S + s] = w;
            if (cumprod) cumprod[r * S + s] = T;
            if (alpha_out) alpha_out[r * S + s] = a;
            T = T * (1.0f - a);                                                  /* :112 exclusive */
        }
        if (rgb) { rgb[r * 3] = c[0]; rgb[r * 3 + 1] = c[1]; rgb[r * 3 + 2] = c[2]; }
        if (depth) depth[r] = dep;
    }
}

/* ---- NeRF.render, src/NeRF.py:109-134 (draws explicit) ------------------------------------------- */
static void render_rays(const float* blob, const float* o, const float* d, const float* z, int64_t N, int S,
                        float alpha, float* rgb, float* weights, float* T, float* a, float* cs) {
    float* pts = (float*)malloc(sizeof(float) * N * S * 3);
    float* view = (float*)malloc(sizeof(float) * N * S * 3);
    float* raw = (float*)malloc(sizeof(float) * N * S * 4);
    for (int64_t r = 0; r < N; ++r)
        for (int s = 0; s < S; ++s)
            for (int k = 0; k < 3; ++k) {
                pts[(r * S + s) * 3 + k] = o[r * 4 + k] + d[r * 4 + k] * z[r * S + s];   /* UtilsCV.py:598 */
                view[(r * S + s) * 3 + k] = d[r * 4 + k];                                /* UtilsCV.py:140-142 */
            }
    oracle_model_predict(blob, pts, view, N * S, alpha, raw);
    oracle_ray_marching(raw, z, N, S, rgb, weights, T, a, cs, NULL);
    free(pts); free(view); free(raw);
}

void oracle_render(const float* blob_c, const float* blob_f, const float* o, const float* d, int64_t N, float near_b,
                   float far_b, int Sc, int Sf, const float* u_c, const float* u_f, float alpha, float* rgb,
                   float* weights, float* cumprod, float* alpha_out, float* rgb_samples, float* z_out) {
    float* zc = (float*)malloc(sizeof(float) * N * Sc);
    oracle_get_z_values(near_b, far_b, N, Sc, u_c, zc);                         /* :127 */
    if (!blob_f || Sf <= 0) {
        render_rays(blob_c, o, d, zc, N, Sc, alpha, rgb, weights, cumprod, alpha_out, rgb_samples);
        memcpy(z_out, zc, sizeof(float) * N * Sc);
        free(zc);
        return;
    }
    float* wc = (float*)malloc(sizeof(float) * N * Sc);
    float* rgb_c = (float*)malloc(sizeof(float) * N * 3);
    render_rays(blob_c, o, d, zc, N, Sc, alpha, rgb_c, wc, NULL, NULL, NULL);   /* :128 */
    float* zn = (float*)malloc(sizeof(float) * N * Sf);
    oracle_sample_pdf(wc, zc, N, Sc, Sf, u_f, zn);                              /* :131 */
    const int St = Sc + Sf;
    for (int64_t r = 0; r < N; ++r) {                                           /* :132 sort(concat) */
        memcpy(z_out + r * St, zn + r * Sf, sizeof(float) * Sf);
        memcpy(z_out + r * St + Sf, zc + r * Sc, sizeof(float) * Sc);
        qsort(z_out + r * St, St, sizeof(float), cmp_float);
    }
    render_rays(blob_f, o, d, z_out, N, St, alpha, rgb, weights, cumprod, alpha_out, rgb_samples);   /* :133 */
    free(zc); free(wc); free(rgb_c); free(zn);
}

// aux_kernels.hip -- the non-MLP functions of the render path as HIP kernels (gfx950):
//   raygen        get_rays_directions            src/UtilsCV.py:467-499 (+ origins, src/NeRF.py:209)
//   z_values      get_z_values                   src/UtilsCV.py:565-581
//   sample_pdf    get_z_vals_from_prob_dist_func src/UtilsCV.py:502-539 (+ sort(concat), src/NeRF.py:132)
//   composite     ray_marching                   src/UtilsNeuralRadianceField.py:88-115 (+ depth, ExecutionRun.py:346)
//   posenc        positional_encoding_for_*      src/UtilsNeuralRadianceField.py:52-85 (standalone; tests/tools)
//
// All HBM-bound fp32/int32 work.  Evaluation order is the canonical one of oracle/nerf_oracle.py:
// sums / cumsum / cumprod run left to right along the sample axis and products are NOT contracted
// into FMAs (explicit __fmul_rn/__fadd_rn), so index selection in the sampler is bit-exact against
// the oracle and the other outputs differ only through expf.
#include "nerf_kernels.h"
#include "nerf_device.h"

namespace nerf {

// ------------------------------------------------------------------------------------------------
// raygen: one thread per ray of the slab [ray_begin, ray_begin + ray_count) of the H*W image.
// ------------------------------------------------------------------------------------------------
struct RaygenArgs {
    float c[16];
    float tan_half;
    int H, W;
    long long ray_begin, ray_count;
    float* orig;
    float* dirs;
};

__global__ void raygen_kernel(const RaygenArgs a) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.ray_count) return;
    const long long ray = a.ray_begin + t;
    const int i = (int)(ray / a.W), j = (int)(ray % a.W);
    // pixel centre -> NDC -> screen space (same tan for both axes, no aspect term)
    const float x_ndc = __fdiv_rn((float)j + 0.5f, (float)a.W);
    const float y_ndc = __fdiv_rn((float)i + 0.5f, (float)a.H);
    const float xs = __fsub_rn(__fmul_rn(2.0f, x_ndc), 1.0f);
    const float ys = __fsub_rn(1.0f, __fmul_rn(2.0f, y_ndc));
    const float xc = __fmul_rn(xs, a.tan_half);
    const float yc = __fmul_rn(ys, a.tan_half);
    float d[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // einsum('ij,...j') with v = (xc, yc, -1, 0): ((c0*x + c1*y) + c2*z) + c3*0
        float s = __fadd_rn(__fmul_rn(a.c[r * 4 + 0], xc), __fmul_rn(a.c[r * 4 + 1], yc));
        s = __fadd_rn(s, __fmul_rn(a.c[r * 4 + 2], -1.0f));
        s = __fadd_rn(s, __fmul_rn(a.c[r * 4 + 3], 0.0f));
        d[r] = s;
    }
    reinterpret_cast<float4*>(a.dirs)[t] = make_float4(d[0], d[1], d[2], d[3]);
    if (a.orig) reinterpret_cast<float4*>(a.orig)[t] = make_float4(a.c[3], a.c[7], a.c[11], a.c[15]);
}

void launch_raygen(const float c2w_host[16], float fov, int H, int W, long long ray_begin,
                   long long ray_count, float* orig, float* dirs, hipStream_t stream) {
    if (ray_count <= 0) return;
    RaygenArgs a;
    for (int i = 0; i < 16; ++i) a.c[i] = c2w_host[i];
    // tf.tan(field_of_view / 2) (UtilsCV.py:488): tan of fp32(fov/2), evaluated in double and rounded once
    a.tan_half = (float)tan((double)(fov * 0.5f));
    a.H = H; a.W = W; a.ray_begin = ray_begin; a.ray_count = ray_count; a.orig = orig; a.dirs = dirs;
    const int bs = 256;
    hipLaunchKernelGGL(raygen_kernel, dim3((unsigned)((ray_count + bs - 1) / bs)), dim3(bs), 0, stream, a);
}

// ------------------------------------------------------------------------------------------------
// z_values: z[r,s] = linspace(near,far,S)[s] + (u * (far-near)) / S
// ------------------------------------------------------------------------------------------------
__global__ void z_values_kernel(float start, float stop, float delta, float span, long long N, int S,
                                const float* __restrict__ u, uint64_t seed, long long ray_base,
                                float* __restrict__ z) {
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= N * S) return;
    const long long r = m / S;
    const int s = (int)(m - r * S);
    // tf.linspace: first = start, last = stop exactly, interior = start + delta*i
    float lin = __fadd_rn(start, __fmul_rn(delta, (float)s));
    if (s == 0) lin = start;
    if (s == S - 1 && S > 1) lin = stop;
    const float uu = u ? u[m] : philox_uniform(seed, (uint64_t)(ray_base + r), s, 0u);
    z[m] = __fadd_rn(lin, __fdiv_rn(__fmul_rn(uu, span), (float)S));
}

void launch_z_values(float near_b, float far_b, long long N, int S, const float* u, uint64_t seed,
                     long long ray_base, float* z, hipStream_t stream) {
    if (N <= 0) return;
    const float delta = S > 1 ? (far_b - near_b) / (float)(S - 1) : 0.f;
    const float span = (float)((double)far_b - (double)near_b);
    const long long total = N * S;
    const int bs = 256;
    hipLaunchKernelGGL(z_values_kernel, dim3((unsigned)((total + bs - 1) / bs)), dim3(bs), 0, stream, near_b,
                       far_b, delta, span, N, S, u, seed, ray_base, z);
}

// ------------------------------------------------------------------------------------------------
// sample_pdf: one ray per wavefront.  LDS per wave: cdf[S], mid[S-1], znew[Sf], zall[S+Sf].
//   1. lane-parallel load of w,z; sequential (lane 0) sum and cumsum -> cdf  (canonical order)
//   2. each lane: fine draws k = lane, lane+64, ...: binary search (searchsorted left), clip,
//      1e-5 floor, lerp between bin midpoints
//   3. rank sort of the Sf new depths, then a binary-search merge with the (already increasing) coarse
//      depths for the fine pass
// ------------------------------------------------------------------------------------------------
constexpr int kPdfWaves = 4;   // rays per workgroup

__global__ __launch_bounds__(64 * kPdfWaves) void sample_pdf_kernel(
    const float* __restrict__ weights, const float* __restrict__ zin, long long N, int S, int Sf,
    const float* __restrict__ u, uint64_t seed, long long ray_base, float* __restrict__ z_new,
    float* __restrict__ z_merged) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long long ray = (long long)blockIdx.x * kPdfWaves + wave;
    if (ray >= N) return;   // whole wave exits; no block-level barrier is used below
    // per-wave LDS arrays, each padded to a multiple of 4 floats so the rank sorts can read float4
    const int S4 = (S + 3) & ~3, Sf4 = (Sf + 3) & ~3, T = S + Sf, T4 = (T + 3) & ~3;
    const int per_wave = S4 + S4 + Sf4 + T4;
    float* cdf = lds + wave * per_wave;   // first holds w, then pdf, then cdf
    float* zc = cdf + S4;
    float* zn = zc + S4;
    float* za = zn + Sf4;
    const float* wr = weights + ray * S;
    const float* zr = zin + ray * S;
    // Canonical left-to-right sum and cumsum of w / (sum + 1e-7).  The order is sequential by definition,
    // but only the additions are: every lane keeps its samples in registers, the running value is formed
    // redundantly in all lanes from v_readlane broadcasts (no LDS round trip, no single-lane loop) and
    // the divisions run 64 wide.  S <= 256 takes this path; longer rays fall back to a one-lane loop.
    constexpr int kMaxChunks = 4;
    if (S <= 64 * kMaxChunks) {
        float wreg[kMaxChunks], creg[kMaxChunks];
#pragma unroll
        for (int c = 0; c < kMaxChunks; ++c) {
            const int s = lane + 64 * c;
            wreg[c] = s < S ? wr[s] : 0.f;
            if (s < S) zc[s] = zr[s];
        }
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxChunks; ++c) {
            const int n = min(64, S - 64 * c);
            for (int i = 0; i < n; ++i) sum = __fadd_rn(sum, __shfl(wreg[c], i));
        }
        const float den = __fadd_rn(sum, 1e-7f);
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxChunks; ++c) {
            const float pdf = __fdiv_rn(wreg[c], den);
            const int n = min(64, S - 64 * c);
            creg[c] = 0.f;
            for (int i = 0; i < n; ++i) {
                acc = __fadd_rn(acc, __shfl(pdf, i));
                if (lane == i) creg[c] = acc;
            }
        }
#pragma unroll
        for (int c = 0; c < kMaxChunks; ++c) {
            const int s = lane + 64 * c;
            if (s < S) cdf[s] = creg[c];
        }
    } else {
        for (int s = lane; s < S; s += 64) { cdf[s] = wr[s]; zc[s] = zr[s]; }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) {
            float sum = 0.f;
            for (int s = 0; s < S; ++s) sum = __fadd_rn(sum, cdf[s]);
            const float den = __fadd_rn(sum, 1e-7f);
            float acc = 0.f;
            for (int s = 0; s < S; ++s) {
                acc = __fadd_rn(acc, __fdiv_rn(cdf[s], den));
                cdf[s] = acc;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float kInf = __builtin_huge_valf();
    for (int k = lane; k < Sf4; k += 64) {
        if (k >= Sf) { zn[k] = kInf; continue; }   // padding never ranks below a real sample
        const float uu = u ? u[ray * Sf + k] : philox_uniform(seed, (uint64_t)(ray_base + ray), k, 1u);
        // searchsorted(cdf, uu, side='left'): first i with cdf[i] >= uu, in [0, S]
        int lo = 0, hi = S;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] < uu) lo = mid + 1; else hi = mid;
        }
        const int idx = lo;
        const int b = max(0, idx - 1);
        const int t = min(S - 1, idx);
        const float c_lo = cdf[b], c_hi = cdf[t];
        const int bz = min(max(b, 0), S - 2), tz = min(max(t, 0), S - 2);
        const float z_lo = __fmul_rn(0.5f, __fadd_rn(zc[bz + 1], zc[bz]));
        const float z_hi = __fmul_rn(0.5f, __fadd_rn(zc[tz + 1], zc[tz]));
        float den = __fsub_rn(c_hi, c_lo);
        den = den < 1e-5f ? 1e-5f : den;
        const float tt = __fdiv_rn(__fsub_rn(uu, c_lo), den);
        zn[k] = __fadd_rn(z_lo, __fmul_rn(tt, __fsub_rn(z_hi, z_lo)));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // rank sort (ties broken by index -> a permutation); every lane sweeps the array as float4 broadcasts
    auto rank_of = [](const float* arr, int n4, float v, int k) -> int {
        int rank = 0;
        for (int i = 0; i < n4; i += 4) {
            const float4 o = *reinterpret_cast<const float4*>(arr + i);
            rank += (o.x < v || (o.x == v && i + 0 < k)) ? 1 : 0;
            rank += (o.y < v || (o.y == v && i + 1 < k)) ? 1 : 0;
            rank += (o.z < v || (o.z == v && i + 2 < k)) ? 1 : 0;
            rank += (o.w < v || (o.w == v && i + 3 < k)) ? 1 : 0;
        }
        return rank;
    };
    for (int k = lane; k < Sf; k += 64) {
        const float v = zn[k];
        const int rank = rank_of(zn, Sf4, v, k);
        if (z_new) z_new[ray * Sf + rank] = v;
        za[rank] = v;                                   // za[0..Sf) = the new depths, sorted
    }
    if (z_merged) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // The stratified coarse depths are increasing by construction (UtilsCV.py:578-580); a caller of
        // the stand-alone entry point may pass anything, so check before relying on it.
        bool bad = false;
        for (int s = lane; s + 1 < S; s += 64) bad |= zc[s + 1] < zc[s];
        if (!__any(bad)) {
            // sort(concat(new, coarse)) = merge of two sorted runs: output slot = own index + number of
            // elements of the OTHER run that precede it, by binary search instead of an O(n^2) sweep
            for (int k = lane; k < Sf; k += 64) {
                const float v = za[k];
                int lo = 0, hi = S;                      // # coarse depths < v
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (zc[mid] < v) lo = mid + 1; else hi = mid; }
                z_merged[ray * T + k + lo] = v;
            }
            for (int j = lane; j < S; j += 64) {
                const float v = zc[j];
                int lo = 0, hi = Sf;                     // # new depths <= v
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (za[mid] <= v) lo = mid + 1; else hi = mid; }
                z_merged[ray * T + j + lo] = v;
            }
        } else {
            // general case: rank sort of the concatenation
            for (int k = lane; k < T4; k += 64) if (k >= Sf) za[k] = k < T ? zc[k - Sf] : kInf;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int k = lane; k < T; k += 64) {
                const float v = za[k];
                z_merged[ray * T + rank_of(za, T4, v, k)] = v;
            }
        }
    }
}

size_t sample_pdf_lds_bytes(int S, int Sf) {
    const int S4 = (S + 3) & ~3, Sf4 = (Sf + 3) & ~3, T4 = (S + Sf + 3) & ~3;
    return (size_t)kPdfWaves * (S4 + S4 + Sf4 + T4) * sizeof(float);
}

void launch_sample_pdf(const float* weights, const float* z, long long N, int S, int Sf, const float* u,
                       uint64_t seed, long long ray_base, float* z_new, float* z_merged, hipStream_t stream) {
    if (N <= 0) return;
    const size_t lds = sample_pdf_lds_bytes(S, Sf);
    hipLaunchKernelGGL(sample_pdf_kernel, dim3((unsigned)((N + kPdfWaves - 1) / kPdfWaves)), dim3(64 * kPdfWaves),
                       lds, stream, weights, z, N, S, Sf, u, seed, ray_base, z_new, z_merged);
}

// ------------------------------------------------------------------------------------------------
// composite (ray_marching, src/UtilsNeuralRadianceField.py:88-115): ONE RAY PER WAVEFRONT.  The 64 lanes take 64
// consecutive samples (one coalesced 1 KiB read of the raw rows), alpha / sigmoid -- the expensive part -- are evaluated
// in parallel, the exclusive transmittance travels through the lanes in the canonical left-to-right order (below),
// carried from one 64-sample chunk to the next, and the rgb / depth sums are per-lane partial sums closed by a
// butterfly (same terms as a left-to-right loop, another association: a few ulps).  Same result on every run and
// for every batching of the rays.  The first version walked one ray per LANE: 64 wavefronts for a 4096-ray batch and
// a 3 KiB stride between the lanes of a load.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                        long long N, int S, float* __restrict__ rgb,
                                                        float* __restrict__ weights, float* __restrict__ cumprod,
                                                        float* __restrict__ alpha_out, float* __restrict__ rgb_samples,
                                                        float* __restrict__ depth) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;                                        // uniform per wavefront
    const float4* rw = reinterpret_cast<const float4*>(raw) + r * S;
    const float* zr = z + r * S;
    float carry = 1.0f, c0 = 0.f, c1 = 0.f, c2 = 0.f, dep = 0.f;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const bool in = s < S;
        const float4 o = in ? rw[s] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float zc = in ? zr[s] : 0.f;
        const float delta = s + 1 < S ? zr[s + 1] - zc : 1e9f;
        const float sigma = fmaxf(o.w, 0.f);
        const float a = in ? 1.0f - expf(-(sigma * delta)) : 0.f;
        const float r0 = 1.0f / (1.0f + expf(-o.x));
        const float r1 = 1.0f / (1.0f + expf(-o.y));
        const float r2 = 1.0f / (1.0f + expf(-o.z));
        // Exclusive transmittance in the canonical order T_l = fl(T_(l-1) * (1 - alpha_(l-1))): lane l reads its left
        // neighbour through a one-lane wavefront shift (DPP wave_shr:1; lane 0 keeps the carry) and multiplies; after k
        // rounds lanes 0..k are final and recomputing them changes nothing, so 63 rounds of one shift + one multiply
        // finish the chunk.  (A tree-shaped prefix product is six steps, but neighbouring T then differ from the
        // ratio 1 - alpha by a few ulps -- the inverse-CDF sampler amplifies exactly that, by up to 1e5.)
        const float om = 1.0f - a;
        const float oms = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(1.0f), __float_as_int(om), 0x138, 0xf, 0xf, false));
        float T = lane == 0 ? carry : 1.0f;
        const int rounds = S - s0 < 64 ? S - s0 - 1 : 63;
        for (int k = 0; k < rounds; ++k) {
            const float Ts = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(T), __float_as_int(T), 0x138, 0xf, 0xf, false));
            T = Ts * (lane == 0 ? 1.0f : oms);
        }
        const float w = a * T;
        c0 += w * r0; c1 += w * r1; c2 += w * r2; dep += w * zc;
        if (in) {
            const long long m = r * S + s;
            if (weights) weights[m] = w;
            if (cumprod) cumprod[m] = T;
            if (alpha_out) alpha_out[m] = a;
            if (rgb_samples) { rgb_samples[m * 3 + 0] = r0; rgb_samples[m * 3 + 1] = r1; rgb_samples[m * 3 + 2] = r2; }
        }
        carry = __shfl(T * om, 63);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c0 += __shfl_xor(c0, o); c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); dep += __shfl_xor(dep, o);
    }
    if (lane == 0) {
        if (rgb) { rgb[r * 3 + 0] = c0; rgb[r * 3 + 1] = c1; rgb[r * 3 + 2] = c2; }
        if (depth) depth[r] = dep;
    }
}

void launch_composite(const float* raw, const float* z, long long N, int S, float* rgb, float* weights,
                      float* cumprod, float* alpha, float* rgb_samples, float* depth, hipStream_t stream) {
    if (N <= 0) return;
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, raw, z, N, S,
                       rgb, weights, cumprod, alpha, rgb_samples, depth);
}

// ------------------------------------------------------------------------------------------------
// posenc (standalone): x (M,3) -> [x, sin0,cos0,...]*3 (passthrough) or [sin0,cos0,...]*3
// ------------------------------------------------------------------------------------------------
__global__ void posenc_kernel(const float* __restrict__ x, long long M, int n_enc, int passthrough,
                              float* __restrict__ out) {
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int per = (passthrough ? 1 : 0) + 2 * n_enc;
    const float kPi = 3.1415927410125732f;
    for (int c = 0; c < 3; ++c) {
        const float v = x[m * 3 + c];
        float* o = out + m * (3 * per) + c * per;
        if (passthrough) *o++ = v;
        for (int k = 0; k < n_enc; ++k) {
            const float th = __fmul_rn(v, kPi * (float)(1 << k));
            o[2 * k] = sin_shifted(th, 0);
            o[2 * k + 1] = sin_shifted(th, 1);
        }
    }
}

void launch_posenc(const float* x, long long M, int n_enc, int passthrough, float* out, hipStream_t stream) {
    if (M <= 0) return;
    const int bs = 256;
    hipLaunchKernelGGL(posenc_kernel, dim3((unsigned)((M + bs - 1) / bs)), dim3(bs), 0, stream, x, M, n_enc,
                       passthrough, out);
}

}  // namespace nerf

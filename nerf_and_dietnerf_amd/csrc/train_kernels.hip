// train_kernels.hip -- HIP kernels of the training path (NeRF.train_step, src/NeRF.py:136-178) for gfx950.
//
// The render path fuses the whole MLP into one kernel and never writes an activation; a training step needs
// every layer's activation again in the backward pass, so here the network runs layer by layer over
// activations resident in HBM (9 KB per sample row; 7 GB for a 4096-ray batch -- 2.5 % of the 288 GB):
//
//   gemm_abt   Out = epi(A . Bt^T)        forward layers (bias + LeakyReLU fused) and the data gradients
//                                         (LeakyReLU' mask and the sigma head's rank-1 term fused)
//   gemm_atb   dW  = A^T . G              weight gradients: the reduction runs over ~10^6 sample rows, so it is
//                                         split into row slabs (partials in HBM) and summed in a fixed order;
//                                         the bias gradient (column sums of G) rides in the same kernel
// both on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation), 128x128 output tile per workgroup,
// 4 waves x (64x64), k-step 16 through double-buffered LDS.
//
// Per-ray backward kernels: compositing (division-free reverse scan), positional encoding, inverse-CDF sampler
// (the reference has no stop_gradient there: src/UtilsCV.py:512-537), MSE, Adam.
#include "train_kernels.h"
#include "nerf_device.h"

namespace nerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// acc += a (x) b on v_mfma_f32_32x32x2_f32.  (Forcing the accumulators into AGPRs with an inline-asm MFMA -- hipcc keeps
// them in arch VGPRs here -- measured 2 % slower: the volatile asm pins the schedule.)
__device__ __forceinline__ void mfma_acc(f32x16& acc, float a, float b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// gemm_abt
// ------------------------------------------------------------------------------------------------
// LDS tiles are k-major: element (k, row) of a [rows x 16] operand tile sits at k * (rows + 4) + row, so that the
// MFMA operand reads (lane = row, 8 k values per lane half) are 8 conflict-free ds_read_b32 -- consecutive lanes hit
// consecutive banks, the two lane halves (k + 8) land 32 banks apart -- and the transposing stores (4 dwords of one
// row each) spread over all 64 banks.  (A row-major tile read with ds_read_b128 measured 0.37 bank-conflict cycles
// per busy cycle: SQ_LDS_BANK_CONFLICT, rocprofv3 --pmc.)

// One k-step (kKT = 32) of the workgroup tile: operands from one LDS buffer into the wave's MFMA tiles.
// A k-step of 32 makes every staged row a full 128-byte line of the row-major operand (64-byte half lines with a
// 16-wide step held both gemm_abt variants at 2.5 TB/s of HBM traffic) and halves the barriers per FLOP.
constexpr int kKT = 32;

template <int WTM, int WTN, int LDA, int LDB>
__device__ __forceinline__ void abt_compute(const float* As, const float* Bs, int a_row0, int b_row0, int li, int lh,
                                            f32x16 (&acc)[WTM][WTN]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // two halves of 8 MFMA steps: 16 operand registers live at a time
        float av[WTM][8], bv[WTN][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int a = 0; a < WTM; ++a) av[a][j] = As[(16 * lh + 8 * h + j) * LDA + a_row0 + a * 32 + li];
#pragma unroll
            for (int b = 0; b < WTN; ++b) bv[b][j] = Bs[(16 * lh + 8 * h + j) * LDB + b_row0 + b * 32 + li];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int a = 0; a < WTM; ++a)
#pragma unroll
                for (int b = 0; b < WTN; ++b)
                    mfma_acc(acc[a][b], av[a][j], bv[b][j]);
    }
}

__device__ __forceinline__ void park4(float* tile, int ld, int k4, int row, const float4& v) {
    tile[(k4 + 0) * ld + row] = v.x;
    tile[(k4 + 1) * ld + row] = v.y;
    tile[(k4 + 2) * ld + row] = v.z;
    tile[(k4 + 3) * ld + row] = v.w;
}

#ifndef NERF_ABT_MAXW
#define NERF_ABT_MAXW 4
#endif
template <int WTM, int WTN, int WVM, int WVN, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, NERF_ABT_MAXW))) void gemm_abt_kernel(const GemmAbt g) {
    constexpr int TM = 32 * WTM * WVM, TN = 32 * WTN * WVN;
    // k-major tiles [32][rows + 2]: operand reads (consecutive lanes = consecutive rows, lane halves 16 k apart = 32
    // banks apart) and the transposing stores (8 lanes x 4 dwords per row) are both bank-conflict free
    constexpr int LDA = TM + 2, LDB = TN + 2;
    static_assert(WVM * WVN == 4 && TM == 128, "4 waves, 128 rows per workgroup");
    static_assert(TN == 128 || TN == 32, "staging below: 4 float4 per thread (128 rows) or 1 (32 rows)");
    __shared__ __attribute__((aligned(16))) float As[2][kKT * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][kKT * LDB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WVN, wn = wave % WVN;
    const int li = lane & 31, lh = lane >> 5;
    // 1-D grid.  Workgroups are dealt round-robin to the 8 XCDs (each with its own L2): the column tiles of one row
    // tile get ids 8 apart, i.e. the same XCD back to back, so the second one finds the A rows in that L2.
    const int n_tiles = g.N / TN;
    const long long lin = blockIdx.x, grp = lin / (8 * n_tiles), rem = lin % (8 * n_tiles);
    long long m_tile = grp * 8 + rem % 8;
    int n_tile = (int)(rem / 8);
    const long long m_tiles = g.M / TM;
    if (grp * 8 + 8 > m_tiles) {          // ragged last group: plain order
        const long long base = grp * 8 * n_tiles, r2 = lin - base, left = m_tiles - grp * 8;
        m_tile = grp * 8 + r2 % left;
        n_tile = (int)(r2 / left);
    }
    const long long m0 = m_tile * TM;
    const int n0 = n_tile * TN;
    constexpr bool B_WIDE = TN == 128;
    // staging slots of this thread: float4 #(t + 256 i) of a [rows x 32] tile = row (t >> 3) + 32 i, k 4 (t & 7)
    const int srow = t >> 3, sk4 = 4 * (t & 7);
    const float* ap = g.A + (m0 + srow) * g.lda + sk4;
    const float* bp = g.Bt + (size_t)(n0 + srow) * g.ldb + sk4;

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int a = 0; a < WTM; ++a)
#pragma unroll
        for (int b = 0; b < WTN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // (scalars, not arrays: hipcc otherwise parks small private arrays in LDS / scratch and waits on every load)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t a1 = (size_t)32 * g.lda, b1 = (size_t)32 * g.ldb;
    float4 ra0, ra1, ra2, ra3, rb0, rb1 = z4, rb2 = z4, rb3 = z4;
#define ABT_FETCH(K0)                                                                      \
    ra0 = *reinterpret_cast<const float4*>(ap + (K0));                                     \
    ra1 = *reinterpret_cast<const float4*>(ap + a1 + (K0));                                \
    ra2 = *reinterpret_cast<const float4*>(ap + 2 * a1 + (K0));                            \
    ra3 = *reinterpret_cast<const float4*>(ap + 3 * a1 + (K0));                            \
    rb0 = *reinterpret_cast<const float4*>(bp + (K0));                                     \
    if (B_WIDE) {                                                                          \
        rb1 = *reinterpret_cast<const float4*>(bp + b1 + (K0));                            \
        rb2 = *reinterpret_cast<const float4*>(bp + 2 * b1 + (K0));                        \
        rb3 = *reinterpret_cast<const float4*>(bp + 3 * b1 + (K0));                        \
    }
#define ABT_PARK(BUF)                                                                      \
    park4(As[BUF], LDA, sk4, srow, ra0);                                                   \
    park4(As[BUF], LDA, sk4, srow + 32, ra1);                                              \
    park4(As[BUF], LDA, sk4, srow + 64, ra2);                                              \
    park4(As[BUF], LDA, sk4, srow + 96, ra3);                                              \
    park4(Bs[BUF], LDB, sk4, srow, rb0);                                                   \
    if (B_WIDE) {                                                                          \
        park4(Bs[BUF], LDB, sk4, srow + 32, rb1);                                          \
        park4(Bs[BUF], LDB, sk4, srow + 64, rb2);                                          \
        park4(Bs[BUF], LDB, sk4, srow + 96, rb3);                                          \
    }
    // data-gradient epilogue: the LeakyReLU' mask comes from the stored activation H.  Fetching it here, before the
    // k loop, hides the 64 scattered loads per lane behind the MFMAs (fetched in the epilogue they were the kernel's
    // main stall: SQ_WAIT_ANY 0.52 of the wave cycles).
    float hmask[WTM][WTN][16];
    if (EPI == EPI_BWD_MASK) {
#pragma unroll
        for (int a = 0; a < WTM; ++a)
#pragma unroll
            for (int b = 0; b < WTN; ++b) {
                const int n = n0 + (wn * WTN + b) * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = m0 + (wm * WTM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    hmask[a][b][r] = g.H[m * g.ldh + n];
                }
            }
    }
    ABT_FETCH(0)
    ABT_PARK(0)
    __syncthreads();
    int buf = 0;
    // steady state: fetch k-step s+1 into registers, multiply k-step s out of LDS, park s+1 in the other buffer
    for (int k0 = kKT; k0 < g.K; k0 += kKT) {
        ABT_FETCH(k0)
        __builtin_amdgcn_sched_barrier(0);   // keep the fetch ahead of the MFMA block (hipcc sinks it to save VGPRs)
        abt_compute<WTM, WTN, LDA, LDB>(As[buf], Bs[buf], wm * WTM * 32, wn * WTN * 32, li, lh, acc);
        __builtin_amdgcn_sched_barrier(0);
        ABT_PARK(buf ^ 1)
        __syncthreads();
        buf ^= 1;
    }
#undef ABT_FETCH
#undef ABT_PARK
    abt_compute<WTM, WTN, LDA, LDB>(As[buf], Bs[buf], wm * WTM * 32, wn * WTN * 32, li, lh, acc);

    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    float vmax = 0.f;
#pragma unroll
    for (int a = 0; a < WTM; ++a)
#pragma unroll
        for (int b = 0; b < WTN; ++b) {
            const int n = n0 + (wn * WTN + b) * 32 + li;
            float bias = 0.f, r1b = 0.f;
            if (EPI == EPI_FWD_LEAKY || EPI == EPI_FWD_LINEAR) bias = g.bias[n];
            if (EPI == EPI_BWD_MASK && g.r1a) r1b = g.r1b[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + (wm * WTM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[a][b][r];
                if (EPI == EPI_FWD_LEAKY) {
                    v += bias;
                    v = v > 0.f ? v : g.alpha * v;
                } else if (EPI == EPI_FWD_LINEAR) {
                    v += bias;
                } else if (EPI == EPI_BWD_MASK) {
                    if (g.r1a) v = fmaf(g.r1a[m * g.r1a_ld], r1b, v);
                    v = hmask[a][b][r] > 0.f ? v : g.alpha * v;
                    vmax = fmaxf(vmax, fabsf(v));
                } else {
                    if (g.accumulate) v += g.Out[m * g.ldo + n];
                }
                if (n < g.n_valid) g.Out[m * g.ldo + n] = v;
            }
        }
    if (EPI == EPI_BWD_MASK && g.gmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        // 64 slots per buffer and a plain read first: same-address atomics from 32 k waves would serialise in L2
        if (lane == 0) {
            unsigned* slot = g.gmax + (blockIdx.x & 63);
            const unsigned vb = __float_as_uint(vmax);               // non-negative floats order like their bits
            if (vb > *slot) atomicMax(slot, vb);
        }
    }
}

template <int EPI>
static void launch_abt_epi(bool narrow, const GemmAbt& g, hipStream_t s) {
    if (narrow)
        hipLaunchKernelGGL((gemm_abt_kernel<1, 1, 4, 1, EPI>), dim3((unsigned)((g.M / 128) * (g.N / 32))), dim3(256), 0,
                           s, g);
    else     // (a 128x256 tile -- A read once, occupancy 2 -- measured 9 % slower than two 128x128 tiles at occupancy 3)
        hipLaunchKernelGGL((gemm_abt_kernel<2, 2, 2, 2, EPI>), dim3((unsigned)((g.M / 128) * (g.N / 128))), dim3(256), 0,
                           s, g);
}

void launch_gemm_abt(int epi, bool narrow, const GemmAbt& g, hipStream_t s) {
    if (g.M <= 0) return;
    switch (epi) {
        case EPI_FWD_LEAKY: launch_abt_epi<EPI_FWD_LEAKY>(narrow, g, s); break;
        case EPI_FWD_LINEAR: launch_abt_epi<EPI_FWD_LINEAR>(narrow, g, s); break;
        case EPI_BWD_MASK: launch_abt_epi<EPI_BWD_MASK>(narrow, g, s); break;
        default: launch_abt_epi<EPI_BWD_PLAIN>(narrow, g, s); break;
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_atb: 128 (k of A) x 128 (n of G) partial tile per workgroup over one slab of rows.
// ------------------------------------------------------------------------------------------------
constexpr int kAtbLd = 132;   // staged row of 128 floats + 4: lane halves (8 rows apart) fall on opposite banks

__device__ __forceinline__ void atb_compute(const float* As, const float* Gs, int wk, int wn, int li, int lh, int t,
                                            bool do_colsum, f32x16 (&acc)[2][2], float& colsum) {
    // MFMA: D[i][j] += sum_kk A[i][kk] * B[kk][j] with i = column of A (row of dW), kk = sample row, j = column of G
    float av[2][8], bv[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float* pa = As + (8 * lh + j) * kAtbLd + wk * 64 + li;
        const float* pg = Gs + (8 * lh + j) * kAtbLd + wn * 64 + li;
        av[0][j] = pa[0]; av[1][j] = pa[32];
        bv[0][j] = pg[0]; bv[1][j] = pg[32];
    }
    if (do_colsum) {
#pragma unroll
        for (int r = 0; r < 16; ++r) colsum += Gs[r * kAtbLd + t];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                mfma_acc(acc[a][b], av[a][j], bv[b][j]);
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 3))) void gemm_atb_kernel(const GemmAtb g) {
    __shared__ __attribute__((aligned(16))) float As[2][16 * kAtbLd];
    __shared__ __attribute__((aligned(16))) float Gs[2][16 * kAtbLd];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    // 1-D grid; the tiles of one row slab get ids 8 apart (same XCD, back to back): the slab's A and G rows are then
    // read from HBM once and served to the other tiles out of that XCD's L2.
    const int kt_n = (g.Kp + 127) / 128, nt_n = (g.Nw + 127) / 128, T = kt_n * nt_n;
    const int n_splits = (int)((g.M + g.rows_per_split - 1) / g.rows_per_split);
    const int lin = blockIdx.x, grp = lin / (8 * T), rem = lin % (8 * T);
    int split = grp * 8 + rem % 8, tile = rem / 8;
    if (grp * 8 + 8 > n_splits) {         // ragged last group: plain order
        const int r2 = lin - grp * 8 * T, left = n_splits - grp * 8;
        split = grp * 8 + r2 % left;
        tile = r2 / left;
    }
    const int kb = (tile % kt_n) * 128, nb = (tile / kt_n) * 128;
    const bool first_ktile = tile % kt_n == 0;
    const long long ms = (long long)split * g.rows_per_split;
    const long long me = ms + g.rows_per_split < g.M ? ms + g.rows_per_split : g.M;
    // staging slots of this thread: float4 #(t + 256 i), i = 0, 1, of a [16 rows x 128 cols] tile
    const int srow = t >> 5, sc4 = t & 31;
    const bool a_on = kb + 4 * sc4 < g.K, g_on = nb + 4 * sc4 < g.N;   // columns past K / N stage zeros
    const float* ap = g.A + (ms + srow) * g.lda + kb + 4 * sc4;
    const float* gp = g.G + (ms + srow) * g.ldg + nb + 4 * sc4;
    const int soff = srow * kAtbLd + 4 * sc4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float colsum = 0.f;
    const bool do_colsum = first_ktile && t < 128;

    if (ms < me) {      // uniform per workgroup
        const size_t a8 = (size_t)8 * g.lda, g8 = (size_t)8 * g.ldg;
        float4 ra0 = a_on ? *reinterpret_cast<const float4*>(ap) : zero4;
        float4 ra1 = a_on ? *reinterpret_cast<const float4*>(ap + a8) : zero4;
        float4 rg0 = g_on ? *reinterpret_cast<const float4*>(gp) : zero4;
        float4 rg1 = g_on ? *reinterpret_cast<const float4*>(gp + g8) : zero4;
        *reinterpret_cast<float4*>(&As[0][soff]) = ra0;
        *reinterpret_cast<float4*>(&As[0][soff + 8 * kAtbLd]) = ra1;
        *reinterpret_cast<float4*>(&Gs[0][soff]) = rg0;
        *reinterpret_cast<float4*>(&Gs[0][soff + 8 * kAtbLd]) = rg1;
        __syncthreads();
        int buf = 0;
        const long long steps = (me - ms) / 16;
        for (long long st = 1; st < steps; ++st) {
            const float* aq = ap + (size_t)st * 16 * g.lda;
            const float* gq = gp + (size_t)st * 16 * g.ldg;
            if (a_on) { ra0 = *reinterpret_cast<const float4*>(aq); ra1 = *reinterpret_cast<const float4*>(aq + a8); }
            if (g_on) { rg0 = *reinterpret_cast<const float4*>(gq); rg1 = *reinterpret_cast<const float4*>(gq + g8); }
            __builtin_amdgcn_sched_barrier(0);   // keep the fetch ahead of the MFMA block
            atb_compute(As[buf], Gs[buf], wk, wn, li, lh, t, do_colsum, acc, colsum);
            __builtin_amdgcn_sched_barrier(0);
            *reinterpret_cast<float4*>(&As[buf ^ 1][soff]) = ra0;
            *reinterpret_cast<float4*>(&As[buf ^ 1][soff + 8 * kAtbLd]) = ra1;
            *reinterpret_cast<float4*>(&Gs[buf ^ 1][soff]) = rg0;
            *reinterpret_cast<float4*>(&Gs[buf ^ 1][soff + 8 * kAtbLd]) = rg1;
            __syncthreads();
            buf ^= 1;
        }
        atb_compute(As[buf], Gs[buf], wk, wn, li, lh, t, do_colsum, acc, colsum);
    }

    float* part = g.partial + (size_t)split * (g.Kp + 1) * g.Nw;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = nb + wn * 64 + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kb + wk * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (k < g.Kp && n < g.Nw) part[(size_t)k * g.Nw + n] = acc[a][b][r];
            }
        }
    if (do_colsum && nb + t < g.Nw) part[(size_t)g.Kp * g.Nw + nb + t] = colsum;
}

void launch_gemm_atb(const GemmAtb& g, hipStream_t s) {
    const int splits = (int)((g.M + g.rows_per_split - 1) / g.rows_per_split);
    const int tiles = ((g.Kp + 127) / 128) * ((g.Nw + 127) / 128);
    hipLaunchKernelGGL(gemm_atb_kernel, dim3((unsigned)(tiles * splits)), dim3(256), 0, s, g);
}

// ------------------------------------------------------------------------------------------------
// gemm_atb_h: the weight-gradient GEMM on the fp16 matrix cores.  Both operands are split x = hi + lo (two fp16 values,
// 22 significant bits) while they are staged, and each product is formed in three passes hi*lo + lo*hi + hi*hi of
// v_mfma_f32_32x32x16_f16 with fp32 accumulation -- the arithmetic of the render path's f16x3 kernel.  Gradients are tiny
// (1e-5 and below), so G is multiplied by a power of two that puts its largest entry (tracked by its producer with an
// atomicMax) at 2^14; the partial sums are divided by it again.  Staging: 128 threads per operand, each a 4 rows x 4
// columns block (four global float4 loads), transposed in registers so that the 16 sample rows of a column land
// contiguously: the MFMA operand of lane (column, half) is one ds_read_b128.  LDS: [plane][column][16 rows] fp16, 48-byte
// column stride (conflict-free for the 64 x 16-byte operand reads).
// ------------------------------------------------------------------------------------------------
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
constexpr int kHColStride = 48;                       // bytes per column of a plane

// hi = the top 11 significand bits (exact in fp16 for normal-range values), lo = v - hi rounded to fp16: two plain VALU ops
// and one v_cvt_pk_f16_f32 per half pair (beside the fp16 MFMA, conversions are the expensive instructions).  Values
// below the fp16 normal range (|v| < 2^-14, i.e. < 2^-28 of the largest entry after scaling) keep an absolute error
// of 2^-25: irrelevant in sums dominated by entries ten orders of magnitude larger.
__device__ __forceinline__ void split_pack2(float v0, float v1, uint32_t& hi, uint32_t& lo) {
    const float h0 = __uint_as_float(__float_as_uint(v0) & 0xFFFFE000u), h1 = __uint_as_float(__float_as_uint(v1) & 0xFFFFE000u);
    const h2v h = {(_Float16)h0, (_Float16)h1};
    const h2v l = {(_Float16)(v0 - h0), (_Float16)(v1 - h1)};
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

// W = tile width and height (128: four waves of 64 x 64; 256: eight waves of 128 x 64, A and G each read once per slab)
// The weight-gradient kernels take a BATCH of GEMMs (GemmAtbBatch): the workgroups of entry e are blockIdx.x in
// [wg_end[e-1], wg_end[e]) -- the trainer hands the eight 256-wide layers of a pass to ONE launch (a chip-filling grid with
// a quarter of the row slabs per layer, hence a quarter of the partial-sum traffic, and one kernel boundary instead of eight).
__device__ __forceinline__ int batch_entry(const GemmAtbBatch& b, int& lin) {
    int e = 0;
    while (e + 1 < b.n && lin >= b.wg_end[e]) ++e;
    if (e > 0) lin -= b.wg_end[e - 1];
    return e;
}

template <int W>
__global__ __launch_bounds__(2 * W) __attribute__((amdgpu_waves_per_eu(2, W == 128 ? 3 : 2))) void gemm_atb_h_kernel(const GemmAtbBatch bat) {
    int lin = blockIdx.x;
    const GemmAtb& g = bat.e[batch_entry(bat, lin)];
    constexpr int kPl = W * kHColStride;          // bytes per plane
    constexpr int WNW = W / 64;                   // waves along n (2 or 4); two along k
    constexpr int KTL = W / 64;                   // 32-row k tiles per wave (2 or 4); two n tiles per wave
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][4 * kPl];    // planes: A hi, A lo, G hi, G lo
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = wave / WNW, wn = wave % WNW;
    const int li = lane & 31, lh = lane >> 5;
    const int kt_n = (g.Kp + W - 1) / W, nt_n = (g.Nw + W - 1) / W, T = kt_n * nt_n;
    const int n_splits = (int)((g.M + g.rows_per_split - 1) / g.rows_per_split);
    if (lin >= T * n_splits) return;              // padding workgroups of a batch entry (its range is a multiple of 8)
    const int grp = lin / (8 * T), rem = lin % (8 * T);
    int split = grp * 8 + rem % 8, tile = rem / 8;
    if (grp * 8 + 8 > n_splits) {
        const int r2 = lin - grp * 8 * T, left = n_splits - grp * 8;
        split = grp * 8 + r2 % left;
        tile = r2 / left;
    }
    const int kb = (tile % kt_n) * W, nb = (tile / kt_n) * W;
    const bool first_ktile = tile % kt_n == 0;
    const long long ms = (long long)split * g.rows_per_split;
    const long long me = ms + g.rows_per_split < g.M ? ms + g.rows_per_split : g.M;
    // power-of-two scale of G: largest entry -> [2^14, 2^15)
    unsigned mb = g.gmax ? g.gmax[lane] : 0u;                  // 64 slots per buffer (see the producers)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned other = __shfl_xor(mb, o); mb = other > mb ? other : mb; }
    int sexp = mb ? 127 + 14 - ((int)((mb >> 23) & 0xFF) - 127) : 127;
    sexp = sexp < 1 ? 1 : sexp > 254 ? 254 : sexp;
    const float gscale = __uint_as_float((unsigned)sexp << 23), ginv = 1.0f / gscale;
    // staging role of this thread: operand (A: t < W, G: t >= W), rows 4 rg .. 4 rg + 3, columns 4 cg .. 4 cg + 3
    const bool isG = t >= W;
    const int b = t & (W - 1), rg = b & 3, cg = b >> 2;
    const int ld = isG ? g.ldg : g.lda;
    const bool on = isG ? (nb + 4 * cg < g.N) : (kb + 4 * cg < g.K);
    // threads whose columns lie outside the matrix read column block 0 (valid memory) and park zeros
    const int col0 = on ? (isG ? nb : kb) + 4 * cg : 0;
    // fragment-major operands (frag_layout.h::frag_index): the thread's 4 rows x 4 columns are 16 consecutive floats
    const int rs = g.frag ? 4 : ld;                                   // floats between two of its rows
    const float* src = (isG ? g.G : g.A) + (g.frag ? frag_index(ms + 4 * rg, col0, ld) : (ms + 4 * rg) * ld + col0);
    const float mul = isG ? gscale : 1.0f;
    const int wbase = (isG ? 2 : 0) * kPl + 4 * cg * kHColStride + rg * 8;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x16 acc[KTL][2];
#pragma unroll
    for (int a = 0; a < KTL; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f;      // column sums of the raw G block (bias gradient)

    // Register stages of operand rows (NERF_ATBH_STAGES, default 3): with one 512-thread workgroup per CU a 16-row step
    // lasts ~800 cycles, less than a global load's latency; loads are issued NS - 2 barriers ahead of their use.  The
    // loop is branch-free (see gemm_atb_f16_kernel: loads inside conditional blocks made the compiler wait for vmcnt(0)
    // before every LDS store); steps past the end read the last rows again and are zeroed when they are parked.
#ifndef NERF_ATBH_STAGES
#define NERF_ATBH_STAGES 3
#endif
    constexpr int NS = NERF_ATBH_STAGES;
    float4 R[NS][4];
    const long long steps = ms < me ? (me - ms) / 16 : 0;
    auto fetch = [&](float4 (&r)[4], long long st) {
        const long long sc = st < steps ? st : steps - 1;
        const float* q_ = src + (g.frag ? (size_t)(sc >> 1) * 32 * ld + (sc & 1) * 64 : (size_t)sc * 16 * ld);
        r[0] = *reinterpret_cast<const float4*>(q_);
        r[1] = *reinterpret_cast<const float4*>(q_ + rs);
        r[2] = *reinterpret_cast<const float4*>(q_ + 2 * (size_t)rs);
        r[3] = *reinterpret_cast<const float4*>(q_ + 3 * (size_t)rs);
    };
    auto col = [&](int jcol, float c0, float c1, float c2, float c3, int buf) {
        uint32_t h01, l01, h23, l23;
        split_pack2(c0 * mul, c1 * mul, h01, l01);
        split_pack2(c2 * mul, c3 * mul, h23, l23);
        unsigned char* w_ = &lds[buf][wbase + jcol * kHColStride];
        *reinterpret_cast<uint2*>(w_) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(w_ + kPl) = make_uint2(l01, l23);
    };
    auto park = [&](const float4 (&r_)[4], long long st, int buf) {
        const bool live = on && st < steps;
        float4 r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = live ? r_[e] : zero4;
        cs0 += (r[0].x + r[1].x) + (r[2].x + r[3].x); cs1 += (r[0].y + r[1].y) + (r[2].y + r[3].y);
        cs2 += (r[0].z + r[1].z) + (r[2].z + r[3].z); cs3 += (r[0].w + r[1].w) + (r[2].w + r[3].w);
        col(0, r[0].x, r[1].x, r[2].x, r[3].x, buf);
        col(1, r[0].y, r[1].y, r[2].y, r[3].y, buf);
        col(2, r[0].z, r[1].z, r[2].z, r[3].z, buf);
        col(3, r[0].w, r[1].w, r[2].w, r[3].w, buf);
    };

    auto compute = [&](int buf) {
        const unsigned char* base = lds[buf];
        h8v ah[KTL], al[KTL], gh[2], gl[2];
#pragma unroll
        for (int q = 0; q < KTL; ++q) {
            const int ca = (wk * (W / 2) + q * 32 + li) * kHColStride + 16 * lh;
            ah[q] = *reinterpret_cast<const h8v*>(base + ca);
            al[q] = *reinterpret_cast<const h8v*>(base + kPl + ca);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int cgd = (wn * 64 + q * 32 + li) * kHColStride + 16 * lh;
            gh[q] = *reinterpret_cast<const h8v*>(base + 2 * kPl + cgd);
            gl[q] = *reinterpret_cast<const h8v*>(base + 3 * kPl + cgd);
        }
#pragma unroll
        for (int a = 0; a < KTL; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], gl[c], acc[a][c], 0, 0, 0);
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], gh[c], acc[a][c], 0, 0, 0);
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], gh[c], acc[a][c], 0, 0, 0);
            }
    };

#ifndef NERF_ATB_STAGGER
#define NERF_ATB_STAGGER 1
#endif
    const bool late_half = NERF_ATB_STAGGER && W == 256 && __builtin_amdgcn_readfirstlane(wave) >= 4;
    if (steps > 0) {
#pragma unroll
        for (int q = 0; q < NS - 1; ++q) fetch(R[q], q);
        park(R[0], 0, 0);
        __syncthreads();
        int buf = 0;
        // step st: registers in slot st % NS; unrolled by NS for static register indices (the last round may run past the
        // end: those steps park zeros).
        // Stagger (NERF_ATB_STAGGER): waves w and w + 4 of a 512-thread workgroup share a SIMD and, running the same
        // program between the same barriers, did their MFMAs together and their staging (vector work) together.  The
        // second half parks FIRST and computes after -- park writes buffer buf ^ 1, compute reads buffer buf, both orders
        // are legal between two barriers -- so one partner's MFMAs run beside the other's split / pack / LDS stores
        // (MI355X_MICROARCH.md, "try a stagger").  Two copies of the whole loop, chosen once per wave: the choice inside
        // the loop body cost 400 spilled registers.
        auto main_loop = [&](auto late_c) {
            for (long long base_st = 1; base_st < steps; base_st += NS) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const long long st = base_st + i;
                    fetch(R[(1 + i + NS - 2) % NS], st + NS - 2);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (decltype(late_c)::value) {
                        park(R[(1 + i) % NS], st, buf ^ 1);
                        __builtin_amdgcn_sched_barrier(0);
                        compute(buf);
                    } else {
                        compute(buf);
                        __builtin_amdgcn_sched_barrier(0);
                        park(R[(1 + i) % NS], st, buf ^ 1);
                    }
                    __syncthreads();
                    buf ^= 1;
                }
            }
        };
        if (late_half) main_loop(std::true_type{});
        else main_loop(std::false_type{});
        compute(buf);
    }

    float* part = g.partial + (size_t)split * (g.Kp + 1) * g.Nw;
#pragma unroll
    for (int a = 0; a < KTL; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n = nb + wn * 64 + c * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kb + wk * (W / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (k < g.Kp && n < g.Nw) part[(size_t)k * g.Nw + n] = acc[a][c][r] * ginv;
            }
        }
    // bias gradient: column sums of the raw (unscaled) G rows of this slab; the four row groups of a column block are
    // four neighbouring lanes
    cs0 += __shfl_xor(cs0, 1); cs1 += __shfl_xor(cs1, 1); cs2 += __shfl_xor(cs2, 1); cs3 += __shfl_xor(cs3, 1);
    cs0 += __shfl_xor(cs0, 2); cs1 += __shfl_xor(cs1, 2); cs2 += __shfl_xor(cs2, 2); cs3 += __shfl_xor(cs3, 2);
    if (first_ktile && isG && rg == 0 && nb + 4 * cg < g.Nw) {
        float* prow = part + (size_t)g.Kp * g.Nw + nb + 4 * cg;
        prow[0] = cs0; prow[1] = cs1; prow[2] = cs2; prow[3] = cs3;
    }
}

// workgroup ranges of a batch: entry e gets tiles x splits workgroups, rounded up to a multiple of 8 so that every entry
// starts on XCD 0 (the tile order inside an entry is XCD-aware)
static int batch_ranges(GemmAtbBatch& b, int W) {
    int end = 0;
    for (int e = 0; e < b.n; ++e) {
        const GemmAtb& g = b.e[e];
        const int splits = (int)((g.M + g.rows_per_split - 1) / g.rows_per_split);
        const int tiles = ((g.Kp + W - 1) / W) * ((g.Nw + W - 1) / W);
        end += (tiles * splits + 7) / 8 * 8;
        b.wg_end[e] = end;
    }
    return end;
}

void launch_gemm_atb_h_batch(GemmAtbBatch& b, hipStream_t s, bool wide) {
    if (b.n <= 0) return;
    const int wgs = batch_ranges(b, wide ? 256 : 128);
    if (wide) hipLaunchKernelGGL(gemm_atb_h_kernel<256>, dim3((unsigned)wgs), dim3(512), 0, s, b);
    else hipLaunchKernelGGL(gemm_atb_h_kernel<128>, dim3((unsigned)wgs), dim3(256), 0, s, b);
}

void launch_gemm_atb_h(const GemmAtb& g, hipStream_t s, bool wide) {
    GemmAtbBatch b{};
    b.n = 1; b.e[0] = g;
    launch_gemm_atb_h_batch(b, s, wide);
}

// ------------------------------------------------------------------------------------------------
// gemm_atb_p: gemm_atb_h for a gradient operand that ARRIVES split ("pair16" buffers, float32 policy of the fused
// trainer, nerf_kernels.h::kPair16).  The backward chain holds every pre-activation gradient as an fp16 (hi, lo) pair --
// it is the next layer's MFMA operand -- and stores that pair in the value's fp32 slot: the 16 bytes of four consecutive
// features of a row are {hi01, hi23, lo01, lo23} (packed halfs); same buffers, same addresses, same loads.
//   A (activations): fp32 rows, split hi + lo while they are staged exactly as in gemm_atb_h (a pair16 stash was built
// too and measured out: the forward had to assemble its store tuple with moves and spilled, +5 % on that kernel).
//   G: the chain's packed operand D' = D * 2^s_row, one power of two PER SAMPLE ROW (mlp_bwd_f16x3.hip: every row's
// operand peaks between 2^5 and 2^12 whatever its gradient is), with 2^-s_row stored per row as the upper half of its
// fp32 bits (g_rs, 2 B per row and buffer).  The contraction runs over rows, so the row factor is applied while staging:
// f_row = 2^-s_row * gscale (gscale: the buffer's true max|D| -> [2^8, 2^9), from the producer's gmax slots as before),
// packed to fp16 pairs and multiplied onto the transposed (row, row + 1) pairs with v_pk_mul_f16 -- exact unless the
// product drops below 2^-14 (absolute error 2^-25 against a buffer maximum of 2^8: 2^-33 of max|D|); rows whose factor
// underflows fp16 altogether (peak below 2^-20 of the largest row's) are dropped, which an fp32 sum over the rows does to
// them as well.  The target 2^8 instead of gemm_atb_h's 2^14 leaves room for a row whose outputs collapsed against its
// operand's peak (factor up to 2^15 / its peak); beyond that the factor is clamped (finite, wrong by the clamp -- needs
// a row-wise gain below 2^-11 in one layer).  Rows past the slab's end take factor 0.  The bias gradient (column sums of
// true G) is v_dot2c_f32_f16 of the scaled pairs with (1, 1).  G's staging is ~3.5 vector instructions per element
// (byte permutes, v_pk_mul, v_dot2c) where the fp32 rows took ~6 (scale, and, subtract, conversions, column sums).
// Waves are specialised by operand (A: waves 0 .. W/64 - 1, G: the rest): two copies of the loop, chosen once per wave --
// for W = 256 these are the SIMD partners of gemm_atb_h's stagger (the G half stages first, then computes).
// What this kernel is bound by (profiles/r4_diagnostic_ab.txt): not its staging work -- with ONE MFMA pass instead of three
// (timing-only build) the batched launch streams at 5.96 TB/s, with three at 4.95: per 16-row step ~2000 cycles of
// barrier + LDS operand reads + staging do not overlap the 1536 MFMA cycles of the SIMD's two waves.
// ------------------------------------------------------------------------------------------------
// SIG: the entry carries the sigma head's weight gradient as a by-product (GemmAtb::sig_g; W = 128, one n tile)
template <int W, bool SIG = false>
__global__ __launch_bounds__(2 * W) __attribute__((amdgpu_waves_per_eu(2, W == 128 ? 3 : 2))) void gemm_atb_p_kernel(const GemmAtbBatch bat) {
    int lin = blockIdx.x;
    const GemmAtb& g = bat.e[batch_entry(bat, lin)];
    constexpr int kPl = W * kHColStride;          // bytes per plane
    constexpr int WNW = W / 64;                   // waves along n (2 or 4); two along k
    constexpr int KTL = W / 64;                   // 32-row k tiles per wave (2 or 4); two n tiles per wave
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][4 * kPl];    // planes: A hi, A lo, G hi, G lo
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = wave / WNW, wn = wave % WNW;
    const int li = lane & 31, lh = lane >> 5;
    const int kt_n = (g.Kp + W - 1) / W, nt_n = (g.Nw + W - 1) / W, T = kt_n * nt_n;
    const int n_splits = (int)((g.M + g.rows_per_split - 1) / g.rows_per_split);
    if (lin >= T * n_splits) return;              // padding workgroups of a batch entry (its range is a multiple of 8)
    const int grp = lin / (8 * T), rem = lin % (8 * T);
    int split = grp * 8 + rem % 8, tile = rem / 8;
    if (grp * 8 + 8 > n_splits) {
        const int r2 = lin - grp * 8 * T, left = n_splits - grp * 8;
        split = grp * 8 + r2 % left;
        tile = r2 / left;
    }
    const int kb = (tile % kt_n) * W, nb = (tile / kt_n) * W;
    const bool first_ktile = tile % kt_n == 0;
    const long long ms = (long long)split * g.rows_per_split;
    const long long me = ms + g.rows_per_split < g.M ? ms + g.rows_per_split : g.M;
    // power-of-two scale of G: largest entry -> [2^8, 2^9)
    unsigned mb = g.gmax ? g.gmax[lane] : 0u;                  // 64 slots per buffer (see the producers)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned other = __shfl_xor(mb, o); mb = other > mb ? other : mb; }
    int sexp = mb ? 127 + 8 - ((int)((mb >> 23) & 0xFF) - 127) : 127;
    sexp = sexp < 1 ? 1 : sexp > 254 ? 254 : sexp;
    const float gscale = __uint_as_float((unsigned)sexp << 23), ginv = 1.0f / gscale;
    // staging role of this thread: operand (A: t < W, G: t >= W), rows 4 rg .. 4 rg + 3, columns 4 cg .. 4 cg + 3
    const bool isG = __builtin_amdgcn_readfirstlane(wave) >= W / 64;
    const int b = t & (W - 1), rg = b & 3, cg = b >> 2;
    const int ld = isG ? g.ldg : g.lda;
    const bool on = isG ? (nb + 4 * cg < g.N) : (kb + 4 * cg < g.K);
    // threads whose columns lie outside the matrix read column block 0 (valid memory); what they stage only reaches
    // outputs beyond Kp / Nw, which are not stored
    const int col0 = on ? (isG ? nb : kb) + 4 * cg : 0;
    // fragment-major operands (frag_layout.h::frag_index): the thread's 4 rows x 4 columns are 64 consecutive bytes
    const uint4* src = reinterpret_cast<const uint4*>((isG ? g.G : g.A) + frag_index(ms + 4 * rg, col0, ld));
    const uint16_t* rsp = g.g_rs + ms + 4 * rg;
    const int wbase = (isG ? 2 : 0) * kPl + 4 * cg * kHColStride + rg * 8;

    f32x16 acc[KTL][2];
#pragma unroll
    for (int a = 0; a < KTL; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f;      // column sums of the scaled G block (bias gradient)
    float sg0 = 0.f, sg1 = 0.f, sg2 = 0.f, sg3 = 0.f, sgb = 0.f;   // SIG, A role: sum_rows A[row][col j] * d_sigma[row]; sum of d_sigma
    const float* sigp = SIG ? g.sig_g + ms + 4 * rg : nullptr;

#ifndef NERF_ATBP_STAGES
#define NERF_ATBP_STAGES 3
#endif
    constexpr int NS = NERF_ATBP_STAGES;
    uint4 R[NS][4];
    uint2 S[NS];                                            // G role: the four rows' factors (upper halves of fp32 bits)
    float4 SGv[SIG ? NS : 1];                               // SIG, A role: d_sigma of the four rows
    const long long steps = ms < me ? (me - ms) / 16 : 0;
    auto fetch = [&](uint4 (&r)[4], uint2& s, float4& sgv, long long st, auto role_g) {
        const long long sc = st < steps ? st : steps - 1;
        const uint4* q_ = src + ((size_t)(sc >> 1) * 32 * ld + (sc & 1) * 64) / 4;
        r[0] = q_[0]; r[1] = q_[1]; r[2] = q_[2]; r[3] = q_[3];
        if constexpr (decltype(role_g)::value) s = *reinterpret_cast<const uint2*>(rsp + sc * 16);
        else if constexpr (SIG) sgv = *reinterpret_cast<const float4*>(sigp + sc * 16);
    };
    // column j of the 4 x 4 block of a plane = halfs (row 0 .. row 3)[j]: two byte permutes (gemm_atb_f16_kernel::park1);
    // a row's 16 bytes are {hi01, hi23, lo01, lo23}
    auto park = [&](const uint4 (&r)[4], const uint2& s, const float4& sgv, long long st, int buf, auto role_g) {
        unsigned char* w_ = &lds[buf][wbase];
        constexpr uint32_t kLoSel = 0x05040100u, kHiSel = 0x07060302u;
        if constexpr (!decltype(role_g)::value) {
            // A: fp32 activations, split hi + lo here exactly as gemm_atb_h does (rows past the slab's end are re-read rows:
            // finite, and they meet a zero factor on the G side)
            auto cola = [&](int j, float c0, float c1, float c2, float c3) {
                uint32_t h01, l01, h23, l23;
                split_pack2(c0, c1, h01, l01);
                split_pack2(c2, c3, h23, l23);
                *reinterpret_cast<uint2*>(w_ + j * kHColStride) = make_uint2(h01, h23);
                *reinterpret_cast<uint2*>(w_ + kPl + j * kHColStride) = make_uint2(l01, l23);
            };
            cola(0, __uint_as_float(r[0].x), __uint_as_float(r[1].x), __uint_as_float(r[2].x), __uint_as_float(r[3].x));
            cola(1, __uint_as_float(r[0].y), __uint_as_float(r[1].y), __uint_as_float(r[2].y), __uint_as_float(r[3].y));
            cola(2, __uint_as_float(r[0].z), __uint_as_float(r[1].z), __uint_as_float(r[2].z), __uint_as_float(r[3].z));
            cola(3, __uint_as_float(r[0].w), __uint_as_float(r[1].w), __uint_as_float(r[2].w), __uint_as_float(r[3].w));
            if constexpr (SIG) {        // (rows past the slab's end are re-read rows: their d_sigma counts as zero)
                const bool live = st < steps;
                const float d0 = live ? sgv.x : 0.f, d1 = live ? sgv.y : 0.f, d2 = live ? sgv.z : 0.f, d3 = live ? sgv.w : 0.f;
                sg0 = fmaf(__uint_as_float(r[3].x), d3, fmaf(__uint_as_float(r[2].x), d2, fmaf(__uint_as_float(r[1].x), d1, fmaf(__uint_as_float(r[0].x), d0, sg0))));
                sg1 = fmaf(__uint_as_float(r[3].y), d3, fmaf(__uint_as_float(r[2].y), d2, fmaf(__uint_as_float(r[1].y), d1, fmaf(__uint_as_float(r[0].y), d0, sg1))));
                sg2 = fmaf(__uint_as_float(r[3].z), d3, fmaf(__uint_as_float(r[2].z), d2, fmaf(__uint_as_float(r[1].z), d1, fmaf(__uint_as_float(r[0].z), d0, sg2))));
                sg3 = fmaf(__uint_as_float(r[3].w), d3, fmaf(__uint_as_float(r[2].w), d2, fmaf(__uint_as_float(r[1].w), d1, fmaf(__uint_as_float(r[0].w), d0, sg3))));
                sgb += (d0 + d1) + (d2 + d3);
            }
        } else {
            // f_row = 2^-s_row * gscale as fp16 pairs (rows 0,1 / rows 2,3); 0 for the steps past the slab's end
            const bool live = st < steps;
            const float kMaxF = 32768.0f;
            const float f0 = fminf(__uint_as_float(s.x << 16) * gscale, kMaxF), f1 = fminf(__uint_as_float(s.x & 0xFFFF0000u) * gscale, kMaxF);
            const float f2 = fminf(__uint_as_float(s.y << 16) * gscale, kMaxF), f3 = fminf(__uint_as_float(s.y & 0xFFFF0000u) * gscale, kMaxF);
            const h2v f01 = {(_Float16)(live ? f0 : 0.f), (_Float16)(live ? f1 : 0.f)};
            const h2v f23 = {(_Float16)(live ? f2 : 0.f), (_Float16)(live ? f3 : 0.f)};
            const h2v one2 = {(_Float16)1.0f, (_Float16)1.0f};
            auto colg = [&](int j, uint32_t a01, uint32_t a23, uint32_t l01, uint32_t l23, float& cs) {
                const h2v h_a = __builtin_bit_cast(h2v, a01) * f01, h_b = __builtin_bit_cast(h2v, a23) * f23;
                const h2v l_a = __builtin_bit_cast(h2v, l01) * f01, l_b = __builtin_bit_cast(h2v, l23) * f23;
                cs = __builtin_amdgcn_fdot2(h_a, one2, cs, false);
                cs = __builtin_amdgcn_fdot2(h_b, one2, cs, false);
                cs = __builtin_amdgcn_fdot2(l_a, one2, cs, false);
                cs = __builtin_amdgcn_fdot2(l_b, one2, cs, false);
                *reinterpret_cast<uint2*>(w_ + j * kHColStride) = make_uint2(__builtin_bit_cast(uint32_t, h_a), __builtin_bit_cast(uint32_t, h_b));
                *reinterpret_cast<uint2*>(w_ + kPl + j * kHColStride) = make_uint2(__builtin_bit_cast(uint32_t, l_a), __builtin_bit_cast(uint32_t, l_b));
            };
#define NERF_ATBP_G(J, X, Z, SEL, CS)                                                                                    \
            colg(J, __builtin_amdgcn_perm(r[1].X, r[0].X, SEL), __builtin_amdgcn_perm(r[3].X, r[2].X, SEL),                \
                 __builtin_amdgcn_perm(r[1].Z, r[0].Z, SEL), __builtin_amdgcn_perm(r[3].Z, r[2].Z, SEL), CS);
            NERF_ATBP_G(0, x, z, kLoSel, cs0) NERF_ATBP_G(1, x, z, kHiSel, cs1) NERF_ATBP_G(2, y, w, kLoSel, cs2) NERF_ATBP_G(3, y, w, kHiSel, cs3)
#undef NERF_ATBP_G
        }
    };

    auto compute = [&](int buf) {
        const unsigned char* base = lds[buf];
        h8v ah[KTL], al[KTL], gh[2], gl[2];
#pragma unroll
        for (int q = 0; q < KTL; ++q) {
            const int ca = (wk * (W / 2) + q * 32 + li) * kHColStride + 16 * lh;
            ah[q] = *reinterpret_cast<const h8v*>(base + ca);
            al[q] = *reinterpret_cast<const h8v*>(base + kPl + ca);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int cgd = (wn * 64 + q * 32 + li) * kHColStride + 16 * lh;
            gh[q] = *reinterpret_cast<const h8v*>(base + 2 * kPl + cgd);
            gl[q] = *reinterpret_cast<const h8v*>(base + 3 * kPl + cgd);
        }
#pragma unroll
        for (int a = 0; a < KTL; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#ifndef NERF_DIAG_ATBP_1PASS     // timing-only diagnostic (wrong results): one MFMA pass instead of three
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], gl[c], acc[a][c], 0, 0, 0);
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], gh[c], acc[a][c], 0, 0, 0);
#endif
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], gh[c], acc[a][c], 0, 0, 0);
            }
    };

    if (steps > 0) {
        // Two copies of the whole pipeline, chosen once per wave (the choice inside the loop body cost hundreds of spilled
        // registers in gemm_atb_h).  The G waves of the 256-wide tile stage FIRST and compute after (gemm_atb_h's stagger of
        // the SIMD partners w / w + 4: park writes buffer buf ^ 1, compute reads buf, both orders are legal between barriers).
        auto run = [&](auto role_g) {
            constexpr bool late = decltype(role_g)::value && NERF_ATB_STAGGER && W == 256;
#pragma unroll
            for (int q = 0; q < NS - 1; ++q) fetch(R[q], S[q], SGv[SIG ? q : 0], q, role_g);
            park(R[0], S[0], SGv[0], 0, 0, role_g);
            __syncthreads();
            int buf = 0;
            for (long long base_st = 1; base_st < steps; base_st += NS) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const long long st = base_st + i;
                    fetch(R[(1 + i + NS - 2) % NS], S[(1 + i + NS - 2) % NS], SGv[SIG ? (1 + i + NS - 2) % NS : 0], st + NS - 2, role_g);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (late) {
                        park(R[(1 + i) % NS], S[(1 + i) % NS], SGv[SIG ? (1 + i) % NS : 0], st, buf ^ 1, role_g);
                        __builtin_amdgcn_sched_barrier(0);
                        compute(buf);
                    } else {
                        compute(buf);
                        __builtin_amdgcn_sched_barrier(0);
                        park(R[(1 + i) % NS], S[(1 + i) % NS], SGv[SIG ? (1 + i) % NS : 0], st, buf ^ 1, role_g);
                    }
                    __syncthreads();
                    buf ^= 1;
                }
            }
            compute(buf);
        };
        if (isG) run(std::true_type{});
        else run(std::false_type{});
    }

    float* part = g.partial + (size_t)split * (g.Kp + 1) * g.Nw;
#pragma unroll
    for (int a = 0; a < KTL; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n = nb + wn * 64 + c * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kb + wk * (W / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (k < g.Kp && n < g.Nw) part[(size_t)k * g.Nw + n] = acc[a][c][r] * ginv;
            }
        }
    // bias gradient: column sums of the G rows of this slab (in G's staged scale); the four row groups of a column block
    // are four neighbouring lanes
    cs0 += __shfl_xor(cs0, 1); cs1 += __shfl_xor(cs1, 1); cs2 += __shfl_xor(cs2, 1); cs3 += __shfl_xor(cs3, 1);
    cs0 += __shfl_xor(cs0, 2); cs1 += __shfl_xor(cs1, 2); cs2 += __shfl_xor(cs2, 2); cs3 += __shfl_xor(cs3, 2);
    if (first_ktile && isG && rg == 0 && nb + 4 * cg < g.Nw) {
        float* prow = part + (size_t)g.Kp * g.Nw + nb + 4 * cg;
        prow[0] = cs0 * ginv; prow[1] = cs1 * ginv; prow[2] = cs2 * ginv; prow[3] = cs3 * ginv;
    }
    if constexpr (SIG) {     // the sigma head's gradient: the four row groups of a column block are four neighbouring lanes
        sg0 += __shfl_xor(sg0, 1); sg1 += __shfl_xor(sg1, 1); sg2 += __shfl_xor(sg2, 1); sg3 += __shfl_xor(sg3, 1); sgb += __shfl_xor(sgb, 1);
        sg0 += __shfl_xor(sg0, 2); sg1 += __shfl_xor(sg1, 2); sg2 += __shfl_xor(sg2, 2); sg3 += __shfl_xor(sg3, 2); sgb += __shfl_xor(sgb, 2);
        if (!isG && rg == 0 && nb == 0) {
            float* sp = g.sig_partial + (size_t)split * (g.Kp + 1);
            const int k0 = kb + 4 * cg;
            if (k0 + 0 < g.Kp) sp[k0 + 0] = sg0;
            if (k0 + 1 < g.Kp) sp[k0 + 1] = sg1;
            if (k0 + 2 < g.Kp) sp[k0 + 2] = sg2;
            if (k0 + 3 < g.Kp) sp[k0 + 3] = sg3;
            if (first_ktile && cg == 0) sp[g.Kp] = sgb;
        }
    }
}

void launch_gemm_atb_p_batch(GemmAtbBatch& b, hipStream_t s, bool wide) {
    if (b.n <= 0) return;
    const int wgs = batch_ranges(b, wide ? 256 : 128);
    if (wide) hipLaunchKernelGGL(gemm_atb_p_kernel<256>, dim3((unsigned)wgs), dim3(512), 0, s, b);
    else if (b.n == 1 && b.e[0].sig_g && b.e[0].Nw <= 128) hipLaunchKernelGGL((gemm_atb_p_kernel<128, true>), dim3((unsigned)wgs), dim3(256), 0, s, b);
    else hipLaunchKernelGGL(gemm_atb_p_kernel<128>, dim3((unsigned)wgs), dim3(256), 0, s, b);
}

void launch_gemm_atb_p(const GemmAtb& g, hipStream_t s, bool wide) {
    GemmAtbBatch b{};
    b.n = 1; b.e[0] = g;
    launch_gemm_atb_p_batch(b, s, wide);
}

// ------------------------------------------------------------------------------------------------
// gemm_atb_f16: the weight-gradient GEMM of the mixed_float16 policy.  A (activations) and G (loss-scaled pre-activation
// gradients) arrive as fp16 rows from the single-pass stash forward / backward chain: half the bytes of the fp32 buffers
// (this GEMM is bound by reading them), no hi/lo split and no conversion while staging -- a 4 x 4 block of halfs is
// transposed with byte permutes -- and ONE v_mfma_f32_32x32x16_f16 pass per product, fp32 accumulation.  Same tiling,
// LDS layout ([plane][column][16 rows], 48-byte column stride), partial-sum layout and bias row as gemm_atb_h.  A
// gradient that left the fp16 range arrives as Inf and makes the partial sums non-finite: the loss-scale logic skips
// that step (src/NeRF.py:159-163 under LossScaleOptimizer).
// ------------------------------------------------------------------------------------------------
// SIG: the entry carries the sigma head's weight gradient as a by-product (GemmAtb::sig_g; W = 128, one n tile): fp16 A x fp32
// d_sigma, accumulated in fp32 (v_fma_mix_f32: no conversion instructions)
template <int W, bool FRAG, bool SIG = false>
__global__ __launch_bounds__(2 * W) __attribute__((amdgpu_waves_per_eu(2, W == 128 && !SIG ? 3 : 2))) void gemm_atb_f16_kernel(const GemmAtbBatch bat) {
    int lin = blockIdx.x;
    const GemmAtb& g = bat.e[batch_entry(bat, lin)];
    constexpr int kPl = W * kHColStride;
    constexpr int WNW = W / 64;
    constexpr int KTL = W / 64;
#ifndef NERF_ATBF_PB
#define NERF_ATBF_PB 2
#endif
    constexpr int PB = NERF_ATBF_PB;      // 16-row sub-steps per barrier
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][PB][2 * kPl];    // planes: A, G
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = wave / WNW, wn = wave % WNW;
    const int li = lane & 31, lh = lane >> 5;
    const int kt_n = (g.Kp + W - 1) / W, nt_n = (g.Nw + W - 1) / W, T = kt_n * nt_n;
    const int n_splits = (int)((g.M + g.rows_per_split - 1) / g.rows_per_split);
    if (lin >= T * n_splits) return;              // padding workgroups of a batch entry
    const int grp = lin / (8 * T), rem = lin % (8 * T);
    int split = grp * 8 + rem % 8, tile = rem / 8;
    if (grp * 8 + 8 > n_splits) {
        const int r2 = lin - grp * 8 * T, left = n_splits - grp * 8;
        split = grp * 8 + r2 % left;
        tile = r2 / left;
    }
    const int kb = (tile % kt_n) * W, nb = (tile / kt_n) * W;
    const bool first_ktile = tile % kt_n == 0;
    const long long ms = (long long)split * g.rows_per_split;
    const long long me = ms + g.rows_per_split < g.M ? ms + g.rows_per_split : g.M;
    const bool isG = t >= W;
    const int b = t & (W - 1), rg = b & 3, cg = b >> 2;
    const int ld = isG ? g.ldg : g.lda;
    const bool on = isG ? (nb + 4 * cg < g.N) : (kb + 4 * cg < g.K);
    // threads whose columns lie outside the matrix read column block 0 (valid memory) and park zeros
    const int col = on ? (isG ? nb : kb) + 4 * cg : 0;
    float sg0 = 0.f, sg1 = 0.f, sg2 = 0.f, sg3 = 0.f, sgb = 0.f;   // SIG, A threads: sum_rows A[row][col j] * d_sigma[row]; sum of d_sigma
    const float* sigp = SIG ? g.sig_g + ms + 4 * rg : nullptr;
    // fragment-major operands (frag_layout.h::frag_index): the thread's 4 rows x 4 columns are 16 consecutive elements
    // -- with FRAG known at compile time they are fetched as two 16-byte loads instead of four 8-byte ones
    const int rs = FRAG ? 4 : ld;                                     // elements between two of its rows
    const uint16_t* src = reinterpret_cast<const uint16_t*>(isG ? g.G : g.A) +
                          (FRAG ? frag_index(ms + 4 * rg, col, ld) : (ms + 4 * rg) * ld + col);
    const int wbase = (isG ? 1 : 0) * kPl + 4 * cg * kHColStride + rg * 8;

    f32x16 acc[KTL][2];
#pragma unroll
    for (int a = 0; a < KTL; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
    float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f;

    // Register stages of operand rows, each PB sub-steps, loaded NS - 2 barriers ahead of their use.  The loop is
    // branch-free on purpose: with the loads inside "if (in range)" blocks the compiler's wait-count bookkeeping gave up
    // at the joins and put s_waitcnt vmcnt(0) in front of every LDS store, i.e. it also waited for the loads it had just
    // issued, and the kernel ran at one global-load latency per step (3.1 TB/s of HBM traffic; 4.6 TB/s without the branches).  Out-of-range sub-steps read the last
    // valid rows again and are zeroed when they are parked.
#ifndef NERF_ATBF_STAGES
#define NERF_ATBF_STAGES 4
#endif
#ifndef NERF_ATBF_STAGES_WIDE
#define NERF_ATBF_STAGES_WIDE 3
#endif
    constexpr int NS = W == 256 ? NERF_ATBF_STAGES_WIDE : NERF_ATBF_STAGES;   // the 256-wide tile has 128 accumulator registers
    uint2 R[NS][PB][4];
    float4 SGv[SIG ? NS : 1][PB];      // SIG: d_sigma of the thread's four rows per sub-step (loaded by every thread: no load sits in a branch)
    auto h2f = [](uint32_t w, int hi) -> float {
        const h2v v = __builtin_bit_cast(h2v, w);
        return (float)v[hi];
    };
    long long steps = 0;
    auto fetch1 = [&](uint2 (&r)[4], long long st) {
        const long long sc = st < steps ? st : steps - 1;
        const uint16_t* q_ = src + (FRAG ? (size_t)(sc >> 1) * 32 * ld + (sc & 1) * 64 : (size_t)sc * 16 * ld);
        if constexpr (FRAG) {
            const uint4 lo = *reinterpret_cast<const uint4*>(q_), hi = *reinterpret_cast<const uint4*>(q_ + 8);
            r[0] = make_uint2(lo.x, lo.y); r[1] = make_uint2(lo.z, lo.w);
            r[2] = make_uint2(hi.x, hi.y); r[3] = make_uint2(hi.z, hi.w);
        } else {
            r[0] = *reinterpret_cast<const uint2*>(q_);
            r[1] = *reinterpret_cast<const uint2*>(q_ + rs);
            r[2] = *reinterpret_cast<const uint2*>(q_ + 2 * (size_t)rs);
            r[3] = *reinterpret_cast<const uint2*>(q_ + 3 * (size_t)rs);
        }
    };
    auto fetch = [&](uint2 (&r)[PB][4], float4 (&sgv)[PB], long long pj) {
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            fetch1(r[p], pj * PB + p);
            if constexpr (SIG) {
                const long long st = pj * PB + p, sc = st < steps ? st : steps - 1;
                sgv[p] = *reinterpret_cast<const float4*>(sigp + sc * 16);
            }
        }
    };
    // column j of the 4 x 4 block = halfs (r0, r1, r2, r3)[j]: two byte permutes per column
    auto park1 = [&](const uint2 (&r_)[4], bool live, unsigned char* lbuf) {
        uint2 r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = make_uint2(live ? r_[e].x : 0u, live ? r_[e].y : 0u);
        if (isG && first_ktile) {
            cs0 += (h2f(r[0].x, 0) + h2f(r[1].x, 0)) + (h2f(r[2].x, 0) + h2f(r[3].x, 0));
            cs1 += (h2f(r[0].x, 1) + h2f(r[1].x, 1)) + (h2f(r[2].x, 1) + h2f(r[3].x, 1));
            cs2 += (h2f(r[0].y, 0) + h2f(r[1].y, 0)) + (h2f(r[2].y, 0) + h2f(r[3].y, 0));
            cs3 += (h2f(r[0].y, 1) + h2f(r[1].y, 1)) + (h2f(r[2].y, 1) + h2f(r[3].y, 1));
        }
        unsigned char* w_ = lbuf + wbase;
        *reinterpret_cast<uint2*>(w_ + 0 * kHColStride) = make_uint2(__builtin_amdgcn_perm(r[1].x, r[0].x, 0x05040100u), __builtin_amdgcn_perm(r[3].x, r[2].x, 0x05040100u));
        *reinterpret_cast<uint2*>(w_ + 1 * kHColStride) = make_uint2(__builtin_amdgcn_perm(r[1].x, r[0].x, 0x07060302u), __builtin_amdgcn_perm(r[3].x, r[2].x, 0x07060302u));
        *reinterpret_cast<uint2*>(w_ + 2 * kHColStride) = make_uint2(__builtin_amdgcn_perm(r[1].y, r[0].y, 0x05040100u), __builtin_amdgcn_perm(r[3].y, r[2].y, 0x05040100u));
        *reinterpret_cast<uint2*>(w_ + 3 * kHColStride) = make_uint2(__builtin_amdgcn_perm(r[1].y, r[0].y, 0x07060302u), __builtin_amdgcn_perm(r[3].y, r[2].y, 0x07060302u));
    };

    auto half_of = [](uint32_t w, int hi) -> _Float16 { return __builtin_bit_cast(h2v, w)[hi]; };
    auto park = [&](const uint2 (&r)[PB][4], const float4 (&sgv)[PB], long long pj, int buf) {
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            park1(r[p], on && pj * PB + p < steps, lds[buf][p]);
            if constexpr (SIG) {
                if (!isG) {      // (wave-uniform: the A threads are whole waves)
                    const bool live = pj * PB + p < steps;
                    const float d[4] = {live ? sgv[p].x : 0.f, live ? sgv[p].y : 0.f, live ? sgv[p].z : 0.f, live ? sgv[p].w : 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sg0 = fmaf((float)half_of(r[p][e].x, 0), d[e], sg0);
                        sg1 = fmaf((float)half_of(r[p][e].x, 1), d[e], sg1);
                        sg2 = fmaf((float)half_of(r[p][e].y, 0), d[e], sg2);
                        sg3 = fmaf((float)half_of(r[p][e].y, 1), d[e], sg3);
                    }
                    sgb += (d[0] + d[1]) + (d[2] + d[3]);
                }
            }
        }
    };

    auto compute1 = [&](const unsigned char* base) {
        h8v ah[KTL], gh[2];
#pragma unroll
        for (int q = 0; q < KTL; ++q)
            ah[q] = *reinterpret_cast<const h8v*>(base + (wk * (W / 2) + q * 32 + li) * kHColStride + 16 * lh);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            gh[q] = *reinterpret_cast<const h8v*>(base + kPl + (wn * 64 + q * 32 + li) * kHColStride + 16 * lh);
#pragma unroll
        for (int a = 0; a < KTL; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], gh[c], acc[a][c], 0, 0, 0);
    };

    auto compute = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PB; ++p) compute1(lds[buf][p]);
    };

    steps = ms < me ? (me - ms) / 16 : 0;
    const bool late_half = NERF_ATB_STAGGER && W == 256 && __builtin_amdgcn_readfirstlane(wave) >= 4;
    if (steps > 0) {
        const long long groups = (steps + PB - 1) / PB;
#pragma unroll
        for (int q = 0; q < NS - 1; ++q) fetch(R[q], SGv[SIG ? q : 0], q);
        park(R[0], SGv[0], 0, 0);
        __syncthreads();
        int buf = 0;
        // group st: registers in slot st % NS; unrolled by NS for static register indices.  The last round may run past
        // the end: those groups park zeros.  (Stagger of the SIMD partners: see gemm_atb_h_kernel.)
        auto main_loop = [&](auto late_c) {
            for (long long base_st = 1; base_st < groups; base_st += NS) {
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const long long st = base_st + i;
                    fetch(R[(1 + i + NS - 2) % NS], SGv[SIG ? (1 + i + NS - 2) % NS : 0], st + NS - 2);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (decltype(late_c)::value) {
                        park(R[(1 + i) % NS], SGv[SIG ? (1 + i) % NS : 0], st, buf ^ 1);
                        __builtin_amdgcn_sched_barrier(0);
                        compute(buf);
                    } else {
                        compute(buf);
                        __builtin_amdgcn_sched_barrier(0);
                        park(R[(1 + i) % NS], SGv[SIG ? (1 + i) % NS : 0], st, buf ^ 1);
                    }
                    __syncthreads();
                    buf ^= 1;
                }
            }
        };
        if (late_half) main_loop(std::true_type{});
        else main_loop(std::false_type{});
        compute(buf);
    }

    float* part = g.partial + (size_t)split * (g.Kp + 1) * g.Nw;
#pragma unroll
    for (int a = 0; a < KTL; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n = nb + wn * 64 + c * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = kb + wk * (W / 2) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (k < g.Kp && n < g.Nw) part[(size_t)k * g.Nw + n] = acc[a][c][r];
            }
        }
    cs0 += __shfl_xor(cs0, 1); cs1 += __shfl_xor(cs1, 1); cs2 += __shfl_xor(cs2, 1); cs3 += __shfl_xor(cs3, 1);
    cs0 += __shfl_xor(cs0, 2); cs1 += __shfl_xor(cs1, 2); cs2 += __shfl_xor(cs2, 2); cs3 += __shfl_xor(cs3, 2);
    if (first_ktile && isG && rg == 0 && nb + 4 * cg < g.Nw) {
        float* prow = part + (size_t)g.Kp * g.Nw + nb + 4 * cg;
        prow[0] = cs0; prow[1] = cs1; prow[2] = cs2; prow[3] = cs3;
    }
    if constexpr (SIG) {     // the sigma head's gradient: the four row groups of a column block are four neighbouring lanes
        sg0 += __shfl_xor(sg0, 1); sg1 += __shfl_xor(sg1, 1); sg2 += __shfl_xor(sg2, 1); sg3 += __shfl_xor(sg3, 1); sgb += __shfl_xor(sgb, 1);
        sg0 += __shfl_xor(sg0, 2); sg1 += __shfl_xor(sg1, 2); sg2 += __shfl_xor(sg2, 2); sg3 += __shfl_xor(sg3, 2); sgb += __shfl_xor(sgb, 2);
        if (!isG && rg == 0 && nb == 0) {
            float* sp = g.sig_partial + (size_t)split * (g.Kp + 1);
            const int k0 = kb + 4 * cg;
            if (k0 + 0 < g.Kp) sp[k0 + 0] = sg0;
            if (k0 + 1 < g.Kp) sp[k0 + 1] = sg1;
            if (k0 + 2 < g.Kp) sp[k0 + 2] = sg2;
            if (k0 + 3 < g.Kp) sp[k0 + 3] = sg3;
            if (first_ktile && cg == 0) sp[g.Kp] = sgb;
        }
    }
}

void launch_gemm_atb_f16_batch(GemmAtbBatch& b, hipStream_t s, bool wide) {
    if (b.n <= 0) return;
    const int wgs = batch_ranges(b, wide ? 256 : 128);
    const bool frag = b.e[0].frag != 0;           // one layout per trainer: all entries agree
    if (wide) {
        if (frag) hipLaunchKernelGGL((gemm_atb_f16_kernel<256, true>), dim3((unsigned)wgs), dim3(512), 0, s, b);
        else hipLaunchKernelGGL((gemm_atb_f16_kernel<256, false>), dim3((unsigned)wgs), dim3(512), 0, s, b);
    } else {
        if (frag && b.n == 1 && b.e[0].sig_g && b.e[0].Nw <= 128) hipLaunchKernelGGL((gemm_atb_f16_kernel<128, true, true>), dim3((unsigned)wgs), dim3(256), 0, s, b);
        else if (frag) hipLaunchKernelGGL((gemm_atb_f16_kernel<128, true>), dim3((unsigned)wgs), dim3(256), 0, s, b);
        else hipLaunchKernelGGL((gemm_atb_f16_kernel<128, false>), dim3((unsigned)wgs), dim3(256), 0, s, b);
    }
}

void launch_gemm_atb_f16(const GemmAtb& g, hipStream_t s, bool wide) {
    GemmAtbBatch b{};
    b.n = 1; b.e[0] = g;
    launch_gemm_atb_f16_batch(b, s, wide);
}

// ------------------------------------------------------------------------------------------------
// gemm_abt_h: the data-gradient GEMM  G_prev = (G . W^T [+ rank-1]) * LeakyReLU'(H)  on the fp16 matrix cores, same
// arithmetic as gemm_atb_h: G is scaled to the fp16 range (its max comes from its producer) and split hi + lo while it is
// staged (rows are k-contiguous: no transpose), W comes pre-split from the relayout kernel; three passes of
// v_mfma_f32_32x32x16_f16, fp32 accumulation, then the usual epilogue on acc / scale.  k-step 32, 80-byte LDS rows.
// ------------------------------------------------------------------------------------------------
constexpr int kAbhStride = 80;
constexpr int kAbhPlane = 128 * kAbhStride;     // 10 240 B: 128 rows x (32 halfs + pad)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_abt_h_kernel(const GemmAbt g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][4 * kAbhPlane];   // planes: A hi, A lo, B hi, B lo
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n_tiles = g.N / 128;
    const long long lin = blockIdx.x, grp = lin / (8 * n_tiles), rem = lin % (8 * n_tiles);
    long long m_tile = grp * 8 + rem % 8;
    int n_tile = (int)(rem / 8);
    const long long m_tiles = g.M / 128;
    if (grp * 8 + 8 > m_tiles) {
        const long long base = grp * 8 * n_tiles, r2 = lin - base, left = m_tiles - grp * 8;
        m_tile = grp * 8 + r2 % left;
        n_tile = (int)(r2 / left);
    }
    const long long m0 = m_tile * 128;
    const int n0 = n_tile * 128;
    unsigned mb = g.gmax_in ? g.gmax_in[lane] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned other = __shfl_xor(mb, o); mb = other > mb ? other : mb; }
    int sexp = mb ? 127 + 14 - ((int)((mb >> 23) & 0xFF) - 127) : 127;
    sexp = sexp < 1 ? 1 : sexp > 254 ? 254 : sexp;
    const float gscale = __uint_as_float((unsigned)sexp << 23), ginv = 1.0f / gscale;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    float hmask[2][2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n = n0 + (wn * 2 + c) * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                hmask[a][c][r] = g.H[m * g.ldh + n];
            }
        }

    // staging slots: A tile 128 rows x 32 k fp32 = 1024 float4 (4 per thread); B planes 128 rows x 32 halfs = 512 x 16 B each
    const int arow = t >> 3, akq = t & 7;            // + 32 i rows
    const float* ap = g.A + (m0 + arow) * g.lda + 4 * akq;
    const size_t a32 = (size_t)32 * g.lda;
    const int brow = t >> 2, bq = t & 3;             // + 64 i rows
    const uint16_t* bph = g.Bhi + (size_t)(n0 + brow) * g.ldb + 8 * bq;
    const uint16_t* bpl = g.Blo + (size_t)(n0 + brow) * g.ldb + 8 * bq;
    const size_t b64 = (size_t)64 * g.ldb;
    float4 ra0, ra1, ra2, ra3;
    uint4 bh0, bh1, bl0, bl1;
#define ABH_FETCH(K0)                                                                         \
    ra0 = *reinterpret_cast<const float4*>(ap + (K0));                                        \
    ra1 = *reinterpret_cast<const float4*>(ap + a32 + (K0));                                  \
    ra2 = *reinterpret_cast<const float4*>(ap + 2 * a32 + (K0));                              \
    ra3 = *reinterpret_cast<const float4*>(ap + 3 * a32 + (K0));                              \
    bh0 = *reinterpret_cast<const uint4*>(bph + (K0));                                        \
    bh1 = *reinterpret_cast<const uint4*>(bph + b64 + (K0));                                  \
    bl0 = *reinterpret_cast<const uint4*>(bpl + (K0));                                        \
    bl1 = *reinterpret_cast<const uint4*>(bpl + b64 + (K0));
#define ABH_A(R, I, BUF)                                                                      \
    {                                                                                         \
        uint32_t h01, l01, h23, l23;                                                          \
        split_pack2((R).x * gscale, (R).y * gscale, h01, l01);                                \
        split_pack2((R).z * gscale, (R).w * gscale, h23, l23);                                \
        unsigned char* w_ = &lds[BUF][(arow + 32 * (I)) * kAbhStride + akq * 8];              \
        *reinterpret_cast<uint2*>(w_) = make_uint2(h01, h23);                                 \
        *reinterpret_cast<uint2*>(w_ + kAbhPlane) = make_uint2(l01, l23);                     \
    }
#define ABH_PARK(BUF)                                                                         \
    ABH_A(ra0, 0, BUF) ABH_A(ra1, 1, BUF) ABH_A(ra2, 2, BUF) ABH_A(ra3, 3, BUF)               \
    *reinterpret_cast<uint4*>(&lds[BUF][2 * kAbhPlane + brow * kAbhStride + bq * 16]) = bh0;  \
    *reinterpret_cast<uint4*>(&lds[BUF][2 * kAbhPlane + (brow + 64) * kAbhStride + bq * 16]) = bh1; \
    *reinterpret_cast<uint4*>(&lds[BUF][3 * kAbhPlane + brow * kAbhStride + bq * 16]) = bl0;  \
    *reinterpret_cast<uint4*>(&lds[BUF][3 * kAbhPlane + (brow + 64) * kAbhStride + bq * 16]) = bl1;

    auto compute = [&](int buf) {
        const unsigned char* base = lds[buf];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8v ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ra = (wm * 64 + q * 32 + li) * kAbhStride + ks * 32 + 16 * lh;
                const int rb = (wn * 64 + q * 32 + li) * kAbhStride + ks * 32 + 16 * lh;
                ah[q] = *reinterpret_cast<const h8v*>(base + ra);
                al[q] = *reinterpret_cast<const h8v*>(base + kAbhPlane + ra);
                bh[q] = *reinterpret_cast<const h8v*>(base + 2 * kAbhPlane + rb);
                bl[q] = *reinterpret_cast<const h8v*>(base + 3 * kAbhPlane + rb);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bl[c], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[a], bh[c], acc[a][c], 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a], bh[c], acc[a][c], 0, 0, 0);
                }
        }
    };

    ABH_FETCH(0)
    ABH_PARK(0)
    __syncthreads();
    int buf = 0;
    for (int k0 = 32; k0 < g.K; k0 += 32) {
        ABH_FETCH(k0)
        __builtin_amdgcn_sched_barrier(0);
        compute(buf);
        __builtin_amdgcn_sched_barrier(0);
        ABH_PARK(buf ^ 1)
        __syncthreads();
        buf ^= 1;
    }
    compute(buf);
#undef ABH_FETCH
#undef ABH_A
#undef ABH_PARK

    float vmax = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n = n0 + (wn * 2 + c) * 32 + li;
            const float r1b = g.r1a ? g.r1b[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long m = m0 + (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[a][c][r] * ginv;
                if (g.r1a) v = fmaf(g.r1a[m * g.r1a_ld], r1b, v);
                v = hmask[a][c][r] > 0.f ? v : g.alpha * v;
                vmax = fmaxf(vmax, fabsf(v));
                g.Out[m * g.ldo + n] = v;
            }
        }
    if (g.gmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if (lane == 0) {
            unsigned* slot = g.gmax + (blockIdx.x & 63);
            const unsigned vb = __float_as_uint(vmax);
            if (vb > *slot) atomicMax(slot, vb);
        }
    }
}

void launch_gemm_abt_h(const GemmAbt& g, hipStream_t s) {
    if (g.M <= 0) return;
    hipLaunchKernelGGL(gemm_abt_h_kernel, dim3((unsigned)((g.M / 128) * (g.N / 128))), dim3(256), 0, s, g);
}

// ------------------------------------------------------------------------------------------------
// head_wgrad: weight gradients of the two heads (N = 4 columns of Graw): a (K x 4) result needs no matrix core.  One
// workgroup per row slab, one thread per column of A; writes the same partial layout as gemm_atb (row Kp = column sums).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_wgrad_kernel(const GemmAtb g) {
    const int split = blockIdx.x, t = threadIdx.x;
    const long long ms = (long long)split * g.rows_per_split;
    const long long me = ms + g.rows_per_split < g.M ? ms + g.rows_per_split : g.M;
    float* part = g.partial + (size_t)split * (g.Kp + 1) * g.Nw;
    for (int k = t; k < g.Kp + 1; k += 256) {
        const bool ones = k == g.Kp;                       // the extra row: column sums of G
        const bool valid = ones || k < g.K;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (valid)
            for (long long m = ms; m < me; m += 8) {       // slabs are multiples of 16 rows: 8 loads in flight
                float av[8];
                float4 gv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    gv[q] = *reinterpret_cast<const float4*>(g.G + (m + q) * g.ldg);
                    av[q] = ones ? 1.0f : g.a_f16 ? (float)reinterpret_cast<const _Float16*>(g.A)[(m + q) * g.lda + k]
                                                  : g.A[(m + q) * g.lda + k];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    a0 = fmaf(av[q], gv[q].x, a0); a1 = fmaf(av[q], gv[q].y, a1);
                    a2 = fmaf(av[q], gv[q].z, a2); a3 = fmaf(av[q], gv[q].w, a3);
                }
            }
        float* o = part + (size_t)k * g.Nw;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    }
}

// fp16 activations (mixed_float16 policy).  A row of A is read as a whole by neighbouring threads (8 bytes = four columns
// each): K / 4 column quads x RL row lanes per workgroup, four rows in flight per thread, the row lanes' partial sums
// combined through LDS in a fixed order.  (The first version, one thread per column pair striding down the rows with
// 4-byte loads, ran at 1.1 TB/s.)
template <bool F16>      // F16: A holds fp16 elements (mixed_float16 policy); else fp32 (16-byte loads of four columns)
__global__ __launch_bounds__(256) void head_wgrad_rows_kernel(const GemmAtb g) {
    __shared__ float red[256][17];                          // [thread][16 sums + 1 pad]
    __shared__ float gsum[256][4];
    const int split = blockIdx.x, t = threadIdx.x;
    const long long ms = (long long)split * g.rows_per_split;
    const long long me = ms + g.rows_per_split < g.M ? ms + g.rows_per_split : g.M;
    float* part = g.partial + (size_t)split * (g.Kp + 1) * g.Nw;
    const uint16_t* A = reinterpret_cast<const uint16_t*>(g.A);
    const int KQ = g.Kp / 4;                               // column quads (Kp is a multiple of 4; <= 80)
    const int RL = 256 / KQ;                               // row lanes
    const int cq = t % KQ, rl = t / KQ;
    const bool active = rl < RL;
    const bool live = 4 * cq < g.K;                        // columns beyond K are padding
    float a[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) a[c][q] = 0.f;
    float gs[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        for (long long m = ms + rl; m < me; m += 4LL * RL) {
            uint2 av[4];
            float4 af[4];
            float4 gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long mm = m + (long long)u * RL;
                const bool in = mm < me;
                gv[u] = in ? *reinterpret_cast<const float4*>(g.G + mm * g.ldg) : make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (F16) av[u] = in && live ? *reinterpret_cast<const uint2*>(A + mm * g.lda + 4 * cq) : make_uint2(0u, 0u);
                else af[u] = in && live ? *reinterpret_cast<const float4*>(g.A + mm * g.lda + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float x[4];
                if constexpr (F16) {
                    const h2v lo = __builtin_bit_cast(h2v, av[u].x), hi = __builtin_bit_cast(h2v, av[u].y);
                    x[0] = (float)lo[0]; x[1] = (float)lo[1]; x[2] = (float)hi[0]; x[3] = (float)hi[1];
                } else {
                    x[0] = af[u].x; x[1] = af[u].y; x[2] = af[u].z; x[3] = af[u].w;
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    a[c][0] = fmaf(x[c], gv[u].x, a[c][0]); a[c][1] = fmaf(x[c], gv[u].y, a[c][1]);
                    a[c][2] = fmaf(x[c], gv[u].z, a[c][2]); a[c][3] = fmaf(x[c], gv[u].w, a[c][3]);
                }
                if (cq == 0) { gs[0] += gv[u].x; gs[1] += gv[u].y; gs[2] += gv[u].z; gs[3] += gv[u].w; }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[t][4 * c + q] = a[c][q];
#pragma unroll
    for (int q = 0; q < 4; ++q) gsum[t][q] = gs[q];
    __syncthreads();
    // column `col` (Kp <= 320 of them), summed over the row lanes in a fixed order; thread 255 adds the column sums of G
    for (int col = t; col < g.Kp; col += 256) {
        const int q4 = col / 4, c = col % 4;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < RL; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] += red[r * KQ + q4][4 * c + q];
        float* dst = part + (size_t)col * g.Nw;
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
    }
    if (t == 255) {
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < RL; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] += gsum[r * KQ][q];
        float* dst = part + (size_t)g.Kp * g.Nw;
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
    }
}

// Fragment-major A (the fused trainer, frag_layout.h::frag_index): for one column quad the 32 rows of a block are 256
// (fp16) or 512 (fp32) consecutive bytes.  Eight neighbouring lanes share a column quad and take four rows each -- a
// whole block per round, one contiguous run per quad -- over all the slab's blocks; their sums meet in a fixed shuffle
// butterfly (no LDS), lane 0 of the eight writes the 4 x 4 results.  A workgroup covers 32 quads per pass over the slab.
template <bool F16>
__global__ __launch_bounds__(256) void head_wgrad_frag_kernel(const GemmAtb g) {
    const int split = blockIdx.x, t = threadIdx.x, rl = t & 7;
    const long long ms = (long long)split * g.rows_per_split;           // multiple of 32
    const long long me = ms + g.rows_per_split < g.M ? ms + g.rows_per_split : g.M;
    float* part = g.partial + (size_t)split * (g.Kp + 1) * g.Nw;
    const uint16_t* A16 = reinterpret_cast<const uint16_t*>(g.A);
    const int KQ = g.Kp / 4;
    for (int cq = t >> 3; cq < KQ; cq += 32) {
        const bool live = 4 * cq < g.K;                    // columns beyond K are padding
        float a[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[c][q] = 0.f;
        float gs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (long long m = ms + 4 * rl; m < me; m += 32) {
            const long long a0 = frag_index(m, 4 * cq, g.lda);
            float x[4][4];
            if constexpr (F16) {
                uint4 v0 = make_uint4(0u, 0u, 0u, 0u), v1 = v0;
                if (live) { v0 = *reinterpret_cast<const uint4*>(A16 + a0); v1 = *reinterpret_cast<const uint4*>(A16 + a0 + 8); }
                const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const h2v lo = __builtin_bit_cast(h2v, w[2 * u]), hi = __builtin_bit_cast(h2v, w[2 * u + 1]);
                    x[u][0] = (float)lo[0]; x[u][1] = (float)lo[1]; x[u][2] = (float)hi[0]; x[u][3] = (float)hi[1];
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 v = live ? *reinterpret_cast<const float4*>(g.A + a0 + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
                    x[u][0] = v.x; x[u][1] = v.y; x[u][2] = v.z; x[u][3] = v.w;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 gv = *reinterpret_cast<const float4*>(g.G + (m + u) * g.ldg);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    a[c][0] = fmaf(x[u][c], gv.x, a[c][0]); a[c][1] = fmaf(x[u][c], gv.y, a[c][1]);
                    a[c][2] = fmaf(x[u][c], gv.z, a[c][2]); a[c][3] = fmaf(x[u][c], gv.w, a[c][3]);
                }
                gs[0] += gv.x; gs[1] += gv.y; gs[2] += gv.z; gs[3] += gv.w;
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) a[c][q] += __shfl_xor(a[c][q], o);
#pragma unroll
            for (int q = 0; q < 4; ++q) gs[q] += __shfl_xor(gs[q], o);
        }
        if (rl == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float* dst = part + (size_t)(4 * cq + c) * g.Nw;
                dst[0] = a[c][0]; dst[1] = a[c][1]; dst[2] = a[c][2]; dst[3] = a[c][3];
            }
            if (cq == 0) {      // column sums of G (bias gradient)
                float* dst = part + (size_t)g.Kp * g.Nw;
                dst[0] = gs[0]; dst[1] = gs[1]; dst[2] = gs[2]; dst[3] = gs[3];
            }
        }
    }
}

void launch_head_wgrad(const GemmAtb& g, hipStream_t s) {
    if (g.frag) {
        const int fsplits = (int)((g.M + g.rows_per_split - 1) / g.rows_per_split);
        if (g.a_f16) hipLaunchKernelGGL(head_wgrad_frag_kernel<true>, dim3((unsigned)fsplits), dim3(256), 0, s, g);
        else hipLaunchKernelGGL(head_wgrad_frag_kernel<false>, dim3((unsigned)fsplits), dim3(256), 0, s, g);
        return;
    }
    const int splits = (int)((g.M + g.rows_per_split - 1) / g.rows_per_split);
    // rows read as a whole by neighbouring threads (needs at most 256 column quads; Kp <= 320 here); the column-strided
    // head_wgrad_kernel stays for wider matrices
    if (g.Kp % 4 == 0 && g.Kp <= 1024 && g.lda % 4 == 0) {
        if (g.a_f16) hipLaunchKernelGGL(head_wgrad_rows_kernel<true>, dim3((unsigned)splits), dim3(256), 0, s, g);
        else hipLaunchKernelGGL(head_wgrad_rows_kernel<false>, dim3((unsigned)splits), dim3(256), 0, s, g);
        return;
    }
    hipLaunchKernelGGL(head_wgrad_kernel, dim3((unsigned)splits), dim3(256), 0, s, g);
}

// ------------------------------------------------------------------------------------------------
// reduce_grad / relayout / adam: element-wise over one layer
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int train_row_of_blob_row(int kb, int rowmap) {
    // layer 4: the blob holds [xyz_enc (33) ; hidden (256)] rows, the training layout [hidden ; xyz_enc]
    if (rowmap == 1) return kb < 33 ? 256 + kb : kb - 33;
    return kb;
}

// L lanes per gradient entry: lane q adds the partials of slabs q, q + L, ... and the L sums are combined in a fixed
// butterfly -- the same order on every run (bit-reproducible gradients), L x the loads in flight of a serial sum.
// L = 8 for the wide layers that do not take the vector path below, 64 for the heads (a few hundred entries, a thousand
// slabs: with 8 lanes each thread walked 128 dependent-latency loads, 41 us per launch).
template <int L>
__global__ void reduce_grad_kernel(const ReduceArgs a) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = tid / L, q = tid % L;
    const int total = (a.K_real + 1) * a.N_real;
    const bool live = e < total;
    const int ee = live ? e : 0;
    const int kb = ee / a.N_real, n = ee % a.N_real;
    const bool is_bias = kb == a.K_real;
    const int kt = is_bias ? a.Kp : train_row_of_blob_row(kb, a.rowmap);
    const float* p = a.partial + (size_t)kt * a.Nw + a.n_src_off + n;
    const size_t stride = (size_t)(a.Kp + 1) * a.Nw;
    float s = 0.f;
    if (live)
        for (int i = q; i < a.splits; i += L) s += p[i * stride];
#pragma unroll
    for (int o = 1; o < L; o <<= 1) s += __shfl_xor(s, o);
    if (live && q == 0) {
        float* dst = is_bias ? a.grad_b + n : a.grad_w + (size_t)kb * a.N_real + n;
        *dst = a.accumulate ? *dst + s : s;
    }
}

// The wide layers (N_real, Nw and the source offset multiples of 4): a workgroup owns 64 float4 columns of the
// [K + 1] x [N] gradient, its eight waves each add every eighth slab with 16-byte loads that cover whole 1 KiB rows of a
// slab, and the eight sums meet in LDS in a fixed order.  (The 8-lane kernel read 32 bytes per row and instruction.)
__global__ __launch_bounds__(512) void reduce_grad_vec_kernel(const ReduceBatch bat) {
    __shared__ float4 sm[8][64];
    const ReduceArgs& a = bat.e[blockIdx.y];      // one grid row per layer of the batch
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n4 = a.N_real / 4, total4 = (a.K_real + 1) * n4;
    if ((int)blockIdx.x * 64 >= total4) return;   // the grid is as wide as the largest layer
    const int c4 = blockIdx.x * 64 + lane;
    const bool live = c4 < total4;
    const int cc = live ? c4 : 0;
    const int kb = cc / n4, n = 4 * (cc % n4);
    const bool is_bias = kb == a.K_real;
    const int kt = is_bias ? a.Kp : train_row_of_blob_row(kb, a.rowmap);
    const float* p = a.partial + (size_t)kt * a.Nw + a.n_src_off + n;
    const size_t stride = (size_t)(a.Kp + 1) * a.Nw;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
#pragma unroll 8
        for (int i = w; i < a.splits; i += 8) {
            const float4 v = *reinterpret_cast<const float4*>(p + i * stride);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sm[w][lane] = s;
    __syncthreads();
    if (w == 0 && live) {
        float r[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* f = reinterpret_cast<const float*>(&sm[0][lane]) + c;
            constexpr int W8 = 64 * 4;      // floats between the rows of sm
            r[c] = ((f[0] + f[W8]) + (f[2 * W8] + f[3 * W8])) + ((f[4 * W8] + f[5 * W8]) + (f[6 * W8] + f[7 * W8]));
        }
        float* dst = is_bias ? a.grad_b + n : a.grad_w + (size_t)kb * a.N_real + n;
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[c] = a.accumulate ? dst[c] + r[c] : r[c];
    }
}

bool reduce_grad_is_wide(const ReduceArgs& a) {
    return a.N_real % 4 == 0 && a.Nw % 4 == 0 && a.n_src_off % 4 == 0 && a.N_real >= 64;
}

// every entry must satisfy reduce_grad_is_wide
void launch_reduce_grad_batch(const ReduceBatch& b, hipStream_t s) {
    if (b.n <= 0) return;
    int blocks = 0;
    for (int e = 0; e < b.n; ++e) {
        const int entries = (b.e[e].K_real + 1) * b.e[e].N_real;
        blocks = blocks > (entries / 4 + 63) / 64 ? blocks : (entries / 4 + 63) / 64;
    }
    hipLaunchKernelGGL(reduce_grad_vec_kernel, dim3(blocks, b.n), dim3(512), 0, s, b);
}

void launch_reduce_grad(const ReduceArgs& a, hipStream_t s) {
    const int entries = (a.K_real + 1) * a.N_real;
    if (reduce_grad_is_wide(a)) {
        ReduceBatch b{};
        b.n = 1; b.e[0] = a;
        launch_reduce_grad_batch(b, s);
    } else if (entries <= 4096) {
        hipLaunchKernelGGL(reduce_grad_kernel<64>, dim3((entries * 64 + 255) / 256), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(reduce_grad_kernel<8>, dim3((entries * 8 + 255) / 256), dim3(256), 0, s, a);
    }
}

__global__ void relayout_kernel(const RelayoutArgs a) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.Kp * a.Np) return;
    const int kt = e / a.Np, n = e % a.Np;
    // inverse of train_row_of_blob_row
    int kb = kt;
    if (a.rowmap == 1) kb = kt < 256 ? kt + 33 : kt - 256;
    const bool valid = n < a.N_real && kb >= 0 && kb < a.K_real && (a.rowmap != 1 || kt < 256 + 33);
    const float v = valid ? a.w[(size_t)kb * a.N_real + n] : 0.f;
    a.W[(size_t)kt * a.Np + n] = v;
    a.WT[(size_t)n * a.Kp + kt] = v;
    if (a.Whi) {
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        a.Whi[(size_t)kt * a.Np + n] = __builtin_bit_cast(uint16_t, hi);
        a.Wlo[(size_t)kt * a.Np + n] = __builtin_bit_cast(uint16_t, lo);
    }
    if (kt == 0) a.bias[n] = n < a.N_real ? a.b[n] : 0.f;
}

void launch_relayout(const RelayoutArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(relayout_kernel, dim3((a.Kp * a.Np + 255) / 256), dim3(256), 0, s, a);
}

__global__ void adam_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ g, size_t n, float lr, float b1, float b2, float eps,
                            const OptState* __restrict__ st) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !st->apply_ok) return;          // LossScaleOptimizer.apply_gradients drops a non-finite step
    // Keras-2.7 Adam, dense non-amsgrad update (lr_t carries the bias correction; epsilon is not rescaled)
    const float lr_t = lr * st->adam_corr;
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] + (gi * gi - v[i]) * (1.f - b2);
    m[i] = mi;
    v[i] = vi;
    w[i] = w[i] - lr_t * mi / (sqrtf(vi) + eps);
}

void launch_adam(float* w, float* m, float* v, const float* g, size_t n, float lr, float beta1, float beta2, float eps,
                 const OptState* st, hipStream_t s) {
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, m, v, g, n, lr, beta1, beta2,
                       eps, st);
}

// One thread: the verdict of the gradients just computed.  mixed_float16 policy = Keras 2.7 LossScaleOptimizer with
// dynamic scaling: a non-finite step is dropped and halves the scale, `growth` finite steps in a row double it.
__global__ void opt_verdict_kernel(OptState* st) {
    const int ok = st->finite;
    st->apply_ok = ok;
    if (st->dynamic) {
        if (ok) {
            if (++st->good >= st->growth) { st->scale *= 2.f; st->good = 0; }
        } else {
            st->scale = st->scale > 1.f ? st->scale * 0.5f : 1.f;
            st->good = 0;
            st->skipped += 1;
        }
        st->inv_scale = 1.0f / st->scale;
    }
    st->finite = 1;
}

__global__ void opt_begin_kernel(OptState* st) { st->finite = 1; }
void launch_opt_begin(OptState* st, hipStream_t s) { hipLaunchKernelGGL(opt_begin_kernel, dim3(1), dim3(1), 0, s, st); }

void launch_opt_verdict(OptState* st, hipStream_t s) { hipLaunchKernelGGL(opt_verdict_kernel, dim3(1), dim3(1), 0, s, st); }

// One thread, after the Adam launches of a step: count the update and prepare the next bias correction (in double)
__global__ void opt_tick_kernel(OptState* st, float b1, float b2) {
    if (!st->apply_ok) return;
    st->iterations += 1;
    const double t = (double)(st->iterations + 1);
    st->adam_corr = (float)(sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
}

void launch_opt_tick(OptState* st, float beta1, float beta2, hipStream_t s) {
    hipLaunchKernelGGL(opt_tick_kernel, dim3(1), dim3(1), 0, s, st, beta1, beta2);
}

// ------------------------------------------------------------------------------------------------
// encode: sample rows -> xyz / view-direction encodings written straight into the concat buffers
//   sample_along_rays  src/UtilsCV.py:584-599, get_view_directions :124-143, positional encodings
//   src/UtilsNeuralRadianceField.py:52-85.  Rows >= N*S (padding to 128) get zeros.
// ------------------------------------------------------------------------------------------------
template <typename T>      // T = float, or _Float16 for the mixed_float16 policy's half-width activation buffers
__global__ void train_encode_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                    const float* __restrict__ z, long long row0, long long M, int S, long long Mp,
                                    int n_angles, int xyz_mode, T* __restrict__ C4, T* __restrict__ C8, int frag) {
    // local row m of this chunk = global sample row row0 + m; xyz_mode: o = xyz (M,3), d = view_dirs (M,3) or null
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= Mp) return;
    // This row's xyz / direction encoding = columns 256 .. of the concat buffers, built in registers and written four columns at
    // a time (16 B of floats, 8 B of halfs): in the fragment-major buffers the slots of neighbouring rows are neighbours, so a
    // store instruction covers one contiguous run (written element by element it touched a 4-byte piece of every slot:
    // 110 -> 60 us per float32-policy step, 66 -> 35 us under mixed_float16)
    float ev[kXyzPad], dv[kDirPad];
#pragma unroll
    for (int i = 0; i < kXyzPad; ++i) ev[i] = 0.f;
#pragma unroll
    for (int i = 0; i < kDirPad; ++i) dv[i] = 0.f;
    if (m < M) {
        const long long gm = row0 + m;
        const float kPi = 3.1415927410125732f;
        float p[3], v3[3] = {0.f, 0.f, 0.f};
        if (xyz_mode) {
            p[0] = o[gm * 3 + 0]; p[1] = o[gm * 3 + 1]; p[2] = o[gm * 3 + 2];
            if (d) { v3[0] = d[gm * 3 + 0]; v3[1] = d[gm * 3 + 1]; v3[2] = d[gm * 3 + 2]; }
        } else {
            const long long r = gm / S;
            const float zz = z[gm];
            const float4 oo = reinterpret_cast<const float4*>(o)[r], dd = reinterpret_cast<const float4*>(d)[r];
            p[0] = __fadd_rn(oo.x, __fmul_rn(dd.x, zz));
            p[1] = __fadd_rn(oo.y, __fmul_rn(dd.y, zz));
            p[2] = __fadd_rn(oo.z, __fmul_rn(dd.z, zz));
            v3[0] = dd.x; v3[1] = dd.y; v3[2] = dd.z;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ev[c * 11] = p[c];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const float th = __fmul_rn(p[c], kPi * (float)(1 << k));
                ev[c * 11 + 1 + 2 * k] = sin_shifted(th, 0);
                ev[c * 11 + 2 + 2 * k] = sin_shifted(th, 1);
            }
        }
        const int ncomp = n_angles > 0 ? n_angles + 1 : 0;     // the xyz-only network has no direction input
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = n_angles == 2 ? v3[c] : (c == 0 ? v3[0] : v3[2]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float th = __fmul_rn(v, kPi * (float)(1 << k));
                const float sn = sin_shifted(th, 0), cs = sin_shifted(th, 1);
                dv[c * 8 + 2 * k] = c < ncomp ? sn : 0.f;
                dv[c * 8 + 2 * k + 1] = c < ncomp ? cs : 0.f;
            }
        }
    }
    auto put4 = [&](T* base, int ld, int i, const float* v) {       // columns 256 + i .. 256 + i + 3, i % 4 == 0
        T* q = base + (frag ? frag_index(m, 256 + i, ld) : m * ld + 256 + i);
        if constexpr (sizeof(T) == 4) {
            *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            const h2v a = {(_Float16)v[0], (_Float16)v[1]}, b = {(_Float16)v[2], (_Float16)v[3]};
            *reinterpret_cast<uint2*>(q) = make_uint2(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b));
        }
    };
#pragma unroll
    for (int i = 0; i < kXyzPad; i += 4) put4(C4, kLdC4, i, ev + i);
#pragma unroll
    for (int i = 0; i < kDirPad; i += 4) put4(C8, kLdC8, i, dv + i);
}

void launch_train_encode(const float* o, const float* d, const float* z, long long row0, long long M, int S,
                         long long Mp, int n_angles, int xyz_mode, float* C4, float* C8, hipStream_t s, bool half_out,
                         bool frag) {
    if (Mp <= 0) return;
    const dim3 grid((unsigned)((Mp + 255) / 256)), block(256);
    if (half_out)
        hipLaunchKernelGGL(train_encode_kernel<_Float16>, grid, block, 0, s, o, d, z, row0, M, S, Mp, n_angles, xyz_mode,
                           reinterpret_cast<_Float16*>(C4), reinterpret_cast<_Float16*>(C8), frag ? 1 : 0);
    else
        hipLaunchKernelGGL(train_encode_kernel<float>, grid, block, 0, s, o, d, z, row0, M, S, Mp, n_angles, xyz_mode, C4, C8,
                           frag ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------
// MSE (Keras MeanSquaredError = mean over all N*3 values, src/NeRF.py:50,151) and its gradient.
// One workgroup, fixed reduction order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void mse_kernel(const float* __restrict__ rgb, const float* __restrict__ tgt,
                                                   long long n3, const OptState* __restrict__ st, float weight,
                                                   float* __restrict__ d_rgb, float* __restrict__ out) {
    __shared__ float red[1024];
    // st->scale: LossScaleOptimizer.get_scaled_loss (1 = none); weight: this term's factor in the ray loss (1 in
    // NeRF.train_step; nerf_train_set_loss_weights) -- out[0] stays the plain MSE (the PSNR metrics read it)
    const float scale = st->scale * 2.0f / (float)n3 * weight;
    float s = 0.f;
    for (long long i = threadIdx.x; i < n3; i += 1024) {
        const float e = rgb[i] - tgt[i];
        s = fmaf(e, e, s);
        d_rgb[i] = e * scale;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] / (float)n3;
}

void launch_mse(const float* rgb, const float* target, long long N, const OptState* st, float weight, float* d_rgb,
                float* mse_out, hipStream_t s) {
    hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(1024), 0, s, rgb, target, N * 3, st, weight, d_rgb, mse_out);
}

// LossScaleOptimizer.get_unscaled_gradients + its finiteness test in one sweep: g *= inv_scale; *all_finite = 0 as soon
// as one entry of either blob is Inf/NaN (the caller sets it to 1 first).  add_a / add_b (nullable): blobs added AFTER the
// test -- nerf_train_render_gradients(accumulate = 1) under mixed_float16 sums its unscaled gradients onto the unscaled
// ray-loss gradients of nerf_train_gradients (like with like; the reference sums both losses before one
// get_scaled_loss / get_unscaled_gradients, src/DietNeRF.py:142-153).
__global__ void unscale_check_kernel(float* __restrict__ ga, float* __restrict__ gb, size_t n, OptState* __restrict__ st,
                                     int check_only, const float* __restrict__ add_a, const float* __restrict__ add_b) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float inv_scale = check_only ? 1.0f : st->inv_scale;
    bool bad = false;
    if (i < n) {
        const float a = ga[i] * inv_scale;
        ga[i] = add_a ? a + add_a[i] : a;
        bad = !(fabsf(a) <= 3.0e38f);
        if (gb) {
            const float b = gb[i] * inv_scale;
            gb[i] = add_b ? b + add_b[i] : b;
            bad |= !(fabsf(b) <= 3.0e38f);
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) st->finite = 0;
}

void launch_unscale_check(float* ga, float* gb, size_t n, OptState* st, hipStream_t s, bool check_only, const float* add_a,
                          const float* add_b) {
    hipLaunchKernelGGL(unscale_check_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ga, gb, n, st,
                       check_only ? 1 : 0, add_a, add_b);
}

// Keras' History keeps the per-epoch MEANS of train_step's metrics (src/NeRF.py:169-177, src/ExecutionRun.py:192): the sums
// are kept on the device (one thread per step), so a training loop reads them once per epoch instead of once per step.
// scal[0] / scal[1] = this step's coarse / fine MSE (mse_kernel); the values added are exactly what read_metrics hands out.
__global__ void metrics_accum_kernel(const float* __restrict__ scal, int fine, float w0, float w1,
                                     double* __restrict__ acc) {
    const float h0 = scal[0], h1 = fine ? scal[1] : 0.f;
    acc[0] += (double)(fine ? w0 * h0 + w1 * h1 : w0 * h0);      // as read_metrics (train_api.hip) forms the loss
    acc[1] += (double)(float)(-10.0 * log10((double)h0));
    acc[2] += fine ? (double)(float)(-10.0 * log10((double)h1)) : 0.0;
    acc[3] += 1.0;
}

void launch_metrics_accum(const float* scal, bool fine, float w0, float w1, double* acc, hipStream_t s) {
    hipLaunchKernelGGL(metrics_accum_kernel, dim3(1), dim3(1), 0, s, scal, fine ? 1 : 0, w0, w1, acc);
}

// LossScaleOptimizer.get_scaled_loss for a caller-supplied upstream gradient (nerf_train_render_gradients under
// mixed_float16): out = in * st->scale, the scale read on the device (the host never knows the current value).
__global__ void scale_by_loss_scale_kernel(const float* __restrict__ in, long long n, const OptState* __restrict__ st,
                                           float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * st->scale;
}

__global__ void gather_blob_kernel(const float* __restrict__ blob, const int32_t* __restrict__ idx, float* __restrict__ out,
                                   size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int32_t t = idx[i];
        out[i] = t ? blob[t - 1] : 0.f;
    }
}

void launch_gather_blob(const float* blob, const int32_t* idx, float* out, size_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(gather_blob_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, blob, idx, out, n);
}

void launch_scale_by_loss_scale(const float* in, long long n, const OptState* st, float* out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(scale_by_loss_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, n, st, out);
}

// ------------------------------------------------------------------------------------------------
// compositing backward (ray_marching, src/UtilsNeuralRadianceField.py:88-115), one ray per wavefront like the forward
// (aux_kernels.hip::composite_kernel).  With T_i the exclusive transmittance and g_w[i] = dL/dw_i:
//     dL/dalpha_i = T_i * (g_w[i] - R_i),   R_{i-1} = g_w[i]*alpha_i + (1 - alpha_i)*R_i,   R_{S-1} = 0
// (no division by 1 - alpha, which is exactly 0 wherever sigma*delta overflows, e.g. the 1e9 last interval).
// alpha, the sigmoids and the output rows are evaluated in parallel over 64-sample chunks taken from the far end of the
// ray; the reverse recurrence travels through the lanes in its canonical order (below).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                            const float* __restrict__ Tin, long long N, int S,
                                                            const float* __restrict__ d_rgb,
                                                            const float* __restrict__ d_w_ext, float* __restrict__ Graw,
                                                            float* __restrict__ d_z) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= N) return;                                        // uniform per wavefront
    const float4* rw = reinterpret_cast<const float4*>(raw) + r * S;
    float4* gr = reinterpret_cast<float4*>(Graw) + r * S;
    const float* zr = z + r * S;
    const float g0 = d_rgb[r * 3 + 0], g1 = d_rgb[r * 3 + 1], g2 = d_rgb[r * 3 + 2];
    float Rc = 0.f;          // R entering the current chunk from the far side
    float dd_far = 0.f;      // dL/ddelta of the first sample of the chunk processed before (the next one along the ray)
    for (int s0 = (S - 1) / 64 * 64; s0 >= 0; s0 -= 64) {
        const int s = s0 + lane;
        const bool in = s < S, last = s + 1 >= S;
        const float4 o = in ? rw[s] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float delta = last ? 1e9f : zr[s + 1] - zr[s];
        const float sigma = fmaxf(o.w, 0.f);
        const float e = in ? expf(-sigma * delta) : 1.0f;       // 1 - alpha
        const float a = 1.0f - e;
        const float T = in ? Tin[r * S + s] : 0.f;
        const float c0 = 1.0f / (1.0f + expf(-o.x)), c1 = 1.0f / (1.0f + expf(-o.y)), c2 = 1.0f / (1.0f + expf(-o.z));
        const float w = a * T;
        float gw = g0 * c0 + g1 * c1 + g2 * c2;
        if (d_w_ext && in) gw += d_w_ext[r * S + s];
        // R in the canonical order R_l = fl(fl(g_w a)_(l+1) + fl(e_(l+1) R_(l+1))): lane l reads its right neighbour through
        // a one-lane wavefront shift (DPP wave_shl:1; lane 63 keeps the value carried in from the chunk beyond) --
        // after k rounds lanes 63-k..63 are final, recomputing them changes nothing.  (A tree-shaped scan over the
        // affine maps is six steps, but neighbouring R then carry unrelated roundings and d_z, a difference of
        // neighbours that the sampler backward multiplies by up to 1e5, loses a digit and a half.)
        const float bs = in ? gw * a : 0.f;
        const float esh = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(1.0f), __float_as_int(e), 0x130, 0xf, 0xf, false));
        const float bsh = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(0.0f), __float_as_int(bs), 0x130, 0xf, 0xf, false));
        float R = lane == 63 ? Rc : 0.f;
        for (int k = 0; k < 63; ++k) {
            const float Rn = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(R), __float_as_int(R), 0x130, 0xf, 0xf, false));
            R = lane == 63 ? Rn : bsh + esh * Rn;
        }
        const float da = T * (gw - R);
        const float dd = last || !in ? 0.f : da * sigma * e;               // dL/ddelta_s
        const float dd_right = __shfl_down(dd, 1);       // (unconditional: a shuffle under "lane != 63" reads 0 from lane 63)
        const float dd_next = lane == 63 ? dd_far : dd_right;
        if (in) {
            float4 out;
            out.x = w * g0 * c0 * (1.0f - c0);
            out.y = w * g1 * c1 * (1.0f - c1);
            out.z = w * g2 * c2 * (1.0f - c2);
            out.w = o.w > 0.f ? da * delta * e : 0.f;
            gr[s] = out;
            if (d_z && !last) d_z[r * S + s + 1] = dd - dd_next;
        }
        Rc = __shfl(bs + e * R, 0);
        dd_far = __shfl(dd, 0);
    }
    if (d_z && lane == 0) d_z[r * S] = -dd_far;
}

void launch_composite_bwd(const float* raw, const float* z, const float* T, long long N, int S, const float* d_rgb,
                          const float* d_w_ext, float* Graw, float* d_z, hipStream_t s) {
    if (N <= 0) return;
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, raw, z, T, N, S, d_rgb,
                       d_w_ext, Graw, d_z);
}

// ------------------------------------------------------------------------------------------------
// rgb head backward: G9[m][j] = (sum_c Graw[m][c] * W9[j][c]) * LeakyReLU'(H9[m][j])    (layers 9 -> 8)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ Graw, const float* __restrict__ W9,
                                                       const float* __restrict__ H9, long long M, float alpha,
                                                       float* __restrict__ G9, unsigned* __restrict__ gmax) {
    __shared__ float w[128 * 3];
    for (int i = threadIdx.x; i < 128 * 3; i += 256) w[i] = W9[(i / 3) * 32 + (i % 3)];
    __syncthreads();
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;   // one float4 of G9 per thread
    const long long m = e >> 5;
    const int j = (int)(e & 31) * 4;
    if (m >= M) return;
    const float4 g = reinterpret_cast<const float4*>(Graw)[m];
    const float4 h = *reinterpret_cast<const float4*>(H9 + m * 128 + j);
    float4 out;
    const float hv[4] = {h.x, h.y, h.z, h.w};
    float ov[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float v = g.x * w[(j + q) * 3 + 0] + g.y * w[(j + q) * 3 + 1] + g.z * w[(j + q) * 3 + 2];
        ov[q] = hv[q] > 0.f ? v : alpha * v;
    }
    out.x = ov[0]; out.y = ov[1]; out.z = ov[2]; out.w = ov[3];
    *reinterpret_cast<float4*>(G9 + m * 128 + j) = out;
    if (gmax) {      // uniform per launch.  One slot check per workgroup: 65 k workgroups each polling per wave was 0.7 ms
        __shared__ float wmax[4];
        float vmax = fmaxf(fmaxf(fabsf(ov[0]), fabsf(ov[1])), fmaxf(fabsf(ov[2]), fabsf(ov[3])));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = vmax;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float m4 = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            unsigned* slot = gmax + (blockIdx.x & 63);
            const unsigned vb = __float_as_uint(m4);
            if (vb > *slot) atomicMax(slot, vb);
        }
    }
}

void launch_head_bwd(const float* Graw, const float* W9, const float* H9, long long M, float alpha, float* G9,
                     unsigned* gmax, hipStream_t s) {
    if (M <= 0) return;
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)((M * 32 + 255) / 256)), dim3(256), 0, s, Graw, W9, H9, M, alpha,
                       G9, gmax);
}

// ------------------------------------------------------------------------------------------------
// positional-encoding backward + sample_along_rays backward: d_z[m] += sum_c dL/dp_c * dir_c
// ------------------------------------------------------------------------------------------------
__global__ void pe_bwd_kernel(const float* __restrict__ dA0, const float* __restrict__ dA0b /* second part to add, or null */,
                              const float* __restrict__ o, const float* __restrict__ d,
                              const float* __restrict__ z, long long M, int S, float* __restrict__ d_z, int frag) {
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const long long r = m / S;
    const float zz = z[m];
    const float kPi = 3.1415927410125732f;
    const float4 oo = reinterpret_cast<const float4*>(o)[r], dd = reinterpret_cast<const float4*>(d)[r];
    const float p[3] = {__fadd_rn(oo.x, __fmul_rn(dd.x, zz)), __fadd_rn(oo.y, __fmul_rn(dd.y, zz)),
                        __fadd_rn(oo.z, __fmul_rn(dd.z, zz))};
    const float dv[3] = {dd.x, dd.y, dd.z};
    // the encoding gradient may come in two parts (fused backward: through layer 4 and through layer 0)
    // frag: the (Mp, 64) rows are fragment-major (frag_layout.h::frag_index; written by the fused backward chain):
    // neighbouring threads read neighbouring 16-byte slots
    float g[36];
#pragma unroll
    for (int q = 0; q < 9; ++q) {                  // 33 floats, read as 9 float4 (rows are 64 floats: no overrun)
        const long long e = frag ? frag_index(m, 4 * q, kXyzPad) : m * kXyzPad + 4 * q;
        float4 v = *reinterpret_cast<const float4*>(dA0 + e);
        if (dA0b) { const float4 w = *reinterpret_cast<const float4*>(dA0b + e); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        g[4 * q] = v.x; g[4 * q + 1] = v.y; g[4 * q + 2] = v.z; g[4 * q + 3] = v.w;
    }
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float dp = g[c * 11];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const float f = kPi * (float)(1 << k);
            const float th = __fmul_rn(p[c], f);
            const float sn = sin_shifted(th, 0), cs = sin_shifted(th, 1);
            dp += (cs * g[c * 11 + 1 + 2 * k] - sn * g[c * 11 + 2 + 2 * k]) * f;
        }
        acc += dp * dv[c];
    }
    d_z[m] += acc;
}

// ------------------------------------------------------------------------------------------------
// Backward of z = sort(concat(z_new, z_coarse)) (src/NeRF.py:132) w.r.t. z_new: sample_pdf_kernel (aux_kernels.hip)
// puts the k-th new depth (they are sorted) into slot k + #{coarse depths < it}; the gradient of that slot is its.
// ------------------------------------------------------------------------------------------------
__global__ void unmerge_grad_kernel(const float* __restrict__ z_new, const float* __restrict__ z_c,
                                    const float* __restrict__ d_zm, long long N, int S, int Sf,
                                    float* __restrict__ d_zf) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Sf) return;
    const long long ray = i / Sf;
    const int k = (int)(i % Sf);
    const float v = z_new[i];
    const float* zc = z_c + ray * S;
    int lo = 0, hi = S;                       // # coarse depths < v
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (zc[mid] < v) lo = mid + 1; else hi = mid; }
    d_zf[i] = d_zm[ray * (S + Sf) + k + lo];
}

void launch_unmerge_grad(const float* z_new, const float* z_c, const float* d_zm, long long N, int S, int Sf, float* d_zf,
                         hipStream_t s) {
    const long long n = N * Sf;
    if (n <= 0) return;
    hipLaunchKernelGGL(unmerge_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z_new, z_c, d_zm, N, S, Sf,
                       d_zf);
}

void launch_pe_bwd(const float* dA0, const float* dA0b, const float* o, const float* d, const float* z, long long N, int S,
                   float* d_z, hipStream_t s, bool frag) {
    const long long M = N * S;
    if (M <= 0) return;
    hipLaunchKernelGGL(pe_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, dA0, dA0b, o, d, z, M, S, d_z,
                       frag ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------
// inverse-CDF sampler backward (get_z_vals_from_prob_dist_func, src/UtilsCV.py:502-539): the reference takes
// gradients through pdf/cdf/gather/lerp/sort (indices are constants), so the fine loss reaches the coarse
// weights.  One ray per wavefront; the forward quantities are recomputed with the forward kernel's arithmetic
// (same cdf bits -> same bins and the same sort permutation).
// ------------------------------------------------------------------------------------------------
constexpr int kBwdWaves = 4;

__global__ __launch_bounds__(64 * kBwdWaves) void sample_pdf_bwd_kernel(
    const float* __restrict__ weights, const float* __restrict__ zin, long long N, int S, int Sf,
    const float* __restrict__ u, uint64_t seed, long long ray_base, const float* __restrict__ d_zf,
    float* __restrict__ d_w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long long ray = (long long)blockIdx.x * kBwdWaves + wave;
    if (ray >= N) return;
    const int S4 = (S + 3) & ~3, Sf4 = (Sf + 3) & ~3;
    const int per_wave = 5 * S4 + 6 * Sf4 + 4;
    float* cdf = lds + wave * per_wave;
    float* zc = cdf + S4;
    float* wv = zc + S4;
    float* dcdf = wv + S4;
    float* dcdf_hi = dcdf + S4;
    float* zn = dcdf_hi + S4;
    float* span = zn + Sf4;       // z_hi - z_lo
    float* tt = span + Sf4;
    float* den = tt + Sf4;        // clamped denominators are stored negated
    int* ilo = reinterpret_cast<int*>(den + Sf4);
    int* ihi = ilo + Sf4;
    float* scal = reinterpret_cast<float*>(ihi + Sf4);   // [0] = sum + eps, [1] = sum_i dpdf_i * w_i
    const float* wr = weights + ray * S;
    const float* zr = zin + ray * S;
    for (int s = lane; s < S; s += 64) { wv[s] = wr[s]; zc[s] = zr[s]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        float sum = 0.f;
        for (int s = 0; s < S; ++s) sum = __fadd_rn(sum, wv[s]);
        const float D = __fadd_rn(sum, 1e-7f);
        float acc = 0.f;
        for (int s = 0; s < S; ++s) {
            acc = __fadd_rn(acc, __fdiv_rn(wv[s], D));
            cdf[s] = acc;
        }
        scal[0] = D;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float kInf = __builtin_huge_valf();
    for (int k = lane; k < Sf4; k += 64) {
        if (k >= Sf) { zn[k] = kInf; continue; }
        const float uu = u ? u[ray * Sf + k] : philox_uniform(seed, (uint64_t)(ray_base + ray), k, 1u);
        int lo = 0, hi = S;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] < uu) lo = mid + 1; else hi = mid;
        }
        const int b = max(0, lo - 1);
        const int t = min(S - 1, lo);
        const float c_lo = cdf[b], c_hi = cdf[t];
        const int bz = min(max(b, 0), S - 2), tz = min(max(t, 0), S - 2);
        const float z_lo = __fmul_rn(0.5f, __fadd_rn(zc[bz + 1], zc[bz]));
        const float z_hi = __fmul_rn(0.5f, __fadd_rn(zc[tz + 1], zc[tz]));
        float dn = __fsub_rn(c_hi, c_lo);
        const bool clamped = dn < 1e-5f;
        dn = clamped ? 1e-5f : dn;
        const float tv = __fdiv_rn(__fsub_rn(uu, c_lo), dn);
        zn[k] = __fadd_rn(z_lo, __fmul_rn(tv, __fsub_rn(z_hi, z_lo)));
        span[k] = __fsub_rn(z_hi, z_lo);
        tt[k] = tv;
        den[k] = clamped ? -dn : dn;
        ilo[k] = b;
        ihi[k] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // gradient of each unsorted sample = gradient of its slot in the sorted output (stable rank, as forward)
    float dlo_r[4], dhi_r[4];
    int nk = 0;
    for (int k = lane; k < Sf; k += 64, ++nk) {
        const float v = zn[k];
        int rank = 0;
        for (int i = 0; i < Sf4; i += 4) {
            const float4 q = *reinterpret_cast<const float4*>(zn + i);
            rank += (q.x < v || (q.x == v && i + 0 < k)) ? 1 : 0;
            rank += (q.y < v || (q.y == v && i + 1 < k)) ? 1 : 0;
            rank += (q.z < v || (q.z == v && i + 2 < k)) ? 1 : 0;
            rank += (q.w < v || (q.w == v && i + 3 < k)) ? 1 : 0;
        }
        const float g = d_zf[ray * Sf + rank];
        const float dsgn = den[k];
        const bool clamped = dsgn < 0.f;
        const float dn = fabsf(dsgn);
        const float d_t = g * span[k];
        float dlo = -d_t / dn, dhi = 0.f;
        if (!clamped) {
            const float dden = -d_t * tt[k] / dn;
            dhi = dden;
            dlo -= dden;
        }
        if (nk < 4) { dlo_r[nk] = dlo; dhi_r[nk] = dhi; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // zn / span are dead now: reuse them for the per-sample cdf gradients
    nk = 0;
    for (int k = lane; k < Sf; k += 64, ++nk) { zn[k] = dlo_r[nk]; span[k] = dhi_r[nk]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // The +-(d_t t / den) pairs that a sample sends to its two bins are 1e5 times larger than what survives after the
    // reverse cumulative sum: the sums run in double so that they cancel exactly (a per-ray kernel: the cost is nil).
    for (int s = lane; s < S; s += 64) {
        double acc = 0.0;
        for (int k = 0; k < Sf; ++k) {
            if (ilo[k] == s) acc += (double)zn[k];
            if (ihi[k] == s) acc += (double)span[k];
        }
        dcdf[s] = (float)acc;
        dcdf_hi[s] = (float)(acc - (double)(float)acc);      // keep the rounding remainder for the scan
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        double run = 0.0, dot = 0.0;
        for (int s = S - 1; s >= 0; --s) {     // cumsum backward = reverse cumsum
            run += (double)dcdf[s] + (double)dcdf_hi[s];
            dcdf[s] = (float)run;
            dcdf_hi[s] = (float)(run - (double)(float)run);
            dot += run * (double)wv[s];
        }
        scal[1] = (float)dot;
        scal[2] = (float)(dot - (double)(float)dot);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // d_w = (dpdf - sum_i dpdf_i pdf_i) / D is again a difference of nearly equal large numbers: double
    const double D = (double)scal[0], dot = (double)scal[1] + (double)scal[2];
    for (int s = lane; s < S; s += 64)
        d_w[ray * S + s] = (float)(((double)dcdf[s] + (double)dcdf_hi[s]) / D - dot / (D * D));
}

size_t sample_pdf_bwd_lds_bytes(int S, int Sf) {
    const int S4 = (S + 3) & ~3, Sf4 = (Sf + 3) & ~3;
    return (size_t)kBwdWaves * (5 * S4 + 6 * Sf4 + 4) * sizeof(float);
}

void launch_sample_pdf_bwd(const float* weights, const float* z, long long N, int S, int Sf, const float* u,
                           uint64_t seed, long long ray_base, const float* d_zf, float* d_w, hipStream_t s) {
    if (N <= 0) return;
    hipLaunchKernelGGL(sample_pdf_bwd_kernel, dim3((unsigned)((N + kBwdWaves - 1) / kBwdWaves)), dim3(64 * kBwdWaves),
                       sample_pdf_bwd_lds_bytes(S, Sf), s, weights, z, N, S, Sf, u, seed, ray_base, d_zf, d_w);
}

}  // namespace nerf

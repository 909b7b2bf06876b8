// Prototype: the float32-policy weight-gradient GEMM  dW[256 x 256] = sum_rows A[row][k] * G[row][n]
// with NO register staging -- operands copied global -> LDS by LDS-DMA (global_load_lds_dwordx4), MFMA operands read with the
// transposing ds_read_b64_tr_b16, a second operand register set in the registers the staging used to hold, one barrier per
// 32 rows -- against what bounds gemm_atb_p today (profiles/r4_diagnostic_ab.txt section 2: ~2000 cycles of barrier + LDS
// operand reads + staging per 16-row step that do not overlap the 1536 MFMA cycles).
//
// Operand layout ("plane8", what the fused trainer's producers could store with NO extra instruction: the MFMA operand quads
// xh[n] / xl[n] as they stand): per 32-row block and fragment n (16 features) 2 KiB = [plane hi | lo][lane half h][row][16 B],
// the 16 bytes = 8 halfs e = 0..7 of features 16 n + 8 (e >> 2) + 4 h + (e & 3).  A: true values hi + lo.  G: per-row scaled
// operand, true = (hi + lo) * rs[row]; the factor rs[row] * gscale is applied to the operand registers after the read.
// LDS image per 32-row step and operand: 32 chunks of 1 KiB = (column tile ct of 32 features, plane, 16-row half), inside
// a chunk [row quad][fragment parity][h][row % 4] x 16 B: one DMA instruction fills a chunk from four 256-byte runs, and the
// 32 lanes of a transposing read cover 256 consecutive bytes (conflict-free).
// Output: checks a small case against a double-precision reference, then times the trainer's shape (8 layers x 524 288 rows,
// 64 row slabs per layer) and prints operand TB/s.        Build: make -C tools/microbench wgrad_dma_tr
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kStepRows = 32;
constexpr int kOpBytes = 32 * 1024;            // one operand tile of a step: 256 features x 32 rows x 4 B
constexpr int kBufBytes = 2 * kOpBytes;        // A | G
constexpr int kFtab = 2 * kBufBytes;           // row-factor table: 2 buffers x 32 halfs (64 B each, 16-byte aligned)
constexpr int kLdsBytes = kFtab + 2 * 64;

struct Entry { const char* A; const char* G; const uint16_t* rs; float* partial; };
struct Args { Entry e[8]; int n_entries; int splits; long long rows_per_split; float gscale; };

__device__ __forceinline__ void dma_piece(const char* sbase, uint32_t voff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

extern __shared__ __attribute__((aligned(16))) char smem[];

template <int PASSES, bool NO_DMA = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_kernel(const Args a) {
    const int per_entry = a.splits;
    const int ent = blockIdx.x / per_entry, split = blockIdx.x % per_entry;
    if (ent >= a.n_entries) return;
    const Entry E = a.e[ent];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wk = wave >> 2, wn = wave & 3;          // wave tile: features 128 wk .. +127 of A, 64 wn .. +63 of G
    const int li = lane & 31, lh = lane >> 5;
    const long long row0 = (long long)split * a.rows_per_split;
    const int steps = (int)(a.rows_per_split / kStepRows);

    // ---- DMA role of this wave: column tile ct = wave of BOTH operands; piece (pl, rh) = 4 pieces per operand and step ----
    // lane l fills LDS slot l of a chunk: rq = l / 16, g' = (l / 8) % 2, h = (l / 4) % 2, q = l % 4
    const int d_rq = lane >> 4, d_g = (lane >> 3) & 1, d_h = (lane >> 2) & 1, d_q = lane & 3;
    const uint32_t d_lane_off = (uint32_t)((2 * wave + d_g) * 2048 + d_h * 512 + (4 * d_rq + d_q) * 16);
    // piece j (0..7) of this wave's share of a step: operand j / 4, plane (j / 2) % 2, row half j % 2
    auto dma_one = [&](int s, int buf, auto jc) {
        if constexpr (NO_DMA) return;          // timing-only: the compute side alone (stale LDS contents)
        constexpr int j = decltype(jc)::value, op = j >> 2, pl = (j >> 1) & 1, rh = j & 1;
        const size_t blk = (size_t)(row0 / kStepRows + s) * kOpBytes;      // 32-row block of a 256-wide buffer: 32 KiB
        const uint64_t bv = (uint64_t)((op ? E.G : E.A) + blk);            // wave-uniform: pin it to scalar registers for the asm
        const char* base = (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bv >> 32)) << 32) |
                                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bv));
        dma_piece(base, d_lane_off + pl * 1024 + rh * 256, (uint32_t)(buf * kBufBytes + op * kOpBytes + (wave * 4 + pl * 2 + rh) * 1024));
    };
    auto dma_step = [&](int s, int buf) {
        dma_one(s, buf, std::integral_constant<int, 0>{}); dma_one(s, buf, std::integral_constant<int, 1>{});
        dma_one(s, buf, std::integral_constant<int, 2>{}); dma_one(s, buf, std::integral_constant<int, 3>{});
        dma_one(s, buf, std::integral_constant<int, 4>{}); dma_one(s, buf, std::integral_constant<int, 5>{});
        dma_one(s, buf, std::integral_constant<int, 6>{}); dma_one(s, buf, std::integral_constant<int, 7>{});
    };
    // row factors of a step -> fp16 table (threads 0..31): f = rs[row] * gscale, clamped.  The rs value is LOADED a phase before
    // it is used (rs_load) and the table is WRITTEN before the step's DMA pieces go out: the compiler does not see the asm
    // pieces on vmcnt, so a load it waits for after them would wait for all of them.
    auto rs_load = [&](int s) -> uint32_t {      // (every lane loads a valid address; threads 0..31 write the table)
        return (uint32_t)E.rs[row0 + (long long)s * kStepRows + (t & 31)];
    };
    auto factors_write = [&](uint32_t b, int buf) {
        if (t < 32) {
            const float f = fminf(__uint_as_float(b << 16) * a.gscale, 32768.0f);
            reinterpret_cast<_Float16*>(smem + kFtab + buf * 64)[t] = (_Float16)f;
        }
    };

    // ---- transposing operand reads ----
    const int lane16 = lane & 15, tq = lane16 >> 2, tp = lane16 & 3, tg = (lane >> 4) & 1;
    const uint32_t tr_lane = (uint32_t)((2 * lh) * 256 + tg * 128 + (tp & 1) * 64 + tq * 16 + 8 * (tp >> 1));
    struct OpSet { u32x4 ah[4], al[4], gh[2], gl[2]; };
    auto rd = [&](uint32_t off) -> u32x4 {
        const s4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(uintptr_t)(off));
        const s4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(uintptr_t)(off + 256));
        const u32x2 x2 = __builtin_bit_cast(u32x2, x), y2 = __builtin_bit_cast(u32x2, y);
        return u32x4{x2[0], x2[1], y2[0], y2[1]};
    };
    // part i (0..3) of a set's reads: A column tile i (hi, lo) and, for i < 2, G column tile i (hi, lo)
    auto read_part = [&](OpSet& S, int buf, int ks, auto ic) {
        constexpr int i = decltype(ic)::value;
        const uint32_t base = (uint32_t)(buf * kBufBytes) + tr_lane + (uint32_t)ks * 1024;
        const uint32_t ca = base + (uint32_t)((wk * 4 + i) * 4) * 1024;
        S.ah[i] = rd(ca);
        S.al[i] = rd(ca + 2048);
        if constexpr (i < 2) {
            const uint32_t cg = base + kOpBytes + (uint32_t)((wn * 2 + i) * 4) * 1024;
            S.gh[i] = rd(cg);
            S.gl[i] = rd(cg + 2048);
        }
    };
    auto read_factors = [&](int buf, int ks) -> u32x4 {      // rows 16 ks + 8 lh .. + 7 of the step in `buf`: four fp16 pairs
        return *reinterpret_cast<const u32x4*>(smem + kFtab + buf * 64 + (16 * ks + 8 * lh) * 2);
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;
    float cs[2] = {0.f, 0.f};
    // G'' = G' * f_row on the operand registers (+ the bias gradient's column sums on the wk == 0 waves)
    auto scale = [&](OpSet& S, const u32x4& f) {
        const h2 one2 = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                // (scalars first: __builtin_bit_cast applied to an ext-vector ELEMENT lvalue reads element 0 whatever the index --
                // hipcc, ROCm 7.2; found in this kernel's ISA)
                const uint32_t wh = S.gh[q][d], wl = S.gl[q][d], wf = f[d];
                const h2 gh2 = __builtin_bit_cast(h2, wh) * __builtin_bit_cast(h2, wf);
                const h2 gl2 = __builtin_bit_cast(h2, wl) * __builtin_bit_cast(h2, wf);
                S.gh[q][d] = __builtin_bit_cast(uint32_t, gh2);
                S.gl[q][d] = __builtin_bit_cast(uint32_t, gl2);
                if (wk == 0) {      // (wave-uniform)
                    cs[q] = __builtin_amdgcn_fdot2(gh2, one2, cs[q], false);
                    cs[q] = __builtin_amdgcn_fdot2(gl2, one2, cs[q], false);
                }
            }
        }
    };
    auto mfma_part = [&](const OpSet& S, auto ic) {
        constexpr int i = decltype(ic)::value;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const h8 ah = __builtin_bit_cast(h8, S.ah[i]), al = __builtin_bit_cast(h8, S.al[i]);
            const h8 gh = __builtin_bit_cast(h8, S.gh[c]), gl = __builtin_bit_cast(h8, S.gl[c]);
            if constexpr (PASSES == 3) {
                acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gl, acc[i][c], 0, 0, 0);
                acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, gh, acc[i][c], 0, 0, 0);
            }
            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gh, acc[i][c], 0, 0, 0);
        }
    };
    // six MFMAs of set X, then the next part of set Y's reads (and, DMA = true, two of the eight LDS-DMA pieces of step dma_s: a
    // piece costs its wave ~60 issue cycles, eight in a row after the barrier idled the matrix pipe), four times
    auto compute_and_read = [&](const OpSet& X, OpSet& Y, int ybuf, int yks, auto dmac, int dma_s, int dma_buf) {
        constexpr bool DMA = decltype(dmac)::value;
        mfma_part(X, std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0);
        read_part(Y, ybuf, yks, std::integral_constant<int, 0>{});
        if constexpr (DMA) { dma_one(dma_s, dma_buf, std::integral_constant<int, 0>{}); dma_one(dma_s, dma_buf, std::integral_constant<int, 1>{}); }
        __builtin_amdgcn_sched_barrier(0);
        mfma_part(X, std::integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0);
        read_part(Y, ybuf, yks, std::integral_constant<int, 1>{});
        if constexpr (DMA) { dma_one(dma_s, dma_buf, std::integral_constant<int, 2>{}); dma_one(dma_s, dma_buf, std::integral_constant<int, 3>{}); }
        __builtin_amdgcn_sched_barrier(0);
        mfma_part(X, std::integral_constant<int, 2>{}); __builtin_amdgcn_sched_barrier(0);
        read_part(Y, ybuf, yks, std::integral_constant<int, 2>{});
        if constexpr (DMA) { dma_one(dma_s, dma_buf, std::integral_constant<int, 4>{}); dma_one(dma_s, dma_buf, std::integral_constant<int, 5>{}); }
        __builtin_amdgcn_sched_barrier(0);
        mfma_part(X, std::integral_constant<int, 3>{}); __builtin_amdgcn_sched_barrier(0);
        read_part(Y, ybuf, yks, std::integral_constant<int, 3>{});
        if constexpr (DMA) { dma_one(dma_s, dma_buf, std::integral_constant<int, 6>{}); dma_one(dma_s, dma_buf, std::integral_constant<int, 7>{}); }
        __builtin_amdgcn_sched_barrier(0);
    };

    OpSet S0, S1;
    if (steps > 0) {
        const uint32_t r0 = rs_load(0), r1 = rs_load(1);
        factors_write(r0, 0);
        factors_write(r1, 1);
        dma_step(0, 0);
        dma_step(steps > 1 ? 1 : 0, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // this wave's 8 pieces of step 0 have landed
        __syncthreads();
        read_part(S0, 0, 0, std::integral_constant<int, 0>{}); read_part(S0, 0, 0, std::integral_constant<int, 1>{});
        read_part(S0, 0, 0, std::integral_constant<int, 2>{}); read_part(S0, 0, 0, std::integral_constant<int, 3>{});
        u32x4 f0 = read_factors(0, 0);
        for (int s = 0; s < steps; ++s) {
            const int buf = s & 1;
            const int s2 = s + 2 < steps ? s + 2 : steps - 1;      // (past the end: the last block again -- valid memory, unused)
            uint32_t rs2 = rs_load(s2);
            scale(S0, f0);                                          // (the compiler waits for S0's reads here: they flew under the
            __builtin_amdgcn_sched_barrier(0);                      //  previous half step's MFMAs)
            compute_and_read(S0, S1, buf, 1, std::false_type{}, 0, 0);
            const u32x4 f1 = read_factors(buf, 1);
            // every read of this buffer has returned (S1 complete), this wave's share of the next buffer has landed
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(rs2));
            scale(S1, f1);
            __syncthreads();
            factors_write(rs2, buf);
            __builtin_amdgcn_sched_barrier(0);
            compute_and_read(S1, S0, buf ^ 1, 0, std::true_type{}, s2, buf);   // (after the last step: stale LDS, never used)
            // issued HERE, behind the reads it belongs to (asm: left to itself the compiler sinks this loop-carried load to the top
            // of the next iteration, in front of the scale that needs it); it returns before them (LDS answers in order), and the
            // compiler's wait for S0's reads at the loop top covers it
            asm volatile("ds_read_b128 %0, %1" : "=v"(f0) : "v"((uint32_t)(kFtab + (buf ^ 1) * 64 + 16 * lh)) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the clamped tail pieces
    }
    const float ginv = 1.0f / a.gscale;
    float* part = E.partial + (size_t)split * 257 * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n = wn * 64 + c * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = wk * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                part[(size_t)k * 256 + n] = acc[i][c][r] * ginv;
            }
        }
    if (wk == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float v = cs[q] + __shfl_xor(cs[q], 32);         // the two row halves of a column
            if (lh == 0) part[(size_t)256 * 256 + wn * 64 + q * 32 + li] = v * ginv;
        }
    }
}

// ---- variant 2: 16-row steps in a FOUR-slot ring (96 KB in flight instead of 64), one barrier per 16 rows, the slab's row
// factors converted once into an LDS table (no vector-memory instruction the compiler knows about inside the loop) ----
constexpr int kSlotBytes = 32 * 1024;              // A 16 KB | G 16 KB of one 16-row step
constexpr int kR4Ftab = 4 * kSlotBytes;            // fp16 factors of the whole slab (<= 8192 rows)
constexpr int kR4MaxRows = 8192;
constexpr int kR4LdsBytes = kR4Ftab + kR4MaxRows * 2;

template <int PASSES>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_ring4_kernel(const Args a) {
    const int per_entry = a.splits;
    const int ent = blockIdx.x / per_entry, split = blockIdx.x % per_entry;
    if (ent >= a.n_entries) return;
    const Entry E = a.e[ent];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wk = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    const long long row0 = (long long)split * a.rows_per_split;
    const int hsteps = (int)(a.rows_per_split / 16);           // 16-row steps

    for (int r = t; r < (int)a.rows_per_split; r += 512) {
        const float f = fminf(__uint_as_float((uint32_t)E.rs[row0 + r] << 16) * a.gscale, 32768.0f);
        reinterpret_cast<_Float16*>(smem + kR4Ftab)[r] = (_Float16)f;
    }
    const int d_rq = lane >> 4, d_g = (lane >> 3) & 1, d_h = (lane >> 2) & 1, d_q = lane & 3;
    const uint32_t d_lane_off = (uint32_t)((2 * wave + d_g) * 2048 + d_h * 512 + (4 * d_rq + d_q) * 16);
    // piece j (0..3) of this wave's share of 16-row step hs: operand j / 2, plane j % 2
    auto dma_one = [&](int hs, auto jc) {
        constexpr int j = decltype(jc)::value, op = j >> 1, pl = j & 1;
        const size_t blk = (size_t)(row0 / kStepRows + (hs >> 1)) * kOpBytes;
        const uint64_t bv = (uint64_t)((op ? E.G : E.A) + blk);
        const char* base = (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(bv >> 32)) << 32) |
                                         (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bv));
        dma_piece(base, d_lane_off + pl * 1024 + (hs & 1) * 256,
                  (uint32_t)((hs & 3) * kSlotBytes + op * (kSlotBytes / 2) + (wave * 2 + pl) * 1024));
    };
    auto dma_all = [&](int hs) {
        dma_one(hs, std::integral_constant<int, 0>{}); dma_one(hs, std::integral_constant<int, 1>{});
        dma_one(hs, std::integral_constant<int, 2>{}); dma_one(hs, std::integral_constant<int, 3>{});
    };
    const int lane16 = lane & 15, tq = lane16 >> 2, tp = lane16 & 3, tg = (lane >> 4) & 1;
    const uint32_t tr_lane = (uint32_t)((2 * lh) * 256 + tg * 128 + (tp & 1) * 64 + tq * 16 + 8 * (tp >> 1));
    struct OpSet { u32x4 ah[4], al[4], gh[2], gl[2]; };
    auto rd = [&](uint32_t off) -> u32x4 {
        const s4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(uintptr_t)(off));
        const s4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(uintptr_t)(off + 256));
        const u32x2 x2 = __builtin_bit_cast(u32x2, x), y2 = __builtin_bit_cast(u32x2, y);
        return u32x4{x2[0], x2[1], y2[0], y2[1]};
    };
    auto read_part = [&](OpSet& S, int hs, auto ic) {
        constexpr int i = decltype(ic)::value;
        const uint32_t base = (uint32_t)((hs & 3) * kSlotBytes) + tr_lane;
        const uint32_t ca = base + (uint32_t)((wk * 4 + i) * 2) * 1024;
        S.ah[i] = rd(ca);
        S.al[i] = rd(ca + 1024);
        if constexpr (i < 2) {
            const uint32_t cg = base + kSlotBytes / 2 + (uint32_t)((wn * 2 + i) * 2) * 1024;
            S.gh[i] = rd(cg);
            S.gl[i] = rd(cg + 1024);
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;
    float cs[2] = {0.f, 0.f};
    auto scale = [&](OpSet& S, const u32x4& f) {
        const h2 one2 = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t wh = S.gh[q][d], wl = S.gl[q][d], wf = f[d];
                const h2 gh2 = __builtin_bit_cast(h2, wh) * __builtin_bit_cast(h2, wf);
                const h2 gl2 = __builtin_bit_cast(h2, wl) * __builtin_bit_cast(h2, wf);
                S.gh[q][d] = __builtin_bit_cast(uint32_t, gh2);
                S.gl[q][d] = __builtin_bit_cast(uint32_t, gl2);
                if (wk == 0) {
                    cs[q] = __builtin_amdgcn_fdot2(gh2, one2, cs[q], false);
                    cs[q] = __builtin_amdgcn_fdot2(gl2, one2, cs[q], false);
                }
            }
        }
    };
    auto mfma_part = [&](const OpSet& S, auto ic) {
        constexpr int i = decltype(ic)::value;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const h8 ah = __builtin_bit_cast(h8, S.ah[i]), al = __builtin_bit_cast(h8, S.al[i]);
            const h8 gh = __builtin_bit_cast(h8, S.gh[c]), gl = __builtin_bit_cast(h8, S.gl[c]);
            if constexpr (PASSES == 3) {
                acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gl, acc[i][c], 0, 0, 0);
                acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, gh, acc[i][c], 0, 0, 0);
            }
            acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gh, acc[i][c], 0, 0, 0);
        }
    };
    // one 16-row step: X holds step hs (read a step ago), Y receives step hs + 1, the DMA pieces of step hs + 3 go out between
    // the MFMA groups.  (Past the slab's end: clamped block index -- valid memory, never used.)
    auto half_step = [&](OpSet& X, OpSet& Y, u32x4& fx, u32x4& fy, int hs) {
        // the next step's row factors (the table is static): issued FIRST, so that every later LDS wait covers it -- the compiler
        // does not know this asm read, and a counted wait of its own for X's early parts would not include a read issued after them
        const int hn = hs + 1 < hsteps ? hs + 1 : hsteps - 1;
        asm volatile("ds_read_b128 %0, %1" : "=v"(fy) : "v"((uint32_t)(kR4Ftab + (hn * 16 + 8 * lh) * 2)) : "memory");
        scale(X, fx);                              // (the compiler waits for X's reads here: they flew under the previous step's MFMAs)
        // step hs + 1 has landed for this wave (the 4 pieces of hs + 2 may still fly); every read of slot hs - 1 has returned
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        const int h3 = hs + 3 < hsteps ? hs + 3 : hsteps - 1;
        __builtin_amdgcn_sched_barrier(0);
        mfma_part(X, std::integral_constant<int, 0>{}); __builtin_amdgcn_sched_barrier(0);
        read_part(Y, hs + 1, std::integral_constant<int, 0>{}); dma_one(h3, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_part(X, std::integral_constant<int, 1>{}); __builtin_amdgcn_sched_barrier(0);
        read_part(Y, hs + 1, std::integral_constant<int, 1>{}); dma_one(h3, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_part(X, std::integral_constant<int, 2>{}); __builtin_amdgcn_sched_barrier(0);
        read_part(Y, hs + 1, std::integral_constant<int, 2>{}); dma_one(h3, std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        mfma_part(X, std::integral_constant<int, 3>{}); __builtin_amdgcn_sched_barrier(0);
        read_part(Y, hs + 1, std::integral_constant<int, 3>{}); dma_one(h3, std::integral_constant<int, 3>{});
        __builtin_amdgcn_sched_barrier(0);
    };
    OpSet S0, S1;
    if (hsteps > 0) {
        dma_all(0);
        dma_all(hsteps > 1 ? 1 : 0);
        dma_all(hsteps > 2 ? 2 : hsteps - 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // step 0's four pieces have landed
        __syncthreads();                                      // (also publishes the factor table)
        read_part(S0, 0, std::integral_constant<int, 0>{}); read_part(S0, 0, std::integral_constant<int, 1>{});
        read_part(S0, 0, std::integral_constant<int, 2>{}); read_part(S0, 0, std::integral_constant<int, 3>{});
        u32x4 f0 = *reinterpret_cast<const u32x4*>(smem + kR4Ftab + (8 * lh) * 2), f1 = f0;
        for (int hs = 0; hs < hsteps; hs += 2) {              // (hsteps is even: 32-row blocks)
            half_step(S0, S1, f0, f1, hs);
            half_step(S1, S0, f1, f0, hs + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const float ginv = 1.0f / a.gscale;
    float* part = E.partial + (size_t)split * 257 * 256;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n = wn * 64 + c * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = wk * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                part[(size_t)k * 256 + n] = acc[i][c][r] * ginv;
            }
        }
    if (wk == 0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float v = cs[q] + __shfl_xor(cs[q], 32);
            if (lh == 0) part[(size_t)256 * 256 + wn * 64 + q * 32 + li] = v * ginv;
        }
    }
}

// ---- host: plane8 packing of an (M x 256) matrix of (hi, lo) pairs ----
static inline size_t plane8_off(long long row, int f, int plane) {     // byte offset of the HALF of feature f
    const long long blk = row / 32; const int j = (int)(row % 32);
    const int n = f / 16, c = f % 16, h = (c >> 2) & 1, e = 4 * (c >> 3) + (c & 3);
    return (size_t)blk * kOpBytes + (size_t)n * 2048 + plane * 1024 + h * 512 + j * 16 + e * 2;
}
static uint16_t f2h(float v) { _Float16 h = (_Float16)v; uint16_t b; memcpy(&b, &h, 2); return b; }
static float h2f(uint16_t b) { _Float16 h; memcpy(&h, &b, 2); return (float)h; }

int main() {
    // ---------------- correctness: 2 slabs of 96 rows ----------------
    {
        const long long M = 192; const int splits = 2;
        std::vector<char> A(M * 1024), G(M * 1024);
        std::vector<uint16_t> rs(M);
        std::vector<double> At(M * 256), Gt(M * 256);
        srand(1);
        for (long long r = 0; r < M; ++r) {
            const int ex = (rand() % 9) - 4;
            const float rf = ldexpf(1.0f, ex);
            uint32_t bits; memcpy(&bits, &rf, 4); rs[r] = (uint16_t)(bits >> 16);
            for (int f = 0; f < 256; ++f) {
                const float va = (float)rand() / RAND_MAX * 2.f - 1.f, vg = ((float)rand() / RAND_MAX * 2.f - 1.f) * 64.f;
                const uint16_t ah = f2h(va), al = f2h(va - h2f(ah)), gh = f2h(vg), gl = f2h(vg - h2f(gh));
                memcpy(&A[plane8_off(r, f, 0)], &ah, 2); memcpy(&A[plane8_off(r, f, 1)], &al, 2);
                memcpy(&G[plane8_off(r, f, 0)], &gh, 2); memcpy(&G[plane8_off(r, f, 1)], &gl, 2);
                At[r * 256 + f] = (double)h2f(ah) + (double)h2f(al);
                Gt[r * 256 + f] = ((double)h2f(gh) + (double)h2f(gl)) * rf;
            }
        }
        char *dA, *dG; uint16_t* drs; float* dp;
        HIP_OK(hipMalloc(&dA, A.size())); HIP_OK(hipMalloc(&dG, G.size())); HIP_OK(hipMalloc(&drs, M * 2));
        HIP_OK(hipMalloc(&dp, (size_t)splits * 257 * 256 * 4));
        HIP_OK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dG, G.data(), G.size(), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(drs, rs.data(), M * 2, hipMemcpyHostToDevice));
        Args a{}; a.n_entries = 1; a.splits = splits; a.rows_per_split = M / splits; a.gscale = 1.0f / 4.0f;
        a.e[0] = Entry{dA, dG, drs, dp};
        HIP_OK(hipFuncSetAttribute((const void*)wgrad_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        HIP_OK(hipFuncSetAttribute((const void*)wgrad_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        hipLaunchKernelGGL(wgrad_kernel<3>, dim3(splits), dim3(512), kLdsBytes, 0, a);
        HIP_OK(hipDeviceSynchronize());
        std::vector<float> p((size_t)splits * 257 * 256);
        HIP_OK(hipMemcpy(p.data(), dp, p.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, wb = 0, mx = 0;
        for (int k = 0; k < 256; ++k)
            for (int n = 0; n < 256; ++n) {
                double ref = 0;
                for (long long r = 0; r < M; ++r) ref += At[r * 256 + k] * Gt[r * 256 + n];
                const double got = (double)p[(size_t)k * 256 + n] + (double)p[(size_t)257 * 256 + (size_t)k * 256 + n];
                worst = fmax(worst, fabs(got - ref)); mx = fmax(mx, fabs(ref));
            }
        for (int n = 0; n < 256; ++n) {
            double ref = 0;
            for (long long r = 0; r < M; ++r) ref += Gt[r * 256 + n];
            const double got = (double)p[(size_t)256 * 256 + n] + (double)p[(size_t)257 * 256 + (size_t)256 * 256 + n];
            wb = fmax(wb, fabs(got - ref));
        }
        printf("check (192 rows, 2 slabs): max |dW - ref| = %.3e of max|dW| %.3e (%.2e relative), bias row max error %.3e\n", worst, mx,
               worst / mx, wb);
        if (!(worst / mx < 1e-5)) { printf("FAILED\n"); return 1; }
        // the four-slot ring variant must give the same sums (same products, same order per accumulator)
        HIP_OK(hipFuncSetAttribute((const void*)wgrad_ring4_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, kR4LdsBytes));
        HIP_OK(hipFuncSetAttribute((const void*)wgrad_ring4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kR4LdsBytes));
        HIP_OK(hipMemset(dp, 0, p.size() * 4));
        hipLaunchKernelGGL(wgrad_ring4_kernel<3>, dim3(splits), dim3(512), kR4LdsBytes, 0, a);
        HIP_OK(hipDeviceSynchronize());
        std::vector<float> p4(p.size());
        HIP_OK(hipMemcpy(p4.data(), dp, p4.size() * 4, hipMemcpyDeviceToHost));
        double d4 = 0;
        for (size_t i = 0; i < p.size(); ++i) d4 = fmax(d4, fabs((double)p4[i] - (double)p[i]));
        printf("check: four-slot ring variant vs the two-buffer variant: max difference %.3e\n", d4);
        if (!(d4 <= 1e-6 * mx)) { printf("FAILED\n"); return 1; }
        hipFree(dA); hipFree(dG); hipFree(drs); hipFree(dp);
    }
    // ---------------- timing: the trainer's fine pass (8 layers x 524288 rows, 64 slabs per layer) ----------------
    {
        const long long M = 524288; const int splits = 64, L = 8;
        char* buf; uint16_t* drs; float* dp;
        const size_t opb = (size_t)M * 1024;
        HIP_OK(hipMalloc(&buf, opb * 2 * L)); HIP_OK(hipMalloc(&drs, M * 2)); HIP_OK(hipMalloc(&dp, (size_t)L * splits * 257 * 256 * 4));
        // operands: fp16 values around 1 (hi) and 1e-3 (lo); row factors 1
        std::vector<uint16_t> pat(1 << 20);
        for (size_t i = 0; i < pat.size(); ++i) pat[i] = f2h(((i >> 3) & 64 ? 1e-3f : 1.0f) * ((float)(rand() % 2048) / 1024.f - 1.f));
        for (size_t o = 0; o < opb * 2 * L; o += pat.size() * 2) HIP_OK(hipMemcpy(buf + o, pat.data(), pat.size() * 2, hipMemcpyHostToDevice));
        std::vector<uint16_t> ones(M, 0x3F80);
        HIP_OK(hipMemcpy(drs, ones.data(), M * 2, hipMemcpyHostToDevice));
        Args a{}; a.n_entries = L; a.splits = splits; a.rows_per_split = M / splits; a.gscale = 1.0f;
        for (int l = 0; l < L; ++l) a.e[l] = Entry{buf + (size_t)(2 * l) * opb, buf + (size_t)(2 * l + 1) * opb, drs, dp + (size_t)l * splits * 257 * 256};
        hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
        HIP_OK(hipFuncSetAttribute((const void*)wgrad_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        for (int passes = 3; passes >= 0; passes -= (passes == 3 ? 2 : 1)) {
            auto launch = [&]() {
                if (passes == 3) hipLaunchKernelGGL(wgrad_kernel<3>, dim3(L * splits), dim3(512), kLdsBytes, 0, a);
                else if (passes == 1) hipLaunchKernelGGL(wgrad_kernel<1>, dim3(L * splits), dim3(512), kLdsBytes, 0, a);
                else hipLaunchKernelGGL((wgrad_kernel<3, true>), dim3(L * splits), dim3(512), kLdsBytes, 0, a);
            };
            launch(); HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipEventRecord(e0));
            const int reps = 5;
            for (int i = 0; i < reps; ++i) launch();
            HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1));
            float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            if (passes == 0) printf("(next line: 3 passes WITHOUT the DMA -- the compute side alone)\n");
            printf("%d MFMA pass(es): %8.1f us per launch (8 layers x 524288 rows): %.2f TB/s of operands  [gemm_atb_p<256> today: ~1730 us, 4.95 TB/s; 1-pass build 5.96 TB/s]\n",
                   passes, ms * 1e3, (double)opb * 2 * L / (ms * 1e-3) / 1e12);
        }
        for (int passes = 3; passes >= 1; passes -= 2) {
            auto launch = [&]() {
                if (passes == 3) hipLaunchKernelGGL(wgrad_ring4_kernel<3>, dim3(L * splits), dim3(512), kR4LdsBytes, 0, a);
                else hipLaunchKernelGGL(wgrad_ring4_kernel<1>, dim3(L * splits), dim3(512), kR4LdsBytes, 0, a);
            };
            launch(); HIP_OK(hipDeviceSynchronize());
            HIP_OK(hipEventRecord(e0));
            const int reps = 5;
            for (int i = 0; i < reps; ++i) launch();
            HIP_OK(hipEventRecord(e1)); HIP_OK(hipEventSynchronize(e1));
            float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("four-slot ring, %d MFMA pass(es): %8.1f us per launch: %.2f TB/s of operands\n", passes, ms * 1e3,
                   (double)opb * 2 * L / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}

// What does each ingredient of the fused f16x3 kernel cost under the chip's power cap?  A ladder of loops, each adding one
// ingredient of mlp_f16x3_kernel's k-step at the kernel's own ratio, all on random fp16 operands, one wave per SIMD on
// every CU, ~0.5 s each; prints the MFMA work rate (PFLOP/s of fp16 MFMA) -- run it under
//   rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
// for the clock each rung holds (tools/mfma_ladder_summary.py -> profiles/r2_mfma_energy_ladder.json).
//   rung 0  bare: 3 MFMAs per k-step (hi*lo, lo*hi, hi*hi), operands in registers
//   rung 1  + the A fragments come from LDS: 2 ds_read_b128 per k-step, the four waves reading the same 2 KiB
//   rung 2  + the L2 -> LDS weight stream by LDS-DMA: 1 KiB per wave per 2 k-steps out of a 2 MB L2-resident buffer
//   rung 3  + the epilogue's VALU share: ~10 plain VALU ops per k-step
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int kRing = 128 * 1024;

template <int RUNG>
__global__ __launch_bounds__(256, 1) void ladder(float* out, const _Float16* in, const char* wstream, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kRing / 16; i += 256)
        reinterpret_cast<f32x4*>(smem)[i] = reinterpret_cast<const f32x4*>(wstream)[i];
    __syncthreads();
    h8 bh[4], bl[4], ah, al;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            bh[i][e] = in[(tid * 64 + i * 8 + e) & 0xFFFFF];
            bl[i][e] = in[(tid * 64 + 32 + i * 8 + e) & 0xFFFFF] * (_Float16)0.001f;
        }
    for (int e = 0; e < 8; ++e) { ah[e] = in[(tid * 8 + e + 4096) & 0xFFFFF]; al[e] = ah[e] * (_Float16)0.001f; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float vsink = (float)tid;
    uint32_t dma_src = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 16; ++n) {           // one output tile: 16 k-steps on accumulator (n & 3)... rotating over 4
            f32x16& a = acc[n & 3];
            if (RUNG >= 1) {
                const uint32_t off = ((uint32_t)(it * 16 + n) * 2048u) & (kRing - 1);
                ah = *reinterpret_cast<const h8*>(smem + off + lane * 16);
                al = *reinterpret_cast<const h8*>(smem + off + 1024 + lane * 16);
            }
            if (RUNG >= 2 && (n & 1) == 0) {
                const uint32_t dst = __builtin_amdgcn_readfirstlane((((uint32_t)(it * 16 + n) * 512u + wave * 1024u) & (kRing - 1)));
                const uint32_t voff = dma_src + lane * 16;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             : : "v"(voff), "s"(wstream), "s"(dst) : "memory");
                dma_src = (dma_src + 4096u) & (2u * 1024 * 1024 - 1);
            }
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[n & 3], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[n & 3], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[n & 3], a, 0, 0, 0);
            if (RUNG >= 3) {
#pragma unroll
                for (int v = 0; v < 5; ++v) { vsink = vsink * 1.0001f + 0.5f; vsink = fmaxf(vsink, 0.05f * vsink); }
            }
        }
        if (RUNG >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = vsink;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + tid] = s;
}

template <int RUNG>
void run(float* out, _Float16* in, char* ws, const char* what) {
    const double flop_per_iter = 48.0 * 32768.0;       // 16 k-steps x 3 MFMAs
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(ladder<RUNG>), hipFuncAttributeMaxDynamicSharedMemorySize, kRing);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((ladder<RUNG>), dim3(256), dim3(256), kRing, 0, out, in, ws, iters);
    hipDeviceSynchronize();
    const int reps = 60;
    hipEventRecord(e0);
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL((ladder<RUNG>), dim3(256), dim3(256), kRing, 0, out, in, ws, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double pf = flop_per_iter * iters * 1024.0 * reps / (ms * 1e-3) / 1e15;
    printf("rung %d (%s): %.1f ms, %.3f PFLOP/s of fp16 MFMA work = %.3f of the 2.5 PF peak\n", RUNG, what, ms, pf, pf / 2.5);
}

int main() {
    float* out; _Float16* in; char* ws;
    hipMalloc(&out, 65536 * 4); hipMalloc(&in, (1 << 20) * 2); hipMalloc(&ws, 2 * 1024 * 1024);
    std::vector<_Float16> h(1 << 20);
    uint32_t x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (_Float16)(((x >> 8) & 0xFFFF) / 65536.0f - 0.5f); }
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(ws, h.data(), 2 * 1024 * 1024, hipMemcpyHostToDevice);
    run<0>(out, in, ws, "bare MFMA, operands in registers");
    run<1>(out, in, ws, "+ A fragments from LDS");
    run<2>(out, in, ws, "+ L2 -> LDS weight stream (LDS-DMA)");
    run<3>(out, in, ws, "+ epilogue VALU");
    run<0>(out, in, ws, "bare MFMA again (thermal state check)");
    return 0;
}

// Microbenchmark: what one VALU instruction of a given kind costs beside v_mfma_f32_32x32x2_f32
// (2 alternating accumulators, one wave per SIMD).  hipcc --offload-arch=gfx950 -O3 valu_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NV>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* in, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc0, acc1;
    for (int j = 0; j < 16; ++j) { acc0[j] = in[j]; acc1[j] = in[j + 1]; }
    float b[8], a[8];
    for (int i = 0; i < 8; ++i) { b[i] = in[threadIdx.x + i * 256]; a[i] = in[lane + i]; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[lane + 16 + i];
    f32x2 pv[4];
    for (int i = 0; i < 4; ++i) { pv[i][0] = v[2 * i]; pv[i][1] = v[2 * i + 1]; }
    const float alpha = in[5];
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[(q + 1) & 7], acc1, 0, 0, 0);
#pragma unroll
            for (int w = 0; w < NV; ++w) {
                const int i = (q * NV + w) & 7;
                if (KIND == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "v"(alpha));          // independent mul
                if (KIND == 1) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[i]) : "v"(alpha));
                if (KIND == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(alpha));
                if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(pv[i & 3]) : "v"(pv[(i + 1) & 3]));
                if (KIND == 4) asm volatile("v_accvgpr_write_b32 a200, %0" ::"v"(v[i]) : "a200");
                if (KIND == 5) asm volatile("v_accvgpr_read_b32 %0, a201" : "=v"(v[i]) :: );
                if (KIND == 6) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(v[i]) : "v"(alpha));
                if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(alpha));
                if (KIND == 8) asm volatile("s_nop 0");
                if (KIND == 9) asm volatile("v_and_b32 %0, %1, %0" : "+v"(v[i]) : "v"(alpha));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += pv[i][0] + pv[i][1];
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NV>
void run(const char* name, float* out, float* in, unsigned long long* cyc) {
    const int iters = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KIND, NV>), dim3(256), dim3(256), 0, 0, out, in, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per_pair = (double)c / (iters * 8.0);
    printf("%-26s NV=%d  %.1f cycles per MFMA pair  => %.2f cycles per extra instr\n", name, NV, per_pair,
           NV ? (per_pair - 128.2) / NV : 0.0);
}

int main() {
    float *out, *in; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&in, 1 << 20); (void)hipMalloc(&cyc, 64);
    (void)hipMemset(in, 0, 1 << 20);
    run<0, 0>("baseline", out, in, cyc);
    run<0, 2>("v_mul_f32", out, in, cyc);  run<0, 6>("v_mul_f32", out, in, cyc);
    run<1, 2>("v_max_f32", out, in, cyc);  run<1, 6>("v_max_f32", out, in, cyc);
    run<2, 2>("v_mov_b32", out, in, cyc);  run<2, 6>("v_mov_b32", out, in, cyc);
    run<3, 2>("v_pk_mul_f32", out, in, cyc); run<3, 6>("v_pk_mul_f32", out, in, cyc);
    run<4, 2>("v_accvgpr_write", out, in, cyc); run<4, 6>("v_accvgpr_write", out, in, cyc);
    run<5, 2>("v_accvgpr_read", out, in, cyc); run<5, 6>("v_accvgpr_read", out, in, cyc);
    run<6, 2>("v_fma_f32", out, in, cyc); run<6, 6>("v_fma_f32", out, in, cyc);
    run<7, 2>("v_cndmask_b32", out, in, cyc); run<7, 6>("v_cndmask_b32", out, in, cyc);
    run<8, 2>("s_nop 0", out, in, cyc); run<8, 6>("s_nop 0", out, in, cyc);
    run<9, 2>("v_and_b32", out, in, cyc); run<9, 6>("v_and_b32", out, in, cyc);
    return 0;
}

// Microbenchmark: cycles per v_mfma_f32_32x32x2_f32 for different stream shapes (one wave per SIMD,
// 4 waves per workgroup, one workgroup per CU).  Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) char smem[];

template <int NACC, int LDS, int VALU>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* in, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = in[j + i];
    float b[16];
    for (int i = 0; i < 16; ++i) b[i] = in[threadIdx.x + i * 256];
    for (int i = threadIdx.x; i < 16384; i += 256) ((float*)smem)[i] = in[i & 1023];
    __syncthreads();
    float a0[4] = {in[lane], in[lane + 1], in[lane + 2], in[lane + 3]};
    float v0 = in[lane], v1 = in[lane + 7];
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    f32x4 a_nx = *(const f32x4*)(smem + lane * 16);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            f32x4 a4;
            if (LDS) { a4 = a_nx; a_nx = *(const f32x4*)(smem + lane * 16 + ((q + 1) & 15) * 1024); }
            else { a4[0] = a0[0]; a4[1] = a0[1]; a4[2] = a0[2]; a4[3] = a0[3]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int n = 0; n < NACC; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], b[(q * 4 + e) & 15], acc[n], 0, 0, 0);
#pragma unroll
                for (int w = 0; w < VALU; ++w) { v0 = fmaf(v0, v1, 1.0f); }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = v0;
    for (int n = 0; n < NACC; ++n) for (int j = 0; j < 16; ++j) s += acc[n][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int LDS, int VALU>
void run(const char* name, float* out, float* in, unsigned long long* cyc) {
    const int iters = 64;
    hipFuncSetAttribute((const void*)k<NACC, LDS, VALU>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<NACC, LDS, VALU>), dim3(256), dim3(256), 65536, 0, out, in, cyc, iters);
        hipDeviceSynchronize();
    }
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %.2f cycles/MFMA\n", name, (double)c / (iters * 64.0 * NACC));
}

int main() {
    float *out, *in; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&in, 1 << 20); hipMalloc(&cyc, 64);
    hipMemset(in, 0, 1 << 20);
    run<1, 0, 0>("1 acc, regs", out, in, cyc);
    run<2, 0, 0>("2 acc, regs", out, in, cyc);
    run<4, 0, 0>("4 acc, regs", out, in, cyc);
    run<1, 1, 0>("1 acc, lds A", out, in, cyc);
    run<2, 1, 0>("2 acc, lds A", out, in, cyc);
    run<2, 1, 2>("2 acc, lds A, 2 valu/mfma", out, in, cyc);
    run<2, 1, 6>("2 acc, lds A, 6 valu/mfma", out, in, cyc);
    run<2, 1, 10>("2 acc, lds A, 10 valu/mfma", out, in, cyc);
    run<1, 1, 4>("1 acc, lds A, 4 valu/mfma", out, in, cyc);
    return 0;
}

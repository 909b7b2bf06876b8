// Microbenchmark: which ingredient of a tiled fp32-MFMA GEMM inner loop costs MFMA issue slots on gfx950?
// One iteration = 32 v_mfma_f32_32x32x2_f32 per wave (4 accumulators) -- the k-step of train_kernels.hip's GEMMs --
// optionally with: 32 ds_read_b32 operand loads, 16 ds_write_b32, a workgroup barrier, 4 global float4 loads.
// 4 waves per workgroup, 1..3 workgroups per CU (occupancy set through the LDS allocation).
// Prints cycles per MFMA per SIMD (ideal 64) assuming 2.4 GHz.  Build: hipcc --offload-arch=gfx950 -O3 gemm_loop_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
extern __shared__ __attribute__((aligned(16))) float smem[];

template <int RD, int WR, int BAR, int GL>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, size_t gstride) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 8192; i += 256) smem[i] = in[i & 1023];
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float av[2][8], bv[2][8];
    for (int j = 0; j < 8; ++j) { av[0][j] = in[lane + j]; av[1][j] = in[lane + 8 + j]; bv[0][j] = in[lane + 16 + j]; bv[1][j] = in[lane + 24 + j]; }
    const float4* gp = reinterpret_cast<const float4*>(in) + (size_t)blockIdx.x * 4096 + t;
    float4 r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
    const int li = lane & 31, lh = lane >> 5;
    for (int it = 0; it < iters; ++it) {
        if (GL) {
            const float4* q = gp + (size_t)(it & 63) * gstride;
            r0 = q[0]; r1 = q[256]; r2 = q[512]; r3 = q[768];
        }
        if (RD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                av[0][j] = smem[(8 * lh + j) * 132 + wave * 32 + li];
                av[1][j] = smem[(8 * lh + j) * 132 + 64 + li];
                bv[0][j] = smem[2112 + (8 * lh + j) * 132 + wave * 32 + li];
                bv[1][j] = smem[2112 + (8 * lh + j) * 132 + 64 + li];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a * 2 + b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][j], bv[b][j], acc[a * 2 + b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (WR) {
            float* w = smem + 4224 + (it & 1) * 1024;
            if (GL) {
                w[t] = r0.x; w[t + 256] = r1.y; w[t + 512] = r2.z; w[t + 768] = r3.w;
                w[t + 1] = r0.y; w[t + 257] = r1.z; w[t + 513] = r2.w; w[t + 769] = r3.x;
                w[t + 2] = r0.z; w[t + 258] = r1.w; w[t + 514] = r2.x; w[t + 770] = r3.y;
                w[t + 3] = r0.w; w[t + 259] = r1.x; w[t + 515] = r2.y; w[t + 771] = r3.z;
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) w[t + 64 * q] = av[q & 1][q & 7];
            }
        }
        if (BAR) __syncthreads();
    }
    float s = r0.x + r1.x + r2.x + r3.x;
    for (int n = 0; n < 4; ++n) for (int j = 0; j < 16; ++j) s += acc[n][j];
    out[(size_t)blockIdx.x * 256 + t] = s;
}

template <int RD, int WR, int BAR, int GL>
void run(const char* name, float* out, float* in, int wg_per_cu) {
    const int iters = 2000;
    const size_t lds = wg_per_cu == 1 ? 100 * 1024 : wg_per_cu == 2 ? 70 * 1024 : 48 * 1024;   // forces the occupancy
    (void)hipFuncSetAttribute((const void*)k<RD, WR, BAR, GL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    hipLaunchKernelGGL((k<RD, WR, BAR, GL>), dim3(grid), dim3(256), lds, 0, out, in, 10, (size_t)grid * 4096);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<RD, WR, BAR, GL>), dim3(grid), dim3(256), lds, 0, out, in, iters, (size_t)grid * 4096);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wg_per_cu waves, each iters * 32 MFMAs
    const double mfma_per_simd = (double)wg_per_cu * iters * 32;
    printf("%-44s %d WG/CU  %7.3f ms  %6.1f cycles/MFMA/SIMD @2.4GHz  (util %.2f)\n", name, wg_per_cu, ms,
           ms * 1e-3 * 2.4e9 / mfma_per_simd, mfma_per_simd * 64 / (ms * 1e-3 * 2.4e9));
}

int main() {
    float *out, *in;
    const size_t nin = (size_t)64 * 768 * 4096 * 4 + 4096;     // 64 k-steps x 768 workgroups x 64 KB
    (void)hipMalloc(&in, nin * sizeof(float));
    (void)hipMalloc(&out, (size_t)768 * 256 * sizeof(float));
    (void)hipMemset(in, 0, nin * sizeof(float));
    for (int w = 1; w <= 3; ++w) {
        run<0, 0, 0, 0>("mfma only", out, in, w);
        run<1, 0, 0, 0>("+ 32 ds_read_b32 operands", out, in, w);
        run<1, 0, 1, 0>("+ reads + barrier", out, in, w);
        run<1, 1, 1, 0>("+ reads + 16 ds_write_b32 + barrier", out, in, w);
        run<1, 1, 1, 1>("+ reads + writes + barrier + 4 global float4", out, in, w);
        run<0, 0, 1, 0>("mfma + barrier only", out, in, w);
        run<0, 0, 0, 1>("mfma + global loads only", out, in, w);
    }
    return 0;
}

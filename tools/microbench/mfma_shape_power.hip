// Which fp16 MFMA shape does the chip sustain more FLOP/s on under its power cap?  Bare MFMA loops on random
// operands in registers, one wave per SIMD on every CU, ~1 s per shape, wall-clock TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void k(float* out, const _Float16* in, int iters) {
    const int t = threadIdx.x + blockIdx.x * 256;
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = in[(t * 64 + i * 8 + e) & 0xFFFFF]; b[i][e] = in[(t * 64 + 32 + i * 8 + e) & 0xFFFFF]; }
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[r], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + r) & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    }
    out[t] = s;
}

template <int SHAPE>
void run(float* out, _Float16* in) {
    // per iteration: SHAPE 32: 16 MFMAs x 32768 flop; SHAPE 16: 32 MFMAs x 16384 flop  -> equal flops
    const double flop_per_iter = 16.0 * 32768.0;
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<SHAPE>), dim3(256), dim3(256), 0, 0, out, in, iters);
    hipDeviceSynchronize();
    const int reps = 40;
    hipEventRecord(e0);
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL((k<SHAPE>), dim3(256), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = flop_per_iter * iters * 1024.0 * reps / (ms * 1e-3) / 1e12;
    printf("%dx%d: %.1f ms, %.1f TFLOP/s\n", SHAPE, SHAPE, ms, tf);
}

int main() {
    float* out; _Float16* in;
    hipMalloc(&out, 65536 * 4); hipMalloc(&in, (1 << 20) * 2);
    std::vector<_Float16> h(1 << 20);
    srand(1);
    for (auto& v : h) v = (_Float16)((float)rand() / RAND_MAX * 2.f - 1.f);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int round = 0; round < 2; ++round) { run<32>(out, in); run<16>(out, in); }
    return 0;
}

// How fast can the fused training kernels' store pattern go by itself?  The stash forward / backward chain write one
// 1 KiB row per sample with the sample on the lane: an instruction stores 16 B per lane at a 1 KiB lane stride (two lane
// halves share a 32 B piece of a line).  Pattern A replays exactly that (8 tiles x 4 quads per "layer", 9 layers), pattern
// B writes the same bytes as full 128-B lines (8 lanes per row).  No compute: the GB/s each sustains on 7.2 GB.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(256, 1) void k(float* buf, long long rows, int layers) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const long long ntiles = rows / 128;
    const f32x4 v = {1.f, 2.f, 3.f, (float)tid};
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int l = 0; l < layers; ++l) {
            float* base = buf + (size_t)l * rows * 256;
            if (PATTERN == 0) {
                float* row = base + (tile * 128 + wave * 32 + j) * 256 + 4 * h;
#pragma unroll
                for (int ut = 0; ut < 8; ++ut)
#pragma unroll
                    for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(row + 32 * ut + 8 * g) = v;
            } else {
                // 32 rows x 1 KiB per wave as 32 instructions of 8 rows x 128 B... here: each instruction = 8 rows x 32 floats
                float* t0 = base + (tile * 128 + wave * 32) * 256;
#pragma unroll
                for (int r8 = 0; r8 < 4; ++r8)
#pragma unroll
                    for (int ut = 0; ut < 8; ++ut)
                        *reinterpret_cast<f32x4*>(t0 + (r8 * 8 + (lane >> 3)) * 256 + 32 * ut + 4 * (lane & 7)) = v;
            }
        }
    }
}

template <int PATTERN>
void run(float* buf, long long rows, int layers, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<PATTERN>), dim3(256), dim3(256), 0, 0, buf, rows, layers);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<PATTERN>), dim3(256), dim3(256), 0, 0, buf, rows, layers);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)rows * 1024.0 * layers * reps;
    printf("%s: %.2f ms per launch, %.2f TB/s\n", what, ms / reps, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const long long rows = 786432;      // one 4096-ray training step: 262144 coarse + 524288 fine rows
    const int layers = 9;
    float* buf;
    if (hipMalloc(&buf, (size_t)rows * 1024 * layers) != hipSuccess) { printf("alloc failed\n"); return 1; }
    run<0>(buf, rows, layers, "A  one row per lane (16 B per lane at a 1 KiB stride): the fused kernels' pattern");
    run<1>(buf, rows, layers, "B  full 128-B lines (8 lanes per row)");
    run<0>(buf, rows, layers, "A  again");
    return 0;
}

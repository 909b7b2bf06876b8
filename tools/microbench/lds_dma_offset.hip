// Does the instruction offset of global_load_lds_dwordx4 apply to the LDS address, the global address, or both?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
extern __shared__ __attribute__((aligned(16))) char smem[];
__global__ void k(const float* src, float* out) {
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) ((float*)smem)[i] = -1.f;
    __syncthreads();
    const unsigned voff = lane * 16;
    unsigned m0v = 0;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\ts_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(src), "s"(m0v) : "memory");
    __syncthreads();
    for (int i = lane; i < 2048; i += 64) out[i] = ((float*)smem)[i];
}
int main() {
    float *src, *out;
    hipMalloc(&src, 16384); hipMalloc(&out, 8192);
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, src, out);
    std::vector<float> o(2048);
    hipMemcpy(o.data(), out, 8192, hipMemcpyDeviceToHost);
    int first = -1; for (int i = 0; i < 2048; ++i) if (o[i] >= 0) { first = i; break; }
    printf("first written LDS float index %d (byte %d), value there %.0f (global float index)\n", first, first * 4, first >= 0 ? o[first] : -1.f);
    printf("=> offset applies to LDS: %s, to global: %s\n", first * 4 == 1024 ? "yes" : "no", (first >= 0 && o[first] == 256.f) ? "yes" : "no");
    return 0;
}

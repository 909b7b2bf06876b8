// Microbenchmark: what ONE vector-memory instruction costs a wave that runs a dependent chain of
// v_mfma_f32_32x32x16_f16 (one wave per SIMD, 256 workgroups of 4 waves): stores of the trainer's shapes in their
// address forms, and the two ways of feeding the LDS weight ring (LDS-DMA vs load-to-register + ds_write_b128).
// Output: cycles per MFMA and the extra cycles per memory instruction over the bare chain (s_memtime of wave 0 of
// workgroup 0, whole-kernel time beside it).   Build: make -C tools/microbench mem_issue_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
extern __shared__ __attribute__((aligned(16))) char smem[];

enum { NONE, ST2_V, ST4_V, ST2_S, ST4_S, ST2_V_PLAIN, DMA, LOAD_DSWRITE, LOAD_ONLY };

template <int KIND, int PER>
__global__ __launch_bounds__(256, 1) void k(char* out, const char* wsrc, const float* in, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc;
    for (int j = 0; j < 16; ++j) acc[j] = in[j];
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (_Float16)in[lane + i + e]; b[i][e] = (_Float16)in[threadIdx.x + i * 8 + e]; }
    u32x4 d4 = {1u + lane, 2u, 3u, 4u};
    u32x2 d2 = {1u + lane, 2u};
    // this wave's output region: n_ops x 1 KiB, streamed
    const size_t n_ops = (size_t)iters * 8 / PER;
    char* region = out + ((size_t)blockIdx.x * 4 + wave) * n_ops * 1024;
    constexpr int kLaneBytes = (KIND == ST2_V || KIND == ST2_S || KIND == ST2_V_PLAIN) ? 8 : 16;    // contiguous per wave-instruction
    unsigned long long vaddr = (unsigned long long)region + lane * kLaneBytes;     // per-lane 64-bit address (vaddr forms)
    unsigned int voff = lane * kLaneBytes;                                        // per-lane 32-bit offset (saddr forms)
    const char* sbase = region;                                           // wave-uniform base
    u32x4 stage[4];
    for (int i = 0; i < 4; ++i) stage[i] = d4;
    unsigned int lds_dst = wave * 4096;
    unsigned int wsoff = wave * 4096 + lane * 16;
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    int op = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q & 3], b[q & 3], acc, 0, 0, 0);
            if ((q % PER) == PER - 1) {
                if (KIND == ST2_V) asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(vaddr), "v"(d2) : "memory");
                if (KIND == ST2_V_PLAIN) asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(vaddr), "v"(d2) : "memory");
                if (KIND == ST4_V) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(vaddr), "v"(d4) : "memory");
                if (KIND == ST2_S) asm volatile("global_store_dwordx2 %0, %1, %2 nt" : : "v"(voff), "v"(d2), "s"(sbase) : "memory");
                if (KIND == ST4_S) asm volatile("global_store_dwordx4 %0, %1, %2 nt" : : "v"(voff), "v"(d4), "s"(sbase) : "memory");
                if (KIND == DMA) {
                    const char* src = wsrc + (size_t)(op & 127) * 16384;
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(wsoff), "s"(src), "s"(lds_dst + (op & 7) * 16384) : "memory");
                }
                if (KIND == LOAD_DSWRITE || KIND == LOAD_ONLY) {
                    const char* src = wsrc + (size_t)(op & 127) * 16384;
                    // the oldest staged quad goes to LDS (its load was issued three operations ago), then its registers are reloaded
                    if (KIND == LOAD_DSWRITE)
                        asm volatile("s_waitcnt vmcnt(3)\n\tds_write_b128 %0, %1" : : "v"(lds_dst + lane * 16 + (op & 7) * 16384), "v"(stage[0]) : "memory");
                    else
                        asm volatile("s_waitcnt vmcnt(3)" : : "v"(stage[0]) : "memory");
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(stage[0]) : "v"(wsoff), "s"(src) : "memory");
                    u32x4 t = stage[0]; stage[0] = stage[1]; stage[1] = stage[2]; stage[2] = stage[3]; stage[3] = t;
                }
                vaddr += 1024; sbase += 1024; ++op;
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += acc[j];
    for (int i = 0; i < 4; ++i) s += (float)stage[i][0];
    if (s == 12345.678f) ((float*)out)[threadIdx.x] = s;      // keep everything alive without touching the store stream
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static double base_cpm = 0.0;
template <int KIND, int PER>
void run(const char* name, char* out, char* wsrc, float* in, unsigned long long* cyc) {
    const int iters = 512;
    (void)hipFuncSetAttribute((const void*)k<KIND, PER>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<KIND, PER>), dim3(256), dim3(256), 131072, 0, out, wsrc, in, cyc, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double cpm = (double)c / (iters * 8.0);
    if (KIND == NONE) base_cpm = cpm;
    const bool st2 = KIND == ST2_V || KIND == ST2_S || KIND == ST2_V_PLAIN, st4 = KIND == ST4_V || KIND == ST4_S;
    const double gb = (st2 || st4) ? 256.0 * 4 * (iters * 8 / PER) * (st2 ? 512 : 1024) / 1e9 : 0.0;
    printf("%-44s 1 per %d MFMA: %6.1f cycles per MFMA, +%6.1f cycles per memory instruction; kernel %.3f ms", name, PER, cpm,
           (cpm - base_cpm) * PER, ms);
    if (gb > 0) printf(" = %.2f TB/s written", gb / ms);
    printf("\n");
    fflush(stdout);
}
#define BOTH(K, name) run<K, 8>(name, out, wsrc, in, cyc); run<K, 4>(name, out, wsrc, in, cyc); run<K, 2>(name, out, wsrc, in, cyc);
int main() {
    char *out, *wsrc; float* in; unsigned long long* cyc;
    const size_t out_bytes = (size_t)256 * 4 * (512 * 8 / 2) * 1024 + 4096;      // 2 GiB: one KiB per operation at 1 per 2 MFMAs
    if (hipMalloc(&out, out_bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMalloc(&wsrc, 128 * 16384 + 65536); (void)hipMalloc(&in, 1 << 20); (void)hipMalloc(&cyc, 64);
    (void)hipMemset(in, 0, 1 << 20); (void)hipMemset(wsrc, 0, 128 * 16384 + 65536);
    run<NONE, 4>("bare dependent MFMA chain", out, wsrc, in, cyc);
    BOTH(ST2_V, "global_store_dwordx2 vaddr nt (8 B/lane)")
    BOTH(ST2_V_PLAIN, "global_store_dwordx2 vaddr (no nt)")
    BOTH(ST4_V, "global_store_dwordx4 vaddr nt (16 B/lane)")
    BOTH(ST2_S, "global_store_dwordx2 saddr+voffset nt")
    BOTH(ST4_S, "global_store_dwordx4 saddr+voffset nt")
    BOTH(DMA, "global_load_lds_dwordx4 (LDS-DMA piece)")
    BOTH(LOAD_ONLY, "global_load_dwordx4 to registers")
    BOTH(LOAD_DSWRITE, "global_load_dwordx4 + ds_write_b128")
    return 0;
}

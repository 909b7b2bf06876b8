// Microbenchmark: cost of VALU / LDS instructions beside v_mfma_f32_32x32x16_f16, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
extern __shared__ __attribute__((aligned(16))) char smem[];

template <int KIND, int NV, int LDS>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* in, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc0;
    for (int j = 0; j < 16; ++j) acc0[j] = in[j];
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) { a[i][e] = (_Float16)in[lane + i + e]; b[i][e] = (_Float16)in[threadIdx.x + i * 8 + e]; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[lane + 16 + i];
    unsigned int hv[8];
    for (int i = 0; i < 8; ++i) hv[i] = lane + i;
    for (int i = threadIdx.x; i < 16384; i += 256) ((float*)smem)[i] = in[i & 1023];
    __syncthreads();
    const float alpha = in[5];
    unsigned long long t0, t1;
    f32x4 l0 = *(const f32x4*)(smem + lane * 16), l1 = l0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            h8 aa = a[q & 3];
            if (LDS) {
                aa = __builtin_bit_cast(h8, (q & 1) ? l1 : l0);
                if (q & 1) l1 = *(const f32x4*)(smem + lane * 16 + ((q + 2) & 15) * 1024);
                else l0 = *(const f32x4*)(smem + lane * 16 + ((q + 2) & 15) * 1024);
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa, b[q & 3], acc0, 0, 0, 0);
#pragma unroll
            for (int w = 0; w < NV; ++w) {
                const int i = (q * NV + w) & 7;
                if (KIND == 0) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "v"(alpha));
                if (KIND == 1) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[i]) : "v"(alpha));
                if (KIND == 2) asm volatile("v_accvgpr_read_b32 %0, a201" : "=v"(v[i]) ::);
                if (KIND == 3) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(hv[i]) : "v"(v[i]));
                if (KIND == 4) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[i]) : "v"(hv[i]));
                if (KIND == 5) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(hv[i]) : "v"(v[i]), "v"(v[(i + 1) & 7]));
                if (KIND == 6) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(v[i]) : "v"(alpha));
                if (KIND == 7) asm volatile("v_pack_b32_f16 %0, %1, %2" : "=v"(hv[i]) : "v"(hv[(i + 1) & 7]), "v"(hv[(i + 2) & 7]));
                if (KIND == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(alpha));
                if (KIND == 9) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(*(double*)&v[(i & 3) * 2]) : "v"(*(double*)&v[((i + 1) & 3) * 2]));
                if (KIND == 10) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(hv[i]) : "v"(hv[(i + 1) & 7]), "v"(hv[(i + 2) & 7]));
                if (KIND == 11) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(hv[i]) : "v"(hv[(i + 1) & 7]), "v"(hv[(i + 2) & 7]));
                if (KIND == 12) asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(hv[i]) : "v"(hv[(i + 1) & 7]));
                if (KIND == 13) asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "v"(v[(i + 2) & 7]));
                if (KIND == 14) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hv[i]) : "v"(v[i]), "v"(v[(i + 1) & 7]));
            }
            // KIND 20: the single-pass backward chain's epilogue of one value per MFMA, as its ISA has it (NV ignored):
            // accumulator read, LeakyReLU' select from the mask bit, scale; every second MFMA: pack, D pair, running max
            if (KIND >= 20 && KIND <= 24) {
                const int i = q & 1;
                // 22: the accumulator already in an architectural VGPR (no v_accvgpr_read); 23: also no v_pk_mul_f16;
                // 24: as 22 with LeakyReLU' applied to the packed pair (shift + v_pk_ashrrev_i16 + v_bfi per PAIR)
                if (KIND == 20 || KIND == 21) asm volatile("v_accvgpr_read_b32 %0, a201" : "=v"(v[i]) ::);
                if (KIND == 24) {
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(alpha));
                    if (q & 1) {
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hv[2]) : "v"(v[0]), "v"(v[1]));
                        asm volatile("v_lshlrev_b32 %0, 7, %1" : "=v"(hv[0]) : "v"(hv[7]));
                        asm volatile("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(hv[0]) : "v"(hv[0]));
                        asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(hv[0]) : "v"(hv[5]), "v"(hv[6]));
                        asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(hv[2]) : "v"(hv[2]), "v"(hv[0]));
                        asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(hv[3]) : "v"(hv[2]), "v"(hv[4]));
                        asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(hv[1]) : "v"(hv[2]));
                    }
                    continue;
                }
                asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(hv[i]) : "v"(hv[7]));
                asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(hv[i]) : "v"(hv[5]), "v"(hv[6]));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(hv[i]));
                if (q & 1) {
                    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hv[2]) : "v"(v[0]), "v"(v[1]));
                    if (KIND == 20 || KIND == 22) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(hv[3]) : "v"(hv[2]), "v"(hv[4]));
                    asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(v[2]) : "v"(v[0]), "v"(v[1]));
                }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = l0[0] + l1[1];
    for (int i = 0; i < 8; ++i) s += v[i] + (float)hv[i];
    for (int j = 0; j < 16; ++j) s += acc0[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND, int NV, int LDS>
void run(const char* name, float* out, float* in, unsigned long long* cyc) {
    const int iters = 512;
    (void)hipFuncSetAttribute((const void*)k<KIND, NV, LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KIND, NV, LDS>), dim3(256), dim3(256), 65536, 0, out, in, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s NV=%d LDS=%d  %.1f cycles per MFMA\n", name, NV, LDS, (double)c / (iters * 8.0));
}
#define RUN3(K, name) run<K, 2, 0>(name, out, in, cyc); run<K, 4, 0>(name, out, in, cyc); run<K, 6, 0>(name, out, in, cyc); run<K, 8, 0>(name, out, in, cyc);
int main() {
    float *out, *in; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&in, 1 << 20); (void)hipMalloc(&cyc, 64);
    (void)hipMemset(in, 0, 1 << 20);
    run<0, 0, 0>("baseline", out, in, cyc);
    run<0, 0, 1>("baseline+lds", out, in, cyc);
    RUN3(0, "v_mul_f32") RUN3(1, "v_max_f32") RUN3(2, "v_accvgpr_read") RUN3(3, "v_cvt_f16_f32") RUN3(4, "v_cvt_f32_f16")
    RUN3(5, "v_cvt_pkrtz_f16_f32") RUN3(6, "v_sub_f32") RUN3(7, "v_pack_b32_f16") RUN3(8, "v_mov_b32") RUN3(9, "v_pk_mul_f32")
    RUN3(10, "v_pk_mul_f16") RUN3(11, "v_bfi_b32") RUN3(12, "v_bfe_i32") RUN3(13, "v_max3_f32 |.|") RUN3(14, "v_cvt_pk_f16_f32")
    // the single-pass backward chain's epilogue (one value per MFMA) and cheaper formulations of it.  (No "+ lds" rows for
    // these: this file's LDS reads are compiler-scheduled two MFMAs ahead and their wait lands wherever the asm blocks
    // leave room -- 42 to 65 cycles with no pattern; the kernels read eight ahead with counted waits.)
    run<20, 0, 0>("bwd f16 epilogue mix", out, in, cyc);
    run<21, 0, 0>("  same without v_pk_mul_f16", out, in, cyc);
    run<22, 0, 0>("  accumulator in a VGPR (no accvgpr_read)", out, in, cyc);
    run<24, 0, 0>("  VGPR acc, LeakyReLU' on the packed pair", out, in, cyc);
    run<0, 4, 1>("v_mul + lds", out, in, cyc);
    return 0;
}

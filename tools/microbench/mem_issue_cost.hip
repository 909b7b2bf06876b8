// Microbenchmark: what ONE vector-memory instruction costs a wave that runs a dependent chain of
// v_mfma_f32_32x32x16_f16 (one wave per SIMD, 256 workgroups of 4 waves): stores of the trainer's shapes in their
// address forms, and the two ways of feeding the LDS weight ring (LDS-DMA vs load-to-register + ds_write_b128).
// The SKEL_* / X_* rows rebuild the fused kernels' skeleton (fragment read per MFMA, DMA piece per four, chunk wait + barrier:
// 36 cycles per MFMA) and put vector work on top: the single-pass backward's epilogue mix as the kernels issue it (one
// dependent chain per step) 66, the same instructions software-pipelined (every one reads registers written at least one
// MFMA earlier, X_EPI_SKEW) 42; six independent v_mul_f32 43, six over three registers 60.
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

enum { NONE, ST2_V, ST4_V, ST2_S, ST4_S, ST2_V_PLAIN, DMA, LOAD_DSWRITE, LOAD_ONLY, SKEL_LDS, SKEL_LDS_DMA, SKEL_FULL, SKEL_EPI, SKEL_EPI_ST, X_NODMA, X_VGPRFRAG, X_NOACCREAD, X_NOWAIT, X_PLAIN6, X_NODMA_PLAIN6, X_VGPRFRAG_NODMA, X_B64_PLAIN6, X_HALFRATE_PLAIN6, X_B32_PLAIN6, X_EARLY_PLAIN6, X_PREVSLOT_PLAIN6, X_VALUFIRST_PLAIN6, X_NOPS, X_PLAIN1, X_PLAIN3, X_SALU6, X_NOLDS_PLAIN6, X_NOLDS_IND6, X_IND6, X_BURST8, X_BURST4, X_BURST4_DMA, X_BURST4_DMA_ST, X_BURST4_EPI_DMA_ST, X_KERNELISH_BURST, X_KERNELISH_PERQ, X_EPI_SKEW, X_EPI_SKEW_ST };

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
    constexpr int kLaneBytes = (KIND == ST2_V || KIND == ST2_S || KIND == ST2_V_PLAIN || KIND >= SKEL_EPI_ST) ? 8 : 16;    // contiguous per wave-instruction
    unsigned long long vaddr = (unsigned long long)region + lane * kLaneBytes;     // per-lane 64-bit address (vaddr forms)
    unsigned int voff = lane * kLaneBytes;                                        // per-lane 32-bit offset (saddr forms)
    const char* sbase = region;                                           // wave-uniform base
    u32x4 stage[4];
    for (int i = 0; i < 4; ++i) stage[i] = d4;
    unsigned int lds_dst = wave * 4096;
    unsigned int wsoff = wave * 4096 + lane * 16;
    float ev[3] = {in[1], in[2], in[3]};
    unsigned int sdummy = 0;
    float ra[4] = {in[1], in[2], in[3], in[4]}, ry[4] = {in[1], in[2], in[3], in[4]};
    unsigned int rm[4] = {1u, 2u, 3u, 4u}, rs[4] = {1u, 2u, 3u, 4u}, rp[2] = {1u, 2u};
    float ind[6] = {in[1], in[2], in[3], in[4], in[5], in[6]};
    unsigned int eh[8];
    for (int i = 0; i < 8; ++i) eh[i] = lane + i;
    f32x4 pfq[8], pfq2[8];
    f32x16 accB = acc;
    u32x4 bop[16], bop2[8];
    for (int r = 0; r < 16; ++r) bop[r] = u32x4{(unsigned)lane + r, 2u, 3u, 4u};
    for (int r = 0; r < 8; ++r) bop2[r] = bop[r];
    const unsigned int lane16 = lane * 16;
    for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(pfq[i]) : "v"(lane16), "n"(0) : "memory");
    for (int i = 0; i < 8; ++i) pfq2[i] = pfq[i];
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    int op = 0;
    for (int it4 = 0; it4 < iters; it4 += 4)
#pragma unroll
    for (int itl = 0; itl < 4; ++itl) {       // (unrolled by 4: the kernel-like rows index registers by itl)
        const int it = it4 + itl;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (KIND >= SKEL_LDS) {
                // the fused kernels' skeleton: the A fragment of every MFMA comes from the LDS ring, read 8 MFMAs ahead
                f32x4& slot = pfq[q];
                f32x4& refill = pfq[(KIND == X_PREVSLOT_PLAIN6) ? ((q + 7) & 7) : q];
                if (KIND == X_VALUFIRST_PLAIN6) {       // vector work BEFORE the wait / read / MFMA of this step
#pragma unroll
                    for (int w = 0; w < 6; ++w) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(ev[w % 3]) : "v"(eh[5]));
                }
                if (KIND == X_EARLY_PLAIN6) {           // MFMA first, then the read, then the vector work
                    asm volatile("s_waitcnt lgkmcnt(7)" : "+a"(slot) : : "memory");
                    const h8 aa0 = __builtin_bit_cast(h8, slot);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa0, b[q & 3], acc, 0, 0, 0);
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(pfq[(q + 7) & 7]) : "v"(lane16 + ((it & 3) * 32768u)), "n"(0) : "memory");
#pragma unroll
                    for (int w = 0; w < 6; ++w) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(ev[w % 3]) : "v"(eh[5]));
                    continue;
                }
                constexpr bool vfrag = KIND == X_VGPRFRAG || KIND == X_VGPRFRAG_NODMA;
                if (KIND == X_KERNELISH_BURST || KIND == X_KERNELISH_PERQ) {
                    // closer to the kernels: the accumulator tile alternates every 16 MFMAs, the epilogue reads register
                    // (itl & 1) * 8 + q of the OTHER tile (just finished) and writes its packed pairs into the B operands the
                    // MFMAs use 16+ steps later
                    const bool odd = (itl >> 1) & 1;
                    if (KIND == X_KERNELISH_BURST) {
                        if ((q & 3) == 0) {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                f32x4& dst = (q & 4) ? pfq[r] : pfq2[r];
                                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(lane16 + (itl * 32768u)), "n"(0) : "memory");
                            }
                        }
                    }
                    f32x4& cur = KIND == X_KERNELISH_BURST ? ((q & 4) ? pfq2[q & 3] : pfq[q & 3]) : pfq[q];
                    if (KIND == X_KERNELISH_PERQ) asm volatile("s_waitcnt lgkmcnt(7)" : "+a"(cur) : : "memory");
                    else asm volatile("" : "+a"(cur));
                    const h8 aak = __builtin_bit_cast(h8, cur);
                    if (KIND == X_KERNELISH_PERQ)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(cur) : "v"(lane16 + (itl * 32768u)), "n"(0) : "memory");
                    const h8 bk = __builtin_bit_cast(h8, bop[(itl & 1) * 8 + q]);
                    float pv;
                    if (odd) { accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(aak, bk, accB, 0, 0, 0); pv = acc[(itl & 1) * 8 + q]; }
                    else { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aak, bk, acc, 0, 0, 0); pv = accB[(itl & 1) * 8 + q]; }
                    const int i = q & 1;
                    ev[i] = pv;
                    asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(eh[i]) : "v"(eh[7]));
                    asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(eh[i]) : "v"(eh[5]), "v"(eh[6]));
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(ev[i]) : "v"(eh[i]));
                    if (q & 1) {
                        unsigned int ph;
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ph) : "v"(ev[0]), "v"(ev[1]));
                        asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(d2[(q >> 1) & 1]) : "v"(ph), "v"(eh[4]));
                        asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(ev[2]) : "v"(ev[0]), "v"(ev[1]));
                        bop2[((itl & 1) * 8 + q) >> 1 & 7][(q >> 1) & 3] = ph;       // the next layer's operand, built in place
                    }
                    if ((q & 3) == 3) {
                        const char* src = wsrc + (size_t)(op & 127) * 16384;
                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(wsoff), "s"(src), "s"(lds_dst + (op & 7) * 16384) : "memory");
                        ++op;
                        asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(vaddr), "v"(d2) : "memory");
                        vaddr += 512;
                    }
                    if (q == 7 && itl == 3) {
                        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
#pragma unroll
                        for (int r = 0; r < 8; ++r) { u32x4 t = bop[r]; bop[r] = bop2[r]; bop2[r] = t; }    // (kept in place by renaming)
                    }
                    continue;
                }
                if (KIND == X_BURST8 || (KIND >= X_BURST4 && KIND <= X_BURST4_EPI_DMA_ST)) {
                    // fragment reads in BURSTS (8 or 4 at a time into the other register bank), MFMA + vector work in between
                    constexpr int B = KIND == X_BURST8 ? 8 : 4;      // (every later KIND: bursts of 4)
                    if ((q % B) == 0) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                        for (int r = 0; r < B; ++r) {
                            f32x4& dst = ((it * 8 + q) / B) & 1 ? pfq[r] : pfq2[r];
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(lane16 + ((it & 3) * 32768u)), "n"(0) : "memory");
                        }
                    }
                    f32x4& cur = ((it * 8 + q) / B) & 1 ? pfq2[q % B] : pfq[q % B];
                    asm volatile("" : "+a"(cur));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, cur), b[q & 3], acc, 0, 0, 0);
                    if (KIND == X_BURST4_EPI_DMA_ST) {      // the backward epilogue's own mix instead of six v_mul
                        const int i = q & 1;
                        asm volatile("v_accvgpr_read_b32 %0, a201" : "=v"(ev[i]) ::);
                        asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(eh[i]) : "v"(eh[7]));
                        asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(eh[i]) : "v"(eh[5]), "v"(eh[6]));
                        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(ev[i]) : "v"(eh[i]));
                        if (q & 1) {
                            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(eh[2]) : "v"(ev[0]), "v"(ev[1]));
                            asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(d2[(q >> 1) & 1]) : "v"(eh[2]), "v"(eh[4]));
                            asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(ev[2]) : "v"(ev[0]), "v"(ev[1]));
                        }
                    } else {
#pragma unroll
                        for (int w = 0; w < 6; ++w) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(ind[w]) : "v"(eh[5]));
                    }
                    if (KIND >= X_BURST4_DMA && (q & 3) == 3) {
                        const char* src = wsrc + (size_t)(op & 127) * 16384;
                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(wsoff), "s"(src), "s"(lds_dst + (op & 7) * 16384) : "memory");
                        ++op;
                        if (KIND >= X_BURST4_DMA_ST) {
                            asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(vaddr), "v"(d2) : "memory");
                            vaddr += 512;
                        }
                    }
                    continue;
                }
                if (KIND == X_NOLDS_PLAIN6 || KIND == X_NOLDS_IND6) {     // no LDS at all: the A operand stays where it is
                    const h8 aa1 = __builtin_bit_cast(h8, slot);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa1, b[q & 3], acc, 0, 0, 0);
#pragma unroll
                    for (int w = 0; w < 6; ++w) {
                        if (KIND == X_NOLDS_PLAIN6) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(ev[w % 3]) : "v"(eh[5]));
                        else asm volatile("v_mul_f32 %0, %1, %0" : "+v"(ind[w]) : "v"(eh[5]));
                    }
                    continue;
                }
                if (KIND == X_NOWAIT || KIND == X_HALFRATE_PLAIN6) asm volatile("" : "+a"(slot) : : "memory");
                else if (vfrag) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(slot) : : "memory");
                else asm volatile("s_waitcnt lgkmcnt(7)" : "+a"(slot) : : "memory");
                const h8 aa = __builtin_bit_cast(h8, slot);
                if (KIND == X_B64_PLAIN6) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=a"(*(double*)&slot) : "v"(lane16 + ((it & 3) * 32768u)), "n"(0) : "memory");
                else if (KIND == X_B32_PLAIN6) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=a"(slot[0]) : "v"(lane16 + ((it & 3) * 32768u)), "n"(0) : "memory");
                else if (KIND == X_HALFRATE_PLAIN6 && (q & 1)) asm volatile("" : "+a"(slot));      // a fragment read every second MFMA only
                else if (vfrag) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(slot) : "v"(lane16 + ((it & 3) * 32768u)), "n"(0) : "memory");
                else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(refill) : "v"(lane16 + ((it & 3) * 32768u)), "n"(0) : "memory");
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa, b[q & 3], acc, 0, 0, 0);
                if (KIND != SKEL_LDS && KIND != X_NODMA && KIND != X_NODMA_PLAIN6 && KIND != X_VGPRFRAG_NODMA && KIND < X_B64_PLAIN6 && (q & 3) == 3) {
                    const char* src = wsrc + (size_t)(op & 127) * 16384;
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(wsoff), "s"(src), "s"(lds_dst + (op & 7) * 16384) : "memory");
                    ++op;
                }
                if (KIND == X_NOPS) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");      // 24 idle issue cycles, no vector work
                if (KIND == X_SALU6) {
#pragma unroll
                    for (int w = 0; w < 6; ++w) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sdummy));
                }
                if (KIND == X_IND6) {
#pragma unroll
                    for (int w = 0; w < 6; ++w) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(ind[w]) : "v"(eh[5]));
                }
                if (KIND == X_PLAIN1 || KIND == X_PLAIN3) {
#pragma unroll
                    for (int w = 0; w < (KIND == X_PLAIN1 ? 1 : 3); ++w) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(ev[w % 3]) : "v"(eh[5]));
                }
                if (KIND == X_PLAIN6 || KIND == X_NODMA_PLAIN6 || (KIND >= X_B64_PLAIN6 && KIND < X_VALUFIRST_PLAIN6)) {
#pragma unroll
                    for (int w = 0; w < 6; ++w) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(ev[w % 3]) : "v"(eh[5]));
                }
                if (KIND == X_EPI_SKEW || KIND == X_EPI_SKEW_ST) {
                    // the same epilogue instructions, software-pipelined: every instruction reads registers written at
                    // least one MFMA earlier (no dependent vector instruction follows its producer within a step)
                    asm volatile("v_accvgpr_read_b32 %0, a201" : "=v"(ra[q & 3]) ::);
                    asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(rm[q & 3]) : "v"(eh[7]));
                    asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(rs[q & 3]) : "v"(rm[(q + 3) & 3]), "v"(eh[5]), "v"(eh[6]));
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ry[q & 3]) : "v"(ra[(q + 2) & 3]), "v"(rs[(q + 3) & 3]));
                    if (q & 1) {
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(rp[(q >> 1) & 1]) : "v"(ry[(q + 3) & 3]), "v"(ry[(q + 2) & 3]));
                        asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(d2[(q >> 1) & 1]) : "v"(rp[((q >> 1) + 1) & 1]), "v"(eh[4]));
                        asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(ev[2]) : "v"(ry[(q + 3) & 3]), "v"(ry[(q + 2) & 3]));
                    }
                    if (KIND == X_EPI_SKEW_ST && (q & 3) == 3) {
                        asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(vaddr), "v"(d2) : "memory");
                        vaddr += 512;
                    }
                }
                if (KIND == SKEL_EPI || KIND == SKEL_EPI_ST || (KIND >= X_NODMA && KIND <= X_NOWAIT) || KIND == X_VGPRFRAG_NODMA) {
                    // + the single-pass backward chain's epilogue of one value per MFMA (valu_cost_f16.hip, KIND 20) and,
                    // SKEL_EPI_ST, its 512-byte D store every fourth MFMA
                    const int i = q & 1;
                    if (KIND != X_NOACCREAD) asm volatile("v_accvgpr_read_b32 %0, a201" : "=v"(ev[i]) ::);
                    asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(eh[i]) : "v"(eh[7]));
                    asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(eh[i]) : "v"(eh[5]), "v"(eh[6]));
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(ev[i]) : "v"(eh[i]));
                    if (q & 1) {
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(eh[2]) : "v"(ev[0]), "v"(ev[1]));
                        asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(d2[(q >> 1) & 1]) : "v"(eh[2]), "v"(eh[4]));
                        asm volatile("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(ev[2]) : "v"(ev[0]), "v"(ev[1]));
                    }
                    if (KIND == SKEL_EPI_ST && (q & 3) == 3) {
                        asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(vaddr), "v"(d2) : "memory");
                        vaddr += 512;
                    }
                }
                if (KIND >= SKEL_FULL && q == 7 && (it & 3) == 3) {      // one chunk = 32 MFMAs: counted wait + barrier
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                continue;
            }
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
    float s = ra[0] + ry[1] + (float)rm[2] + (float)rs[3] + (float)rp[0] + ev[0] + ev[1] + ev[2] + (float)eh[2] + (float)sdummy + ind[0] + ind[1] + ind[2] + ind[3] + ind[4] + ind[5];
    for (int j = 0; j < 16; ++j) s += acc[j] + accB[j];
    for (int r = 0; r < 16; ++r) s += (float)bop[r][0];
    for (int r = 0; r < 8; ++r) s += (float)bop2[r][1];
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
    run<SKEL_LDS, 4>("skeleton: + LDS fragment read per MFMA", out, wsrc, in, cyc);
    run<SKEL_LDS_DMA, 4>("skeleton: + LDS-DMA piece per 4 MFMAs", out, wsrc, in, cyc);
    run<SKEL_FULL, 4>("skeleton: + vmcnt wait and s_barrier per 32", out, wsrc, in, cyc);
    run<SKEL_EPI, 4>("skeleton: + the backward epilogue's VALU mix", out, wsrc, in, cyc);
    run<SKEL_EPI_ST, 4>("skeleton: + its D store per 4 MFMAs", out, wsrc, in, cyc);
    run<X_EPI_SKEW, 4>("  epilogue software-pipelined (no dependent pair within a step)", out, wsrc, in, cyc);
    run<X_EPI_SKEW_ST, 4>("  epilogue software-pipelined + the D store", out, wsrc, in, cyc);
    run<X_NODMA, 4>("  epilogue, no DMA pieces", out, wsrc, in, cyc);
    run<X_VGPRFRAG, 4>("  epilogue, fragments read into VGPRs", out, wsrc, in, cyc);
    run<X_VGPRFRAG_NODMA, 4>("  epilogue, VGPR fragments, no DMA", out, wsrc, in, cyc);
    run<X_NOACCREAD, 4>("  epilogue without v_accvgpr_read", out, wsrc, in, cyc);
    run<X_NOWAIT, 4>("  epilogue, no lgkmcnt wait", out, wsrc, in, cyc);
    run<X_PLAIN6, 4>("  six v_mul_f32 instead of the epilogue", out, wsrc, in, cyc);
    run<X_NODMA_PLAIN6, 4>("  six v_mul_f32, no DMA pieces", out, wsrc, in, cyc);
    run<X_B64_PLAIN6, 4>("  six v_mul_f32, no DMA, ds_read_b64 per MFMA", out, wsrc, in, cyc);
    run<X_B32_PLAIN6, 4>("  six v_mul_f32, no DMA, ds_read_b32 per MFMA", out, wsrc, in, cyc);
    run<X_HALFRATE_PLAIN6, 4>("  six v_mul_f32, no DMA, b128 every 2nd MFMA, no wait", out, wsrc, in, cyc);
    run<X_PREVSLOT_PLAIN6, 4>("  six v_mul_f32, no DMA, refill the PREVIOUS MFMA's slot", out, wsrc, in, cyc);
    run<X_EARLY_PLAIN6, 4>("  six v_mul_f32, no DMA, order: MFMA, read (prev slot), VALU", out, wsrc, in, cyc);
    run<X_VALUFIRST_PLAIN6, 4>("  six v_mul_f32, no DMA, order: VALU, wait, read, MFMA", out, wsrc, in, cyc);
    run<X_NOPS, 4>("  no vector work, 24 cycles of s_nop per MFMA, no DMA", out, wsrc, in, cyc);
    run<X_SALU6, 4>("  six s_add_u32 per MFMA, no DMA", out, wsrc, in, cyc);
    run<X_PLAIN1, 4>("  one v_mul_f32 per MFMA, no DMA", out, wsrc, in, cyc);
    run<X_PLAIN3, 4>("  three v_mul_f32 per MFMA, no DMA", out, wsrc, in, cyc);
    run<X_IND6, 4>("  six INDEPENDENT v_mul_f32 per MFMA, LDS reads, no DMA", out, wsrc, in, cyc);
    run<X_BURST8, 4>("  six independent v_mul_f32, fragment reads in bursts of 8", out, wsrc, in, cyc);
    run<X_BURST4, 4>("  six independent v_mul_f32, fragment reads in bursts of 4", out, wsrc, in, cyc);
    run<X_BURST4_DMA, 4>("  ... bursts of 4 + a DMA piece per 4 MFMAs", out, wsrc, in, cyc);
    run<X_BURST4_DMA_ST, 4>("  ... bursts of 4 + DMA piece + 512-byte store per 4 MFMAs", out, wsrc, in, cyc);
    run<X_BURST4_EPI_DMA_ST, 4>("  backward epilogue mix, bursts of 4 + DMA piece + store per 4", out, wsrc, in, cyc);
    run<X_KERNELISH_PERQ, 4>("  kernel-like tiles + in-place operands, a read per MFMA", out, wsrc, in, cyc);
    run<X_KERNELISH_BURST, 4>("  kernel-like tiles + in-place operands, reads in bursts of 4", out, wsrc, in, cyc);
    run<X_NOLDS_PLAIN6, 4>("  six v_mul_f32 (3 registers), NO LDS reads", out, wsrc, in, cyc);
    run<X_NOLDS_IND6, 4>("  six independent v_mul_f32, NO LDS reads", out, wsrc, in, cyc);
    return 0;
}

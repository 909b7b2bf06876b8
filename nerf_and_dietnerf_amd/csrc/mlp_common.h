// mlp_common.h -- machinery shared by the fused PE+MLP kernels (mlp_fp32.hip, mlp_f16x3.hip):
// the LDS weight ring fed by LDS-DMA, its counted-vmcnt / barrier synchronisation, compile-time loops
// and the diagnostic cycle stamps.  gfx950 only.
#pragma once
#include <type_traits>

#include "nerf_device.h"
#include "nerf_kernels.h"

namespace nerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_AS __attribute__((address_space(3)))
#define GLB_AS __attribute__((address_space(1)))

__device__ __forceinline__ const char* smem_base() {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    return smem;
}

__device__ __forceinline__ f32x4 lds_read4(uint32_t byte_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    return *reinterpret_cast<const f32x4*>(smem + byte_off);
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct Pipe {
    int ck;             // chunk being consumed (monotonic; ring position = ck % kRingChunks)
    int src_next;       // next chunk index of the cyclic weight stream to DMA (0..kStreamChunks-1)
    int n_chunks;       // length of the cyclic stream in chunks
    const char* wbase;  // packed weight stream (wave-uniform)
    uint32_t voff;      // this lane's byte offset inside a chunk: wave*4 KiB + lane*16
    uint32_t wave_lds;  // wave*4 KiB
    // the chunk whose 4 pieces are being dealt out after the latest sync
    const char* cur_src;
    uint32_t cur_dst;
};

// One LDS-DMA piece: global_load_lds_dwordx4 moves 1 KiB (64 lanes x 16 B) from
// sbase + voff (per-lane byte offset) to LDS at m0_dst + lane*16.  A wave's share of a 16 KiB chunk
// is 4 pieces.  Inline asm on purpose: with the builtin form hipcc (ROCm 7.2) treats every later
// ds_read as possibly aliasing the in-flight DMA and degrades all its LDS waits to lgkmcnt(0); hidden
// from the compiler, its ds_read waits stay counted and the DMA is ordered by our own vmcnt/barrier.
// Each piece costs ~60 issue cycles, about one 64-cycle MFMA: pieces are dealt out one per MFMA gap
// (4 in a row after the barrier cost ~170 idle MFMA cycles per chunk, measured).
__device__ __forceinline__ void dma_piece(const char* sbase, uint32_t voff, uint32_t lds_dst) {
    // M0 carries the LDS destination.  It is NOT saved/restored (worth 1 % of the f16x3 kernel): nothing
    // else in these kernels uses M0, which tools/check_m0.py verifies on the generated ISA at build time.
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

// Piece j of a group of 4 that shares one M0: the 13-bit instruction offset advances the global AND the LDS
// address alike (measured: tools/microbench/lds_dma_offset.hip), so only the first piece of a group
// writes M0 and no per-piece address arithmetic is needed.
template <int OFF, bool SET_M0>
__device__ __forceinline__ void dma_piece_off(const char* sbase, uint32_t voff, uint32_t lds_dst) {
    if constexpr (SET_M0)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                     : : "v"(voff), "s"(sbase), "s"(lds_dst), "n"(OFF) : "memory");
    else
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}

// Mid-chunk synchronisation point of chunk p.ck:
//   vmcnt(4*(R-3)): this wave's share of chunk ck+1 has landed (only the 4 DMAs each of chunks
//                ck+2 .. ck+R-2 may still be pending; R = kRingChunks);
//   lgkmcnt(8): every ds_read of chunk ck-1 has returned (at most this chunk's first 8 pending);
//   barrier:  => all waves' shares of ck+1 are visible, and ring slot (ck-1)%R is free for reuse.
// Then issue chunk ck+R-1 into that free slot.
__device__ __forceinline__ void pipe_piece(Pipe& p, int j) {
#if !(defined(NERF_DIAG) && NERF_DIAG == 2)
    dma_piece(p.cur_src, p.voff + j * kQuadBytes, p.cur_dst + j * kQuadBytes);
#endif
}

__device__ __forceinline__ void pipe_sync(Pipe& p, bool all_pieces) {
#if defined(NERF_DIAG) && NERF_DIAG == 2   // timing-only diagnostic: no wait, no barrier, no DMA
    asm volatile("" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(8)" ::"n"(4 * (kRingChunks - 3)) : "memory");
#if !(defined(NERF_DIAG) && NERF_DIAG == 1)   // NERF_DIAG 1: timing-only, no barrier
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
#endif
    p.cur_src = p.wbase + (size_t)p.src_next * kChunkBytes;
    p.cur_dst = kLdsRing + ((p.ck + kRingChunks - 1) & (kRingChunks - 1)) * kChunkBytes + p.wave_lds;
    p.src_next = (p.src_next + 1 == p.n_chunks) ? 0 : p.src_next + 1;
    pipe_piece(p, 0);
    if (all_pieces) { pipe_piece(p, 1); pipe_piece(p, 2); pipe_piece(p, 3); }
}

// ---- generalised form: chunks of CQ quads (CQ KiB), ring of RING slots, CQ/4 pieces per wave ----
// (the fp16 kernel runs 32 KiB chunks in a 4-slot ring: its 32-cycle MFMAs make a 16 KiB chunk last
//  only ~770 cycles, and one s_barrier per chunk was costing it several per cent)
template <int CQ>
__device__ __forceinline__ void pipe_piece_t(Pipe& p, int j) {
#if !(defined(NERF_DIAG) && NERF_DIAG == 2)
    dma_piece(p.cur_src, p.voff + j * kQuadBytes, p.cur_dst + j * kQuadBytes);
#endif
}

// Mid-chunk sync of chunk p.ck; issues piece 0 (the caller deals out the others, or passes
// first_unplaced < CQ/4 to have pieces [first_unplaced, CQ/4) issued here because the body ends first).
template <int CQ, int RING>
__device__ __forceinline__ void pipe_sync_t(Pipe& p, int first_unplaced) {
    constexpr int NPIECE = CQ / 4;
#if defined(NERF_DIAG) && NERF_DIAG == 2
    asm volatile("" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(8)" ::"n"(NPIECE * (RING - 3)) : "memory");
#if !(defined(NERF_DIAG) && NERF_DIAG == 1)
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
#endif
    p.cur_src = p.wbase + (size_t)p.src_next * (CQ * kQuadBytes);
    p.cur_dst = kLdsRing + ((p.ck + RING - 1) & (RING - 1)) * (CQ * kQuadBytes) + p.wave_lds;
    p.src_next = (p.src_next + 1 == p.n_chunks) ? 0 : p.src_next + 1;
    pipe_piece_t<CQ>(p, 0);
    for (int j = first_unplaced; j < NPIECE; ++j) pipe_piece_t<CQ>(p, j);
}

// Offset-form pieces (compile-time piece index J): pieces 4g..4g+3 share one M0 / one source base and
// differ only in the instruction offset.  FORCE_M0 must be set when the piece is not issued right after
// piece J-1 of the same chunk (the tail case, where several pieces go out at the sync).
template <int J, bool FORCE_M0>
__device__ __forceinline__ void pipe_piece_c(Pipe& p) {
#if !(defined(NERF_DIAG) && NERF_DIAG == 2)
    constexpr int G = J >> 2, O = J & 3;
    dma_piece_off<O * kQuadBytes, (O == 0) || FORCE_M0>(p.cur_src + G * 4 * kQuadBytes, p.voff,
                                                        p.cur_dst + G * 4 * kQuadBytes);
#endif
}

// pipe_sync_t with a compile-time first_unplaced (so its pieces can use the offset form)
// EXTRA: vector-memory operations (stores of the training kernels) that the caller GUARANTEES to have issued since the
// previous sync: gfx9 counts loads and stores on one in-order vmcnt, so they may stay outstanding beside the NPIECE
// DMA pieces of the chunk after next without the wait becoming weaker for the chunk it is about.
template <int CQ, int RING, int FIRST_UNPLACED, int EXTRA = 0>
__device__ __forceinline__ void pipe_sync_c(Pipe& p) {
    constexpr int NPIECE = CQ / 4;
#if defined(NERF_DIAG) && NERF_DIAG == 2
    asm volatile("" ::: "memory");
#else
#if defined(NERF_DIAG) && NERF_DIAG == 3   // timing-only diagnostic: DMA and barrier stay, the wait for the landing is dropped
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(8)" ::"n"(NPIECE * (RING - 3) + EXTRA) : "memory");
#endif
#if !(defined(NERF_DIAG) && NERF_DIAG == 1)
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
#endif
    p.cur_src = p.wbase + (size_t)p.src_next * (CQ * kQuadBytes);
    p.cur_dst = kLdsRing + ((p.ck + RING - 1) & (RING - 1)) * (CQ * kQuadBytes) + p.wave_lds;
    p.src_next = (p.src_next + 1 == p.n_chunks) ? 0 : p.src_next + 1;
    pipe_piece_c<0, true>(p);
    static_for<FIRST_UNPLACED, NPIECE>([&](auto jc) { pipe_piece_c<decltype(jc)::value, true>(p); });
}

enum { BODY_PE = 0, BODY_HID = 1, BODY_SKIP = 2, BODY_LAST = 3 };

#ifdef NERF_STAMPS   // diagnostic build only: per-phase cycle sums of wave 0 of workgroup 0
#define STAMP(var)                                                      \
    do {                                                                \
        __builtin_amdgcn_sched_barrier(0);                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                              \
    } while (0)
#else
#define STAMP(var) do { } while (0)
#endif



}  // namespace nerf

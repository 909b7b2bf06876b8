// frag_layout.h -- the fragment-major element order of the fused trainer's activation / gradient buffers.
// Plain C++ (no HIP types) so that the host tests can compile it with g++ (tests/test_host_and_abi.py).
#pragma once
#if defined(__HIPCC__)
#define NERF_HD __host__ __device__
#else
#define NERF_HD
#endif

namespace nerf {

// Inside every block of 32 rows the elements are ordered [feature / 8][(feature / 4) % 2][row % 32][feature % 4] -- the
// order in which the 64 lanes of a wave hold a 32-sample x 8-feature accumulator slice (lane = 32 * half + sample, four
// consecutive features per lane).  One store instruction of the fused forward / backward kernels then writes 512 (fp16) or
// 1024 (fp32) contiguous bytes instead of 32 pieces of 16 / 32 bytes on 32 different rows, and the weight-gradient GEMMs
// read a thread's 4 x 4 block as 32 / 64 contiguous bytes.  Same footprint as row-major with pitch ld (ld % 8 == 0); a
// column offset c (c % 8 == 0) is the pointer offset 32 * c.
NERF_HD inline long long frag_index(long long row, int col, int ld) {
    return (row >> 5) * 32 * ld + (((col >> 3) * 64 + ((col >> 2) & 1) * 32 + (int)(row & 31)) * 4 + (col & 3));
}

}  // namespace nerf

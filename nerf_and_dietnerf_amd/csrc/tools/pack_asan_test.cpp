// Host-side packers and gather-table builders under AddressSanitizer (CPU only: nothing here touches a GPU).
// Every output buffer has exactly the documented size, so a write past it is reported.
//   hipcc -fsanitize=address -fno-gpu-sanitize -O1 -g --offload-arch=gfx950 -std=c++17 -I.. tools/pack_asan_test.cpp \
//         mlp_fp32.hip mlp_f16x3.hip mlp_bwd_f16x3.hip -o /tmp/pack_asan && /tmp/pack_asan
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "nerf_kernels.h"
using namespace nerf;
int main() {
    double s = 0;
    for (int n_angles : {2, 1, 0}) {
        const size_t nblob = n_angles == 0 ? 577028 : n_angles == 2 ? 514332 : 514332 - 8 * 129;
        std::vector<float> blob(nblob);
        for (size_t i = 0; i < blob.size(); ++i) blob[i] = (float)rand() / RAND_MAX - 0.5f;
        std::vector<float> st((n_angles == 0 ? kStreamBytesXyzF32 : kStreamBytes) / 4), cs(kConstFloats);
        pack_weights_fp32(blob.data(), n_angles, st.data(), cs.data());
        std::vector<uint16_t> sth(f16_stream_bytes(n_angles, false) / 2), sth1(f16_stream_bytes(n_angles, true) / 2);
        std::vector<float> csh(kConstFloats);
        pack_weights_f16x3(blob.data(), n_angles, sth.data(), csh.data());
        pack_weights_f16(blob.data(), n_angles, sth1.data(), csh.data());
        for (bool hi_only : {false, true}) {
            std::vector<int32_t> si(f16_stream_bytes(n_angles, hi_only) / 2), ci(kConstFloats);
            build_f16x3_gather(n_angles, hi_only, si.data(), ci.data());
            for (auto v : si) if (v < 0 || (size_t)(v >> 1) > nblob) { printf("gather index out of the blob\n"); return 1; }
            for (auto v : ci) if (v < 0 || (size_t)v > nblob) { printf("const index out of the blob\n"); return 1; }
            for (bool dx : {false, true}) {
                std::vector<int32_t> bi(kBwdStreamBytes / 2);
                build_bwd_gather(n_angles, dx, hi_only, bi.data());
                for (auto v : bi) if (v < 0 || (size_t)(v >> 1) > nblob) { printf("bwd gather index out of the blob\n"); return 1; }
                s += bi[17];
            }
        }
        for (auto v : st) s += v;
        for (auto v : sth) s += v;
        for (auto v : sth1) s += v;
    }
    printf("ok %f\n", s);
    return 0;
}

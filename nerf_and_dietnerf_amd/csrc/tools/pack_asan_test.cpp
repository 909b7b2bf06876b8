#include <vector>
#include <cstdio>
#include <cstdlib>
#include "nerf_kernels.h"
using namespace nerf;
int main() {
    std::vector<float> blob(514332);
    for (size_t i = 0; i < blob.size(); ++i) blob[i] = (float)rand() / RAND_MAX - 0.5f;
    std::vector<float> st(kStreamBytes / 4), cs(kConstFloats);
    pack_weights_fp32(blob.data(), 2, st.data(), cs.data()); pack_weights_fp32(blob.data(), 1, st.data(), cs.data());
    std::vector<uint16_t> sth(kStreamBytesF16 / 2);
    std::vector<float> csh(kConstFloats);
    pack_weights_f16x3(blob.data(), 2, sth.data(), csh.data()); pack_weights_f16x3(blob.data(), 1, sth.data(), csh.data());
    double s = 0; for (auto v : st) s += v; for (auto v : sth) s += v;
    printf("ok %f\n", s);
    return 0;
}

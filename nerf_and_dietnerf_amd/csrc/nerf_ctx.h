// nerf_ctx.h -- the context object behind the C ABI, shared by nerf_api.hip (render path) and
// train_api.hip (training path).  Internal: nothing here is part of include/nerf_mi355.h.
#pragma once
#include "../../include/nerf_mi355.h"

#include <string>
#include <utility>
#include <vector>

#include "nerf_kernels.h"

namespace nerf {

int fail(const char* fmt, ...);   // sets the thread-local message of nerf_last_error(), returns 1

#define HIP_OK(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) return nerf::fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                                 __FILE__, __LINE__);                                   \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct NetWeights {
    float* stream = nullptr;     // kStreamBytes      (fp32 MFMA operand stream)
    float* cst = nullptr;        // kConstBytes
    void* stream_h = nullptr;    // kStreamBytesF16   (fp16 hi/lo fragment stream)
    void* stream_h1 = nullptr;   // kStreamBytesF16Hi (fp16 hi-only stream of the single-pass mode)
    float* cst_h = nullptr;      // kConstBytes
    bool loaded = false;
    std::vector<float> host_blob;   // last blob handed to nerf_load_weights (Keras order): seed of the trainer
};

struct TrainState;   // train_api.hip
struct CommState;    // comm_api.hip

}  // namespace nerf

struct nerf_ctx {
    nerf_config cfg;
    int num_cus = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;         // device-to-host copies of nerf_render_image(NERF_MEM_HOST), beside the kernels
    std::vector<hipEvent_t> copy_ev;           // "batch k is done" events the copy stream waits on (ring)
    nerf::NetWeights net[2];
    // scratch arena (grow-only)
    nerf::DevBuf b_orig, b_dirs, b_zc, b_zf, b_raw, b_wc, b_u0, b_u1, b_in0, b_in1, b_in2;
    nerf::DevBuf b_out[7];
    // timing
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    long long timed_rows = 0;
    unsigned long long* nonfinite = nullptr;   // device counter fed by the MLP kernels
    nerf::TrainState* train = nullptr;         // optimizer state + training buffers (nerf_train_begin)
    nerf::CommState* comm = nullptr;           // RCCL communicator + slab buffers (nerf_comm_init)
};

namespace nerf {

int ensure(nerf_ctx* c, DevBuf& b, size_t bytes);   // grow-only device buffer
int h2d(nerf_ctx* c, DevBuf& b, const void* src, size_t bytes);
int enter(nerf_ctx* c);                             // NULL check + hipSetDevice
#define ENTER(c) do { if (int r__ = nerf::enter(c)) return r__; } while (0)

void train_free(nerf_ctx* c);                       // train_api.hip: releases c->train (called by nerf_ctx_destroy)
void comm_free(nerf_ctx* c);                        // comm_api.hip: releases c->comm (called by nerf_ctx_destroy)
int comm_allreduce_mean(nerf_ctx* c, float* buf, size_t n);   // comm_api.hip: no-op without a communicator / one rank
int comm_world(const nerf_ctx* c);                            // ranks of the ctx's communicator (1 without one)
int upload_packed_weights(nerf_ctx* c, int which, const float* blob_host);   // nerf_api.hip: pack + upload streams
int train_on_load(nerf_ctx* c, int which);          // train_api.hip: no-op without a trainer
int train_flush_weights(nerf_ctx* c, int which, bool to_host = false);
// (train_flush_weights: train_api.hip, no-op unless optimizer steps changed the weights)

}  // namespace nerf

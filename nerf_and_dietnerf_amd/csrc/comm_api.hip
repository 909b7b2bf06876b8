// comm_api.hip -- multi-GPU assembly of a frame from C: one process (or thread + ctx) per GPU, rays of independent
// pixels in contiguous slabs (the partition of nerf_and_dietnerf_amd/sharding.py), ONE ncclAllGather per requested output
// over RCCL/xGMI on the ctx stream (rgb alone for a frame; weights and z -- or the fused depth -- for the video loop's
// depth frames, src/ExecutionRun.py:339-356; all six for the special ray plots, :487).  The reference has no distributed layer: this is the C-ABI twin of
// the torch.distributed path bench.py uses.  RCCL is bound at run time (dlopen) so that single-GPU users of
// libnerf_mi355.so carry no dependency on it; in a process that already holds a librccl (PyTorch-ROCm bundles one) that
// copy is reused.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <rccl/rccl.h>   // types and enums only: every entry point is resolved with dlsym

#include <vector>

#include "nerf_ctx.h"

using namespace nerf;

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    // NERF_RCCL_LIB: path of the library that provides the six nccl* entry points used here (a site's own RCCL build;
    // the tests' two-ranks-on-one-GPU stand-in, tests/stub_rccl.c).  It changes who moves the bytes, never the result.
    const char* override_path = getenv("NERF_RCCL_LIB");
    if (override_path && override_path[0]) {
        h = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
        if (!h) return fail("NERF_RCCL_LIB=%s cannot be loaded: %s", override_path, dlerror());
    }
    for (const char* n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     // a copy this process already holds (e.g. torch's)
    for (const char* n : names)
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail("RCCL is not available (dlopen librccl.so.1: %s)", dlerror());
    Rccl r;
    r.lib = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce || !r.GetErrorString)
        return fail("librccl lacks an expected entry point");
    g_rccl = r;
    return 0;
}

#define NCCL_OK(expr)                                                                                   \
    do {                                                                                                \
        ncclResult_t r__ = (expr);                                                                      \
        if (r__ != ncclSuccess) return fail("%s failed: %s", #expr, g_rccl.GetErrorString(r__));        \
    } while (0)

}  // namespace

namespace nerf {

struct CommState {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    DevBuf slab[7], full[7];        // per requested output: this rank's padded slab, the gathered image
    std::vector<hipEvent_t> ev;     // gather -> host-copy hand-over (one per output)
};

__global__ void scale_kernel(float* __restrict__ x, size_t n, float s) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= s;
}

// mean over the ranks of the ctx's communicator, in place on the ctx stream; no-op without a communicator
int comm_world(const nerf_ctx* c) { return c->comm && c->comm->comm ? c->comm->world : 1; }

int comm_allreduce_mean(nerf_ctx* c, float* buf, size_t n) {
    if (!c->comm || !c->comm->comm || c->comm->world == 1) return 0;
    NCCL_OK(g_rccl.AllReduce(buf, buf, n, ncclFloat, ncclSum, c->comm->comm, c->stream));
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, buf, n,
                       1.0f / (float)c->comm->world);
    HIP_OK(hipGetLastError());
    return 0;
}

void comm_free(nerf_ctx* c) {
    if (!c->comm) return;
    if (c->comm->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm->comm);
    for (DevBuf& b : c->comm->slab) if (b.p) (void)hipFree(b.p);
    for (DevBuf& b : c->comm->full) if (b.p) (void)hipFree(b.p);
    for (hipEvent_t e : c->comm->ev) (void)hipEventDestroy(e);
    delete c->comm;
    c->comm = nullptr;
}

}  // namespace nerf

extern "C" {

int nerf_comm_unique_id(void* id) {
    if (!id) return fail("id is NULL");
    if (int r = load_rccl()) return r;
    static_assert(sizeof(ncclUniqueId) == NERF_COMM_ID_BYTES, "NERF_COMM_ID_BYTES must match ncclUniqueId");
    NCCL_OK(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id)));
    return 0;
}

int nerf_comm_init(nerf_ctx* c, const void* id, int32_t rank, int32_t world) {
    ENTER(c);
    if (!id) return fail("id is NULL");
    if (world < 1 || rank < 0 || rank >= world) return fail("bad rank %d of %d", rank, world);
    if (int r = load_rccl()) return r;
    comm_free(c);
    CommState* s = new CommState();
    s->rank = rank; s->world = world;
    c->comm = s;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t rc = g_rccl.CommInitRank(&s->comm, world, uid, rank);
    if (rc != ncclSuccess) {
        s->comm = nullptr;
        comm_free(c);
        return fail("ncclCommInitRank failed: %s", g_rccl.GetErrorString(rc));
    }
    return 0;
}

int nerf_comm_destroy(nerf_ctx* c) {
    ENTER(c);
    HIP_OK(hipStreamSynchronize(c->stream));
    comm_free(c);
    return 0;
}

// The seven outputs of a nerf_outputs, in declaration order: floats per ray for S samples of the last pass.
static size_t out_floats(int i, int S) {
    switch (i) { case 0: return 3; case 4: return 3 * (size_t)S; case 6: return 1; default: return (size_t)S; }
}
static float** out_slot(nerf_outputs& o, int i) {
    switch (i) {
        case 0: return &o.rgb; case 1: return &o.weights; case 2: return &o.cumprod; case 3: return &o.alpha;
        case 4: return &o.rgb_samples; case 5: return &o.z; default: return &o.depth;
    }
}

int nerf_render_image_sharded_outputs(nerf_ctx* c, const float* c2w, float fov, int32_t H, int32_t W, int64_t batch,
                                      int32_t Sc, int32_t Sf, uint64_t seed, const nerf_outputs* outs, int mem) {
    ENTER(c);
    if (!c->comm || !c->comm->comm) return fail("nerf_comm_init has not been called");
    if (!c2w || !outs) return fail("NULL argument");
    if (H <= 0 || W <= 0 || Sc <= 0 || Sf < 0) return fail("bad shape %dx%d, Sc=%d Sf=%d", H, W, Sc, Sf);
    if (mem != NERF_MEM_HOST && mem != NERF_MEM_DEVICE) return fail("bad mem %d", mem);
    CommState* s = c->comm;
    const bool fine = Sf > 0 && c->net[NERF_NET_FINE].loaded;
    const int S = fine ? Sc + Sf : Sc;
    const int64_t total = (int64_t)H * W;
    const int64_t per = (total + s->world - 1) / s->world;             // equal (padded) slabs: a plain all-gather
    int64_t begin = (int64_t)s->rank * per;
    if (begin > total) begin = total;
    const int64_t count = total - begin < per ? total - begin : per;
    const bool exact = per * s->world == total;                         // no padding: gather straight into device destinations
    bool any = false;
    nerf_outputs want = *outs;
    nerf_outputs slab{};                                                // this rank's slab of every requested output
    for (int i = 0; i < 7; ++i) {
        float* dst = *out_slot(want, i);
        if (!dst) continue;
        any = true;
        const size_t fpr = out_floats(i, S);
        if (int r = ensure(c, s->slab[i], (size_t)per * fpr * sizeof(float))) return r;
        if (!(exact && mem == NERF_MEM_DEVICE))
            if (int r = ensure(c, s->full[i], (size_t)per * s->world * fpr * sizeof(float))) return r;
        *out_slot(slab, i) = (float*)s->slab[i].p;
        if (count < per)
            HIP_OK(hipMemsetAsync((float*)s->slab[i].p + count * fpr, 0, (size_t)(per - count) * fpr * sizeof(float),
                                  c->stream));
    }
    if (!any) return fail("no output requested");
    if (count > 0)
        if (int r = nerf_render_image(c, c2w, fov, H, W, begin, count, batch, Sc, Sf, nullptr, nullptr, seed, &slab,
                                      NERF_MEM_DEVICE))
            return r;
    // ONE ncclAllGather per requested output; host destinations leave on the ctx's copy stream
    // behind an event, so the copy of one output runs under the gather of the next (page-locked destinations from
    // nerf_host_alloc: DMA at link speed, as in nerf_render_image)
    if (mem == NERF_MEM_HOST && !c->copy_stream) HIP_OK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    int k = 0;
    for (int i = 0; i < 7; ++i) {
        float* dst = *out_slot(want, i);
        if (!dst) continue;
        const size_t fpr = out_floats(i, S);
        float* gathered = (exact && mem == NERF_MEM_DEVICE) ? dst : (float*)s->full[i].p;
        NCCL_OK(g_rccl.AllGather(s->slab[i].p, gathered, (size_t)per * fpr, ncclFloat, s->comm, c->stream));
        if (mem == NERF_MEM_DEVICE) {
            if (gathered != dst)
                HIP_OK(hipMemcpyAsync(dst, gathered, (size_t)total * fpr * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        } else {
            if (s->ev.size() <= (size_t)k) {
                hipEvent_t e;
                HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                s->ev.push_back(e);
            }
            HIP_OK(hipEventRecord(s->ev[k], c->stream));
            HIP_OK(hipStreamWaitEvent(c->copy_stream, s->ev[k], 0));
            HIP_OK(hipMemcpyAsync(dst, gathered, (size_t)total * fpr * sizeof(float), hipMemcpyDeviceToHost, c->copy_stream));
            ++k;
        }
    }
    if (mem == NERF_MEM_HOST) HIP_OK(hipStreamSynchronize(c->copy_stream));
    return 0;
}

int nerf_render_image_sharded(nerf_ctx* c, const float* c2w, float fov, int32_t H, int32_t W, int64_t batch, int32_t Sc,
                              int32_t Sf, uint64_t seed, float* rgb, int mem) {
    if (!rgb) return fail("NULL argument");
    nerf_outputs o{};
    o.rgb = rgb;
    return nerf_render_image_sharded_outputs(c, c2w, fov, H, W, batch, Sc, Sf, seed, &o, mem);
}

}  // extern "C"

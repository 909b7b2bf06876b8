/* stub_rccl.c -- TEST-ONLY stand-in for the six librccl entry points csrc/comm_api.hip binds with dlsym
 * (ncclGetUniqueId / CommInitRank / CommDestroy / AllGather / AllReduce / GetErrorString), so that the library's
 * multi-rank code paths (nerf_render_image_sharded, the data-parallel nerf_train_step, the mixed policy's second
 * finiteness test) can run with world > 1 on a ONE-GPU box: RCCL refuses two ranks on one device.
 *
 * Selected through NERF_RCCL_LIB=<this .so>; built by tests/test_gpu_multirank.py into a temporary directory, never
 * shipped, never linked into libnerf_mi355.so.  Ranks are processes that share the device; payloads travel
 * device -> POSIX shared memory -> device through the HIP runtime, barriers are spin-waits on C11 atomics in the same
 * segment.  Unlike RCCL the calls block the host (they synchronise the stream); results are what RCCL's would be:
 * all-gather = concatenation in rank order, all-reduce(sum) = ((r0 + r1) + r2) ... in rank order on every rank, so all
 * ranks hold bit-identical sums.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <errno.h>
#include <fcntl.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

/* ABI-compatible restatements of the rccl.h types used by the six entry points */
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;      /* ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 */
typedef int ncclDataType_t;    /* ncclFloat = 7 */
typedef int ncclRedOp_t;       /* ncclSum = 0 */
enum { kSuccess = 0, kSystemError = 2, kInvalidArgument = 4, kFloat = 7, kSum = 0 };

#define SLOT_BYTES ((size_t)8 << 20)   /* per rank and round; larger payloads go in rounds */
#define MAX_WORLD 8
#define TIMEOUT_S 120.0

typedef struct {
    atomic_int ready;        /* creator finished initialising the header */
    atomic_int attached;     /* ranks that mapped the segment */
    atomic_int arrived;      /* barrier: ranks in the current generation */
    atomic_int generation;
    int world;
} Header;

typedef struct Comm {
    Header* h;
    char* slots;             /* world x SLOT_BYTES */
    size_t map_bytes;
    int rank, world;
    char name[128];
    float* sum;              /* host scratch of the all-reduce */
} Comm;
typedef Comm* ncclComm_t;

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int barrier(Comm* c) {
    Header* h = c->h;
    const int gen = atomic_load(&h->generation);
    if (atomic_fetch_add(&h->arrived, 1) + 1 == c->world) {
        atomic_store(&h->arrived, 0);
        atomic_fetch_add(&h->generation, 1);
        return 0;
    }
    const double t0 = now_s();
    while (atomic_load(&h->generation) == gen) {
        usleep(50);
        if (now_s() - t0 > TIMEOUT_S) return 1;      /* a peer died: fail instead of hanging the GPU box */
    }
    return 0;
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return kInvalidArgument;
    memset(id, 0, sizeof *id);
    struct timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof id->internal, "/nerfstub_%ld_%lx%lx", (long)getpid(), (unsigned long)t.tv_sec,
             (unsigned long)t.tv_nsec);
    return kSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    if (!out || world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return kInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    if (strncmp(id.internal, "/nerfstub_", 10) != 0) return kInvalidArgument;
    Comm* c = (Comm*)calloc(1, sizeof(Comm));
    if (!c) return kSystemError;
    c->rank = rank; c->world = world;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->map_bytes = 4096 + (size_t)world * SLOT_BYTES;
    int creator = 1;
    int fd = shm_open(c->name, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0 && errno == EEXIST) { creator = 0; fd = shm_open(c->name, O_RDWR, 0600); }
    if (fd < 0) { free(c); return kSystemError; }
    if (creator && ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); shm_unlink(c->name); free(c); return kSystemError; }
    if (!creator) {                                   /* wait until the creator has sized the segment */
        const double t0 = now_s();
        struct stat st;
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < c->map_bytes) {
            usleep(100);
            if (now_s() - t0 > TIMEOUT_S) { close(fd); free(c); return kSystemError; }
        }
    }
    void* p = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { free(c); return kSystemError; }
    c->h = (Header*)p;
    c->slots = (char*)p + 4096;
    if (creator) {                                    /* a fresh segment is zero-filled: only world and ready to set */
        c->h->world = world;
        atomic_store(&c->h->ready, 1);
    }
    const double t0 = now_s();
    while (!atomic_load(&c->h->ready)) {
        usleep(100);
        if (now_s() - t0 > TIMEOUT_S) { munmap(p, c->map_bytes); free(c); return kSystemError; }
    }
    if (c->h->world != world) { munmap(p, c->map_bytes); free(c); return kInvalidArgument; }
    atomic_fetch_add(&c->h->attached, 1);
    while (atomic_load(&c->h->attached) < world) {    /* like ncclCommInitRank: returns once every rank has joined */
        usleep(100);
        if (now_s() - t0 > TIMEOUT_S) { munmap(p, c->map_bytes); free(c); return kSystemError; }
    }
    if (barrier(c)) { munmap(p, c->map_bytes); free(c); return kSystemError; }
    if (rank == 0) shm_unlink(c->name);               /* everyone holds a mapping: the name can go (no /dev/shm leak) */
    c->sum = (float*)malloc(SLOT_BYTES);
    *out = c;
    return kSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return kSuccess;
    munmap((void*)c->h, c->map_bytes);
    free(c->sum);
    free(c);
    return kSuccess;
}

static int d2h(void* dst, const void* src, size_t n, hipStream_t s) {
    return hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess;
}
static int h2d(void* dst, const void* src, size_t n, hipStream_t s) {
    return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t c, hipStream_t s) {
    if (!c || dt != kFloat) return kInvalidArgument;
    const size_t bytes = count * sizeof(float);
    for (size_t off = 0; off < bytes || (bytes == 0 && off == 0); off += SLOT_BYTES) {
        const size_t n = bytes - off < SLOT_BYTES ? bytes - off : SLOT_BYTES;
        if (n && d2h(c->slots + (size_t)c->rank * SLOT_BYTES, (const char*)send + off, n, s)) return kSystemError;
        if (barrier(c)) return kSystemError;
        for (int r = 0; r < c->world && n; ++r)
            if (h2d((char*)recv + (size_t)r * bytes + off, c->slots + (size_t)r * SLOT_BYTES, n, s)) return kSystemError;
        if (barrier(c)) return kSystemError;          /* nobody overwrites a slot a peer still reads */
        if (bytes == 0) break;
    }
    return kSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t s) {
    if (!c || dt != kFloat || op != kSum) return kInvalidArgument;
    const size_t bytes = count * sizeof(float);
    for (size_t off = 0; off < bytes || (bytes == 0 && off == 0); off += SLOT_BYTES) {
        const size_t n = bytes - off < SLOT_BYTES ? bytes - off : SLOT_BYTES;
        if (n && d2h(c->slots + (size_t)c->rank * SLOT_BYTES, (const char*)send + off, n, s)) return kSystemError;
        if (barrier(c)) return kSystemError;
        const size_t nf = n / sizeof(float);
        memcpy(c->sum, c->slots, n);
        for (int r = 1; r < c->world; ++r) {
            const float* x = (const float*)(c->slots + (size_t)r * SLOT_BYTES);
            for (size_t i = 0; i < nf; ++i) c->sum[i] += x[i];
        }
        if (n && h2d((char*)recv + off, c->sum, n, s)) return kSystemError;
        if (barrier(c)) return kSystemError;
        if (bytes == 0) break;
    }
    return kSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case kSuccess: return "no error";
        case kSystemError: return "stub_rccl: system error (shm / HIP copy failed or a peer timed out)";
        case kInvalidArgument: return "stub_rccl: invalid argument";
        default: return "stub_rccl: error";
    }
}

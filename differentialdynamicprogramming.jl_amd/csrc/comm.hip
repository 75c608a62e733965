// comm.hip — the ONE collective of the path, owned by the C ABI: trajectories are sharded over one process per GPU
// and never exchange data; what a multi-GPU job shares is a small statistics vector per pass / per solve (Σ cost, Σ expected
// reduction, #diverged, max iterations ...).  ddp_allreduce_stats_f64_dev reduces such a vector over the ranks with SUM for its
// first entries and MAX for the rest in a single RCCL call on the handle's stream: an all-gather of the R tiny vectors followed
// by a local reduction kernel (latency-bound either way; xGMI bandwidth is irrelevant for <= 64 doubles).
// RCCL is loaded at first use (dlopen), so the library — and every single-GPU host — has no link-time dependency on it.
#include <dlfcn.h>
#include <string.h>
#include "ddp_internal.h"

namespace {

typedef struct { char internal[DDP_COMM_ID_BYTES]; } rccl_unique_id;          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void *rccl_comm;
enum { RCCL_FLOAT64 = 8 };                                                     // ncclDataType_t: ncclFloat64 = ncclDouble = 8

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(rccl_unique_id *) = nullptr;
    int (*CommInitRank)(rccl_comm *, int, rccl_unique_id, int) = nullptr;
    int (*CommDestroy)(rccl_comm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, rccl_comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    int version = 0;
    bool preloaded = false;      // the process already held a librccl (e.g. torch's own copy): that instance is used, no second one is loaded
};
constexpr int RCCL_MIN_VERSION = 21800;   // NCCL_VERSION(2,18,0): the ABI declared above (ncclUniqueId by value, ncclFloat64 = 8) is that of NCCL 2.x

Rccl &rccl_state()
{
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        // a host that has RCCL loaded already (PyTorch ships torch/lib/librccl.so) must not end up with two instances in one process:
        // first ask the loader for a copy it holds (RTLD_NOLOAD), by the names above and through an already-resolved symbol
        for (const char *name : names) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            if (r.lib) break;
        }
        if (!r.lib) {
            if (void *sym = dlsym(RTLD_DEFAULT, "ncclGetUniqueId")) {
                Dl_info info;
                if (dladdr(sym, &info) && info.dli_fname) r.lib = dlopen(info.dli_fname, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
            }
        }
        r.preloaded = r.lib != nullptr;
        for (const char *name : names) {
            if (r.lib) break;
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (r.lib) {
            r.GetUniqueId = (int (*)(rccl_unique_id *))dlsym(r.lib, "ncclGetUniqueId");
            r.CommInitRank = (int (*)(rccl_comm *, int, rccl_unique_id, int))dlsym(r.lib, "ncclCommInitRank");
            r.CommDestroy = (int (*)(rccl_comm))dlsym(r.lib, "ncclCommDestroy");
            r.AllGather = (int (*)(const void *, void *, size_t, int, rccl_comm, hipStream_t))dlsym(r.lib, "ncclAllGather");
            r.GetErrorString = (const char *(*)(int))dlsym(r.lib, "ncclGetErrorString");
            r.GetVersion = (int (*)(int *))dlsym(r.lib, "ncclGetVersion");
            if (r.GetVersion && r.GetVersion(&r.version) != 0) r.version = 0;
        }
    }
    return r;
}

Rccl *rccl()
{
    Rccl &r = rccl_state();
    // older than the declared ABI (or no ncclGetVersion at all): refuse rather than call through mismatched prototypes
    return (r.lib && r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.version >= RCCL_MIN_VERSION) ? &r : nullptr;
}

#define DDP_RCCL(call)                                                                                             \
    do {                                                                                                           \
        int e_ = (call);                                                                                           \
        if (e_ != 0) {                                                                                             \
            ddp_set_error("%s failed: %s", #call, R->GetErrorString ? R->GetErrorString(e_) : "RCCL error");       \
            return -3;                                                                                             \
        }                                                                                                          \
    } while (0)

// buf[0..nsum) <- Σ_r g[r][.],  buf[nsum..nsum+nmax) <- max_r g[r][.]
__global__ void reduce_gathered_kernel(int R, int nsum, int nmax, const double *g, double *buf)
{
    const int e = threadIdx.x, n = nsum + nmax;
    if (e >= n) return;
    double acc = g[e];
    for (int r = 1; r < R; ++r) {
        const double v = g[(size_t)n * r + e];
        acc = (e < nsum) ? acc + v : (v > acc || v != v ? v : acc);
    }
    buf[e] = acc;
}

}   // namespace

struct ddp_comm_s {
    rccl_comm comm;
    int nranks, rank, device;
    double *gather;                 // [nranks][DDP_COMM_MAX_STATS] on the device
};

extern "C" {

// RCCL as this library sees it: version code (NCCL_VERSION_CODE of the loaded librccl, 0 = not loadable / too old) and whether the
// instance was already resident in the process (1) or loaded here (0)
int ddp_comm_rccl_info(int *version, int *preloaded)
{
    Rccl &R = rccl_state();
    if (version) *version = R.version;
    if (preloaded) *preloaded = R.preloaded ? 1 : 0;
    if (!rccl()) {
        ddp_set_error(R.lib ? "RCCL %d is older than the ABI this library declares (>= %d)" : "RCCL (librccl.so) could not be loaded", R.version,
                      RCCL_MIN_VERSION);
        return -3;
    }
    return 0;
}

int ddp_comm_unique_id(char id[DDP_COMM_ID_BYTES])
{
    Rccl *R = rccl();
    DDP_CHECK(R, "ddp_comm_unique_id: RCCL (librccl.so) could not be loaded, or is older than 2.18");
    DDP_CHECK(id, "ddp_comm_unique_id: id is NULL");
    rccl_unique_id u;
    DDP_RCCL(R->GetUniqueId(&u));
    memcpy(id, u.internal, DDP_COMM_ID_BYTES);
    return 0;
}

int ddp_comm_create(ddp_handle h, int nranks, int rank, const char id[DDP_COMM_ID_BYTES], ddp_comm *out)
{
    DDP_DEVICE(h);
    Rccl *R = rccl();
    DDP_CHECK(R, "ddp_comm_create: RCCL (librccl.so) could not be loaded, or is older than 2.18");
    DDP_CHECK(out && id && nranks >= 1 && rank >= 0 && rank < nranks, "ddp_comm_create: bad argument (nranks=%d rank=%d)", nranks, rank);
    rccl_unique_id u;
    memcpy(u.internal, id, DDP_COMM_ID_BYTES);
    ddp_comm c = new ddp_comm_s();
    c->nranks = nranks; c->rank = rank; c->device = h->device; c->gather = nullptr; c->comm = nullptr;
    int e = R->CommInitRank(&c->comm, nranks, u, rank);
    if (e != 0) {
        ddp_set_error("ncclCommInitRank failed: %s", R->GetErrorString ? R->GetErrorString(e) : "RCCL error");
        delete c;
        return -3;
    }
    if (hipMalloc((void **)&c->gather, (size_t)nranks * DDP_COMM_MAX_STATS * sizeof(double)) != hipSuccess) {
        R->CommDestroy(c->comm);
        delete c;
        ddp_set_error("ddp_comm_create: hipMalloc failed");
        return -2;
    }
    *out = c;
    return 0;
}

int ddp_comm_destroy(ddp_comm c)
{
    if (!c) return 0;
    Rccl *R = rccl();
    hipSetDevice(c->device);
    if (c->gather) hipFree(c->gather);
    if (R && c->comm) R->CommDestroy(c->comm);
    delete c;
    return 0;
}

int ddp_allreduce_stats_f64_dev(ddp_handle h, ddp_comm c, double *buf, int nsum, int nmax)
{
    DDP_DEVICE(h);
    Rccl *R = rccl();
    DDP_CHECK(R && c && buf, "allreduce_stats: null argument or RCCL missing");
    const int n = nsum + nmax;
    DDP_CHECK(nsum >= 0 && nmax >= 0 && n >= 1 && n <= DDP_COMM_MAX_STATS, "allreduce_stats: %d + %d entries (1 .. %d allowed)", nsum, nmax,
              DDP_COMM_MAX_STATS);
    DDP_RCCL(R->AllGather(buf, c->gather, (size_t)n, RCCL_FLOAT64, c->comm, h->stream));
    hipLaunchKernelGGL(reduce_gathered_kernel, dim3(1), dim3(64), 0, h->stream, c->nranks, nsum, nmax, c->gather, buf);
    DDP_HIP(hipGetLastError());
    return 0;
}

}   // extern "C"

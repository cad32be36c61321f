// capi_comm.hip -- see capi_internal.h for the map of the C-ABI files.
#include "capi_internal.h"

extern "C" {

// ---- RCCL -------------------------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void *lib = nullptr;
    int (*get_unique_id)(void *) = nullptr;
    int (*allreduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*destroy)(void *) = nullptr;
    const char *(*errstr)(int) = nullptr;
};
struct UniqueId {
    char internal[128];
};
RcclApi g_rccl;
int load_rccl() {
    if (g_rccl.lib) return FH_OK;
    // The RCCL that sits BESIDE the HIP runtime this process runs on.  A process can hold two ROCm stacks (this library
    // loaded first, on /opt/rocm; then `import torch`, which brings its bundled librccl / libhsa-runtime64): a bare
    // dlopen("librccl.so.1") then returns the bundled RCCL, which opens the bundled -- never initialised -- HSA runtime and
    // ncclCommInitRank fails with "no ROCm-capable device is detected".
    void *lib = nullptr;
    Dl_info hip_rt{};
    if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &hip_rt) && hip_rt.dli_fname) {
        std::string dir(hip_rt.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash);
            lib = dlopen((dir + "/librccl.so.1").c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!lib) lib = dlopen((dir + "/librccl.so").c_str(), RTLD_NOW | RTLD_LOCAL);
        }
    }
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(FH_ERR_HIP, "cannot load librccl: %s", dlerror());
    g_rccl.get_unique_id = (int (*)(void *))dlsym(lib, "ncclGetUniqueId");
    g_rccl.allreduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclAllReduce");
    g_rccl.destroy = (int (*)(void *))dlsym(lib, "ncclCommDestroy");
    g_rccl.errstr = (const char *(*)(int))dlsym(lib, "ncclGetErrorString");
    if (!g_rccl.get_unique_id || !dlsym(lib, "ncclCommInitRank") || !g_rccl.allreduce || !g_rccl.destroy)
        return fail(FH_ERR_HIP, "librccl lacks a required symbol");
    g_rccl.lib = lib;
    return FH_OK;
}
}  // namespace

int fh_comm_unique_id(char id[128]) {
    int rc = load_rccl();
    if (rc) return rc;
    int s = g_rccl.get_unique_id(id);
    if (s != 0) return fail(FH_ERR_HIP, "ncclGetUniqueId: %s", g_rccl.errstr ? g_rccl.errstr(s) : "?");
    return FH_OK;
}

int fh_comm_create(const char id[128], int rank, int world, int device, fh_comm **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(FH_ERR_INVALID, "fh_comm_create: bad argument");
    int rc = load_rccl();
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    fh_comm *cm = new fh_comm();
    cm->rank = rank;
    cm->world = world;
    cm->device = device;
    if (hipEventCreate(&cm->ev0) != hipSuccess || hipEventCreate(&cm->ev1) != hipSuccess) {
        delete cm;
        return fail(FH_ERR_HIP, "fh_comm_create: hipEventCreate failed");
    }
    UniqueId uid;
    memcpy(uid.internal, id, 128);
    // ncclResult_t ncclCommInitRank(ncclComm_t*, int nranks, ncclUniqueId commId, int rank): the id travels by value
    typedef int (*init_fn)(void **, int, UniqueId, int);
    init_fn init = (init_fn)dlsym(g_rccl.lib, "ncclCommInitRank");
    int s = init(&cm->comm, world, uid, rank);
    if (s != 0) {
        delete cm;
        return fail(FH_ERR_HIP, "ncclCommInitRank: %s", g_rccl.errstr ? g_rccl.errstr(s) : "?");
    }
    *out = cm;
    return FH_OK;
}

void fh_comm_destroy(fh_comm *cm) {
    if (!cm) return;
    (void)hipSetDevice(cm->device);
    if (cm->comm && g_rccl.destroy) g_rccl.destroy(cm->comm);
    if (cm->ev0) (void)hipEventDestroy(cm->ev0);
    if (cm->ev1) (void)hipEventDestroy(cm->ev1);
    delete cm;
}

int fh_comm_allreduce_stats(fh_comm *cm, fh_ctx *c) {
    if (!cm || !c || !c->stats_sum.p) return fail(FH_ERR_INVALID, "fh_comm_allreduce_stats: bad argument");
    {
        const int rcs = settle_reset(c);
        if (rcs) return rcs;
    }
    if (c->device != cm->device) return fail(FH_ERR_INVALID, "fh_comm_allreduce_stats: context and communicator live on different devices");
    HIP_TRY(hipSetDevice(c->device));
    enum { kFloat64 = 8, kSum = 0, kMax = 2 };  // ncclDataType_t / ncclRedOp_t values (rccl.h)
    // the buffer fh_stats_finalize reads: the packed tile triangle, or the dense (N+1)^2 Gram of the rows + dgemm path
    // (N > 303: it lives in stats_sum; debris model at N <= 303: in wide_G) -- always with its two trailing scalars
    double *buf = use_wide(c) ? dense_gram(c) : c->stats_sum.p;
    const size_t len = use_wide(c) ? dense_tail(c) + 2 : c->stats_sum.n;
    HIP_TRY(hipEventRecord(cm->ev0, c->stream));
    int s = g_rccl.allreduce(buf, buf, len, kFloat64, kSum, cm->comm, c->stream);
    if (s == 0) s = g_rccl.allreduce(c->stats_minmax.p, c->stats_minmax.p, 2, kFloat64, kMax, cm->comm, c->stream);
    if (s != 0) return fail(FH_ERR_HIP, "ncclAllReduce: %s", g_rccl.errstr ? g_rccl.errstr(s) : "?");
    HIP_TRY(hipEventRecord(cm->ev1, c->stream));
    cm->timed = true;
    c->have_device_Mj = false;
    return FH_OK;
}

int fh_comm_last_allreduce_ms(fh_comm *cm, float *ms) {
    if (!cm || !ms) return fail(FH_ERR_INVALID, "fh_comm_last_allreduce_ms: NULL argument");
    if (!cm->timed) return fail(FH_ERR_INVALID, "no all-reduce recorded yet");
    HIP_TRY(hipSetDevice(cm->device));
    HIP_TRY(hipEventSynchronize(cm->ev1));
    HIP_TRY(hipEventElapsedTime(ms, cm->ev0, cm->ev1));
    return FH_OK;
}

int fh_comm_size(const fh_comm *cm) { return cm ? cm->world : 0; }


}  // extern "C"

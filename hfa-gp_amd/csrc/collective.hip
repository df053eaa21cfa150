// hfagp_allreduce_f32 (include/hfagp.h; SURVEY.md §8b "optional"): the one exchange step of the path — the all-reduce of
// the shared gradient buffer (latent basis, driver net, generator when tuned; /root/reference/code/train_rgb.py:53-57,196-202
// wraps the module in DDP for it) — for a host that is NOT PyTorch.  The communicator is the host's (ncclCommInitRank over
// its own launcher); the call is ncclAllReduce in place on the caller's stream.  No link-time dependency on RCCL: the symbol
// is looked up in the running process (a host that uses RCCL has it loaded; PyTorch-ROCm loads its own librccl.so), then in
// librccl.so on the loader path.
#include <dlfcn.h>
#include "common.h"

namespace hfagp {

typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*GetVersionFn)(int*);
typedef int (*CommCountFn)(void*, int*);

struct Rccl {
    AllReduceFn allreduce = nullptr;
    CommCountFn count = nullptr;      // ncclCommCount: the mean by hand when ncclAvg cannot be relied upon
    int version = 0;                  // ncclGetVersion: major * 10000 + minor * 100 + patch (0: the entry point is absent)
    char why[256] = "no dlerror";     // the loader's message at the moment the lookup failed
};

// Looked up ONCE per process (function-local static: thread-safe initialisation), the dlerror() text captured at the failing
// call — a second dlerror() returns NULL.
static const Rccl& rccl() {
    static const Rccl r = [] {
        Rccl x;
        void* h = nullptr;
        x.allreduce = reinterpret_cast<AllReduceFn>(dlsym(RTLD_DEFAULT, "ncclAllReduce"));
        if (!x.allreduce) {
            h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) { const char* e = dlerror(); if (e) snprintf(x.why, sizeof(x.why), "%s", e); }
            else {
                x.allreduce = reinterpret_cast<AllReduceFn>(dlsym(h, "ncclAllReduce"));
                if (!x.allreduce) { const char* e = dlerror(); if (e) snprintf(x.why, sizeof(x.why), "%s", e); }
            }
        }
        GetVersionFn gv = reinterpret_cast<GetVersionFn>(h ? dlsym(h, "ncclGetVersion") : dlsym(RTLD_DEFAULT, "ncclGetVersion"));
        if (gv && gv(&x.version) != 0) x.version = 0;
        x.count = reinterpret_cast<CommCountFn>(h ? dlsym(h, "ncclCommCount") : dlsym(RTLD_DEFAULT, "ncclCommCount"));
        return x;
    }();
    return r;
}

__global__ void __launch_bounds__(256) scale_kernel(float* __restrict__ buf, size_t n, float f) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) buf[i] *= f;
}

}  // namespace hfagp

using namespace hfagp;

extern "C" int hfagp_allreduce_f32(void* buf, size_t n, void* comm, int32_t average, void* stream) {
    HFAGP_REQUIRE(comm, HFAGP_EBADARG, "allreduce_f32: null communicator");
    if (n == 0) return HFAGP_OK;
    HFAGP_REQUIRE(buf, HFAGP_EBADARG, "allreduce_f32: null buffer");
    const Rccl& lib = rccl();
    HFAGP_REQUIRE(lib.allreduce, HFAGP_EUNSUPPORTED, "allreduce_f32: ncclAllReduce not found (load librccl.so in the host process "
                                                     "or put it on the loader path): %s", lib.why);
    // ncclSum = 0 and ncclFloat32 = 7 in every NCCL / RCCL; ncclAvg = 4 exists since 2.10.  A library that is older — or whose
    // ncclGetVersion entry point is missing, so that its version is unknown (that used to be refused outright although
    // ncclAllReduce had been found) — gets ncclSum and the division by ncclCommCount on the same stream.
    constexpr int kFloat32 = 7, kSum = 0, kAvg = 4;                     // rccl.h: ncclFloat32, ncclSum, ncclAvg
    const bool native_avg = lib.version >= 21000;
    HFAGP_REQUIRE(!average || native_avg || lib.count, HFAGP_EUNSUPPORTED,
                  "allreduce_f32: RCCL version %d has no ncclAvg and ncclCommCount was not found for the mean by hand", lib.version);
    int rc = lib.allreduce(buf, buf, n, kFloat32, (average && native_avg) ? kAvg : kSum, comm, (hipStream_t)stream);
    if (rc == 0 && average && !native_avg) {
        int world = 0;
        rc = lib.count(comm, &world);
        if (rc == 0 && world > 1) {
            scale_kernel<<<(unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), 256, 0, (hipStream_t)stream>>>(
                static_cast<float*>(buf), n, 1.0f / (float)world);
            const int lrc = check_launch("allreduce_f32 (mean by hand)");
            if (lrc != HFAGP_OK) return lrc;
        }
    }
    int dev = -1;
    (void)hipGetDevice(&dev);
    HFAGP_REQUIRE(rc == 0, HFAGP_ELAUNCH, "allreduce_f32: ncclAllReduce returned %d (device %d, %zu floats, RCCL %d)", rc, dev, n,
                  lib.version);
    return HFAGP_OK;
}

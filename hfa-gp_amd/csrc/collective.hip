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

struct Rccl {
    AllReduceFn allreduce = nullptr;
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
        return x;
    }();
    return r;
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
    // the enum values below are those of rccl.h since NCCL 2.10 (ncclAvg appeared there); an older or unidentifiable library
    // is refused rather than called with a reduction op it may number differently
    HFAGP_REQUIRE(lib.version >= 21000, HFAGP_EUNSUPPORTED, "allreduce_f32: RCCL reports version %d (need >= 2.10.0 for ncclAvg)",
                  lib.version);
    constexpr int kFloat32 = 7, kSum = 0, kAvg = 4;                     // rccl.h: ncclFloat32, ncclSum, ncclAvg
    const int rc = lib.allreduce(buf, buf, n, kFloat32, average ? kAvg : kSum, comm, (hipStream_t)stream);
    int dev = -1;
    (void)hipGetDevice(&dev);
    HFAGP_REQUIRE(rc == 0, HFAGP_ELAUNCH, "allreduce_f32: ncclAllReduce returned %d (device %d, %zu floats, RCCL %d)", rc, dev, n,
                  lib.version);
    return HFAGP_OK;
}

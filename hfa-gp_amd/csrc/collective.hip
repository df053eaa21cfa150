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

static AllReduceFn find_allreduce() {
    static AllReduceFn fn = nullptr;
    static bool looked = false;
    if (looked) return fn;
    looked = true;
    fn = reinterpret_cast<AllReduceFn>(dlsym(RTLD_DEFAULT, "ncclAllReduce"));
    if (!fn) {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (h) fn = reinterpret_cast<AllReduceFn>(dlsym(h, "ncclAllReduce"));
    }
    return fn;
}

}  // namespace hfagp

using namespace hfagp;

extern "C" int hfagp_allreduce_f32(void* buf, size_t n, void* comm, int32_t average, void* stream) {
    HFAGP_REQUIRE(comm, HFAGP_EBADARG, "allreduce_f32: null communicator");
    if (n == 0) return HFAGP_OK;
    HFAGP_REQUIRE(buf, HFAGP_EBADARG, "allreduce_f32: null buffer");
    AllReduceFn fn = find_allreduce();
    HFAGP_REQUIRE(fn, HFAGP_EUNSUPPORTED, "allreduce_f32: ncclAllReduce not found (load librccl.so in the host process or put it "
                                          "on the loader path): %s", dlerror() ? dlerror() : "no dlerror");
    constexpr int kFloat32 = 7, kSum = 0, kAvg = 4;                     // rccl.h: ncclFloat32, ncclSum, ncclAvg
    const int rc = fn(buf, buf, n, kFloat32, average ? kAvg : kSum, comm, (hipStream_t)stream);
    HFAGP_REQUIRE(rc == 0, HFAGP_ELAUNCH, "allreduce_f32: ncclAllReduce returned %d", rc);
    return HFAGP_OK;
}

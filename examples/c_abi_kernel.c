/* Plain-C host that LAUNCHES kernels of libhfagp_hip.so without Python or PyTorch: device memory from the HIP runtime's
 * C API, a stream, hfagp_bias_act_fwd (bias + leaky-ReLU * sqrt 2 + clamp) and hfagp_upfirdn2d_fwd (EG3D's upsample2d)
 * on it, results checked on the host.  This is all a non-Python caller needs: the library
 * links libamdhip64 itself; a host that ALSO loads another HIP runtime (PyTorch-ROCm ships its own) must load that one
 * first — hfa-gp_amd/_lib.py imports torch before dlopen for that reason.
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_kernel.c \
 *       -L hfa-gp_amd -lhfagp_hip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/hfa-gp_amd -Wl,-rpath,/opt/rocm/lib -o c_abi_kernel */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime_api.h>
#include "hfagp.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 10; } } while (0)

int main(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { printf("no GPU\n"); return 77; }
    hipStream_t s;
    CHECK_HIP(hipStreamCreate(&s));

    /* ---- bias_act on [N=2][C=4][inner=6] */
    enum { N = 2, C = 4, IN = 6, TOT = N * C * IN };
    float hx[TOT], hb[C], hy[TOT];
    for (int i = 0; i < TOT; ++i) hx[i] = 0.37f * (float)(i - 20);
    for (int c = 0; c < C; ++c) hb[c] = 0.5f * (float)c - 1.0f;
    float *dx, *db, *dy;
    CHECK_HIP(hipMalloc((void**)&dx, sizeof hx)); CHECK_HIP(hipMalloc((void**)&db, sizeof hb)); CHECK_HIP(hipMalloc((void**)&dy, sizeof hy));
    CHECK_HIP(hipMemcpyAsync(dx, hx, sizeof hx, hipMemcpyHostToDevice, s));
    CHECK_HIP(hipMemcpyAsync(db, hb, sizeof hb, hipMemcpyHostToDevice, s));
    const float alpha = 0.2f, gain = 1.41421356f, clamp = 5.0f;
    int rc = hfagp_bias_act_fwd(dx, db, dy, TOT, C, IN, HFAGP_ACT_LRELU, alpha, gain, clamp, s);
    if (rc != HFAGP_OK) { printf("bias_act_fwd -> %d (%s)\n", rc, hfagp_last_error()); return 2; }
    CHECK_HIP(hipMemcpyAsync(hy, dy, sizeof hy, hipMemcpyDeviceToHost, s));
    CHECK_HIP(hipStreamSynchronize(s));
    double worst = 0.0;
    for (int i = 0; i < TOT; ++i) {
        float v = hx[i] + hb[(i / IN) % C];
        v = (v < 0.f ? v * alpha : v) * gain;
        v = v > clamp ? clamp : (v < -clamp ? -clamp : v);
        if (fabs((double)v - hy[i]) > worst) worst = fabs((double)v - hy[i]);
    }
    printf("bias_act: max |device - host| = %.3g\n", worst);
    if (worst > 1e-6) return 3;

    /* ---- upsample2d of a constant image stays constant away from the border (FIR gain 4 = up^2) */
    enum { H = 6, W = 5 };
    float himg[H * W], hf[16], hup[4 * H * W];
    const float k[4] = {1.f, 3.f, 3.f, 1.f};
    for (int i = 0; i < H * W; ++i) himg[i] = 2.5f;
    for (int i = 0; i < 16; ++i) hf[i] = k[i / 4] * k[i % 4] / 64.f;
    float *dimg, *df, *dup;
    CHECK_HIP(hipMalloc((void**)&dimg, sizeof himg)); CHECK_HIP(hipMalloc((void**)&df, sizeof hf)); CHECK_HIP(hipMalloc((void**)&dup, sizeof hup));
    CHECK_HIP(hipMemcpyAsync(dimg, himg, sizeof himg, hipMemcpyHostToDevice, s));
    CHECK_HIP(hipMemcpyAsync(df, hf, sizeof hf, hipMemcpyHostToDevice, s));
    rc = hfagp_upfirdn2d_fwd(dimg, df, dup, 1, 1, H, W, 4, 4, 2, 1, 2, 1, 2, 1, 4.0f, s);
    if (rc != HFAGP_OK) { printf("upfirdn2d_fwd -> %d (%s)\n", rc, hfagp_last_error()); return 4; }
    CHECK_HIP(hipMemcpyAsync(hup, dup, sizeof hup, hipMemcpyDeviceToHost, s));
    CHECK_HIP(hipStreamSynchronize(s));
    const float centre = hup[(2 * H / 2) * (2 * W) + W];
    printf("upsample2d: centre %.4f (want 2.5)\n", centre);
    if (fabsf(centre - 2.5f) > 1e-5f) return 5;
    hipFree(dx); hipFree(db); hipFree(dy); hipFree(dimg); hipFree(df); hipFree(dup);
    hipStreamDestroy(s);
    printf("c_abi_kernel OK\n");
    return 0;
}

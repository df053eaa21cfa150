/* Host-side sanitizer driver: every host code path of libhfagp_hip.so that runs WITHOUT a GPU — planning
 * (split-K / tiling / segment choice), workspace and scratch sizing, argument validation, the error string — over a sweep of
 * shapes, modes and precisions, under AddressSanitizer + UndefinedBehaviorSanitizer (tools/sanitize/run.sh builds the
 * library's host side with -fsanitize=address,undefined and runs this).  No kernel is launched: every *_fwd call here is one
 * that validation rejects. */
#include <stdio.h>
#include <string.h>
#include "hfagp.h"

static int fails = 0;
#define EXPECT(cond, what) do { if (!(cond)) { printf("FAIL: %s (last error: %s)\n", what, hfagp_last_error()); ++fails; } } while (0)

int main(void) {
    EXPECT(hfagp_abi_version() == HFAGP_ABI_VERSION, "abi version");
    static const int dims[][5] = {  /* B, H, W, Cin, Cout */
        {1, 4, 4, 512, 512}, {2, 8, 8, 512, 512}, {32, 64, 64, 512, 256}, {32, 128, 128, 32, 256}, {32, 256, 256, 256, 128},
        {5, 37, 21, 32, 128}, {1, 1, 1, 16, 128}, {3, 257, 255, 64, 96}, {2, 16, 16, 24, 40}, {64, 512, 512, 128, 128}};
    static const int precs[] = {HFAGP_PREC_F32, HFAGP_PREC_BF16X3, HFAGP_PREC_BF16X6, HFAGP_PREC_F16, HFAGP_PREC_F16X3};
    size_t total = 0;
    for (unsigned d = 0; d < sizeof dims / sizeof dims[0]; ++d)
        for (unsigned pi = 0; pi < sizeof precs / sizeof precs[0]; ++pi)
            for (int mode = 0; mode <= 4; ++mode)
                for (int ks = 0; ks <= 3; ks += 3) {
                    HfagpModconvArgs a;
                    memset(&a, 0, sizeof a);
                    a.x = (const float*)64; a.wt = (const void*)64; a.y = (float*)64;     /* never dereferenced on the host */
                    a.B = dims[d][0]; a.H = dims[d][1]; a.W = dims[d][2]; a.Cin = dims[d][3]; a.Cout = dims[d][4];
                    a.mode = mode; a.act = HFAGP_ACT_LRELU; a.clamp = -1.0f; a.ksplit = ks; a.precision = precs[pi];
                    total += hfagp_modconv_workspace_bytes(&a);
                    total += (size_t)hfagp_modconv_rgb_parts(&a);
                    total += hfagp_upconv_fir_scratch_bytes(&a);
                    HfagpWgradArgs w;
                    memset(&w, 0, sizeof w);
                    w.B = a.B; w.H = a.H; w.W = a.W; w.Cin = a.Cin; w.Cout = a.Cout; w.mode = mode <= 2 ? mode : 0;
                    w.ksplit = ks ? ks : 1; w.precision = precs[pi] == HFAGP_PREC_F32 ? HFAGP_PREC_F32 : HFAGP_PREC_BF16X3;
                    total += hfagp_wgrad_workspace_bytes(&w);
                }
    printf("planned %zu bytes of workspace over the sweep\n", total);
    total += hfagp_pool_mse_workspace_bytes();

    /* every entry point must reject null / malformed arguments with a code and a message, not touch memory */
    EXPECT(hfagp_raymarch_fwd(NULL, NULL) < 0, "raymarch_fwd(NULL)");
    EXPECT(hfagp_raymarch_bwd(NULL, NULL) < 0, "raymarch_bwd(NULL)");
    EXPECT(hfagp_style_fwd(NULL, NULL) < 0, "style_fwd(NULL)");
    EXPECT(hfagp_style_batch_fwd(NULL, 3, NULL) < 0, "style_batch_fwd(NULL)");
    EXPECT(hfagp_style_bwd(NULL, NULL) < 0, "style_bwd(NULL)");
    EXPECT(hfagp_style_batch_bwd(NULL, 3, NULL) < 0, "style_batch_bwd(NULL)");
    EXPECT(hfagp_qr_gram_fwd(NULL, NULL, NULL, NULL, 50, NULL) < 0, "qr_gram_fwd(NULL)");
    EXPECT(hfagp_qr_refine_fwd(NULL, NULL, NULL, NULL, 50, NULL) < 0, "qr_refine_fwd(NULL)");
    EXPECT(hfagp_fc_fwd(NULL, NULL, NULL, NULL, 2, 512, 512, 1.f, 0, 0.2f, 1.f, NULL) < 0, "fc_fwd(NULL)");
    EXPECT(hfagp_weight_prep(NULL, NULL, NULL, 128, 128, 9, NULL) < 0, "weight_prep(NULL)");
    EXPECT(hfagp_weight_prep_split(NULL, NULL, 128, 128, 9, 2, NULL) < 0, "weight_prep_split(NULL)");
    EXPECT(hfagp_weight_prep_prec(NULL, NULL, 128, 128, 9, HFAGP_PREC_F16X3, NULL) < 0, "weight_prep_prec(NULL)");
    EXPECT(hfagp_modconv_fwd(NULL, NULL) < 0, "modconv_fwd(NULL)");
    EXPECT(hfagp_upfir_epilogue_fwd(NULL, NULL) < 0, "upfir_epilogue_fwd(NULL)");
    EXPECT(hfagp_upconv_fir_fwd(NULL, NULL, NULL) < 0, "upconv_fir_fwd(NULL)");
    EXPECT(hfagp_skip_upsample_add(NULL, NULL) < 0, "skip_upsample_add(NULL)");
    EXPECT(hfagp_torgb_skip_fwd(NULL, NULL) < 0, "torgb_skip_fwd(NULL)");
    EXPECT(hfagp_torgb_fwd(NULL, NULL) < 0, "torgb_fwd(NULL)");
    EXPECT(hfagp_torgb_finish_fwd(NULL, NULL) < 0, "torgb_finish_fwd(NULL)");
    EXPECT(hfagp_pointwise_bwd(NULL, NULL) < 0, "pointwise_bwd(NULL)");
    EXPECT(hfagp_upfir_bwd(NULL, NULL, 1, 8, 8, 32, NULL) < 0, "upfir_bwd(NULL)");
    EXPECT(hfagp_upsample2d_bwd(NULL, NULL, 1, 8, 8, 32, NULL) < 0, "upsample2d_bwd(NULL)");
    EXPECT(hfagp_planes_to_nhwc(NULL, NULL, 1, 8, 8, 32, NULL) < 0, "planes_to_nhwc(NULL)");
    EXPECT(hfagp_conv_wgrad(NULL, NULL) < 0, "conv_wgrad(NULL)");
    EXPECT(hfagp_affine_grad(NULL, NULL, NULL, NULL, 2, 512, 512, 14, NULL) < 0, "affine_grad(NULL)");
    EXPECT(hfagp_pool_mse_fwd(NULL, NULL, NULL, NULL, NULL, 1, 3, 256, 2, NULL) < 0, "pool_mse_fwd(NULL)");
    EXPECT(hfagp_pool_mse_bwd(NULL, NULL, NULL, NULL, 1, 3, 256, 2, NULL) < 0, "pool_mse_bwd(NULL)");
    EXPECT(hfagp_upfirdn2d_fwd(NULL, NULL, NULL, 1, 1, 8, 8, 4, 4, 2, 1, 2, 1, 2, 1, 4.f, NULL) < 0, "upfirdn2d_fwd(NULL)");
    EXPECT(hfagp_upfirdn2d_bwd(NULL, NULL, NULL, 1, 1, 8, 8, 4, 4, 2, 1, 2, 1, 2, 1, 4.f, NULL) < 0, "upfirdn2d_bwd(NULL)");
    EXPECT(hfagp_bias_act_fwd(NULL, NULL, NULL, 16, 4, 4, HFAGP_ACT_LRELU, 0.2f, 1.f, -1.f, NULL) < 0, "bias_act_fwd(NULL)");
    EXPECT(hfagp_bias_act_bwd(NULL, NULL, NULL, 16, HFAGP_ACT_LRELU, 0.2f, 1.f, -1.f, NULL) < 0, "bias_act_bwd(NULL)");
    EXPECT(hfagp_bias_act_fwd(NULL, NULL, NULL, 0, 4, 4, HFAGP_ACT_LRELU, 0.2f, 1.f, -1.f, NULL) == 0, "bias_act_fwd(n = 0) is a no-op");
    EXPECT(hfagp_blur_down_fwd(NULL, NULL, 1, 8, 8, 32, NULL) < 0, "blur_down_fwd(NULL)");
    EXPECT(hfagp_blur_down_bwd(NULL, NULL, 1, 8, 8, 32, NULL) < 0, "blur_down_bwd(NULL)");
    EXPECT(hfagp_nchw_to_nhwc(NULL, NULL, 1, 8, 8, 8, NULL) < 0, "nchw_to_nhwc(NULL)");
    EXPECT(hfagp_nhwc_to_nchw(NULL, NULL, 1, 8, 8, 8, NULL) < 0, "nhwc_to_nchw(NULL)");
    EXPECT(hfagp_allreduce_f32((void*)64, 16, NULL, 0, NULL) < 0, "allreduce_f32(NULL communicator)");
    EXPECT(strlen(hfagp_last_error()) > 0, "last error is set");

    /* malformed but non-null */
    HfagpModconvArgs a;
    memset(&a, 0, sizeof a);
    a.x = (const float*)64; a.wt = (const void*)64; a.y = (float*)64;
    a.B = 2; a.H = 8; a.W = 8; a.Cin = 6; a.Cout = 512; a.mode = HFAGP_CONV3X3; a.act = HFAGP_ACT_LRELU; a.clamp = -1.f;
    a.precision = HFAGP_PREC_F16X3;
    EXPECT(hfagp_modconv_fwd(&a, NULL) == HFAGP_EUNSUPPORTED, "modconv_fwd(Cin = 6)");
    a.Cin = 512; a.mode = 17;
    EXPECT(hfagp_modconv_fwd(&a, NULL) < 0, "modconv_fwd(mode = 17)");
    a.mode = HFAGP_CONVT3X3_UP2; a.Cout = 96;
    EXPECT(hfagp_upconv_fir_scratch_bytes(&a) == 0, "upconv_fir: Cout = 96 is not supported");
    EXPECT(hfagp_upconv_fir_fwd(&a, (void*)64, NULL) < 0, "upconv_fir_fwd(unsupported shape)");
    printf(fails ? "host sanitizer driver: %d FAILED\n" : "host sanitizer driver OK\n", fails);
    return fails ? 1 : 0;
}

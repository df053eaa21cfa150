// LDS layout constants shared by the 16-bit matrix-pipe conv kernels (modconv_bf16.hip, upconv_fir.hip).
#pragma once
#include "modconv_plan.h"
#include "split_mfma.h"

namespace hfagp {

constexpr int APITCH = 48;       // LDS bytes per patch position and part: 16 bf16 + 16 B pad (3 x 16-B slots, odd)
// LDS row pitch of the patch in positions.  32 (a multiple of 16) makes every 16-lane group of a ds_read_b128
// cover 16 consecutive columns -> 16 distinct 16-B slots, no bank conflicts (the groups are {0-3,12-15,20-27},...);
// the 3-part image would not fit twice per CU at that pitch and keeps the dense one (1 extra LDS cycle per group).
template <int NP> struct RowPitch { static constexpr int value = NP <= 2 ? 32 : PW + 2; };
constexpr int BNB = 128;         // output channels per block


// the streaming up-sampling layer for Cin = 32 (upfir_lean.hip), planned and launched from upconv_fir.hip's entry points
struct LeanParams {
    const float* x; const void* wt; const float* styles; const float* dcoef; const float* noise; const float* bias;
    const float* x_absmax; float* y_absmax; float* y;
    long long x_batch_stride;
    int B, H, W, Cout;
    int nstrip, nseg, nsteps;       // column strips of 28 output columns, row segments per strip, steps (8 output rows) per column
    int act; float noise_strength, alpha, gain, clamp;
};
bool upfir_lean_plan(const HfagpModconvArgs* a, LeanParams& lp, long long min_blocks);
int launch_upfir_lean(const HfagpModconvArgs* a, LeanParams& lp, hipStream_t s);

}  // namespace hfagp

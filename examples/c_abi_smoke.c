/* Plain-C caller of libhfagp_hip.so: what a host language other than Python binds (include/hfagp.h).
 * No GPU work: prints the ABI version, the split-K workspace the library asks for on one layer, and shows that
 * argument validation returns codes + messages instead of aborting.
 *   gcc -std=c99 -Iinclude examples/c_abi_smoke.c -L hfa-gp_amd -lhfagp_hip -Wl,-rpath,$PWD/hfa-gp_amd -o c_abi_smoke */
#include <stdio.h>
#include <string.h>
#include "hfagp.h"

int main(void) {
    printf("abi %d\n", hfagp_abi_version());
    if (hfagp_abi_version() != HFAGP_ABI_VERSION) return 1;

    int rc = hfagp_raymarch_fwd(NULL, NULL);
    printf("raymarch_fwd(NULL) -> %d (%s)\n", rc, hfagp_last_error());
    if (rc != HFAGP_EBADARG) return 2;

    HfagpModconvArgs a;
    memset(&a, 0, sizeof a);
    a.x = (const float*)16; a.wt = (const void*)16; a.y = (float*)16;     /* never dereferenced on the host */
    a.B = 2; a.H = 8; a.W = 8; a.Cin = 512; a.Cout = 512;
    a.mode = HFAGP_CONV3X3; a.act = HFAGP_ACT_LRELU; a.clamp = -1.0f;
    a.precision = HFAGP_PREC_F16X3;
    printf("3x3 512->512 @8x8, B=2: split-K workspace %zu bytes\n", hfagp_modconv_workspace_bytes(&a));

    a.Cin = 6;                                     /* not a multiple of the K chunk */
    rc = hfagp_modconv_fwd(&a, NULL);
    printf("modconv_fwd(Cin=6) -> %d (%s)\n", rc, hfagp_last_error());
    return rc == HFAGP_EUNSUPPORTED ? 0 : 3;
}

/* Plain C99 consumer of include/ocean_hip.h: the header must be valid C and the library must link from C.
 * Without a GPU every call has to fail cleanly (status < 0, message available), never crash. */
#include <stdio.h>
#include <string.h>
#include "ocean_hip.h"

int main(void) {
    OceanContext* ctx = (OceanContext*)0;
    OceanPropagateLocals pl = {0.0f, 512, 1000.0f};
    OceanCorrectionLocals cl = {512u};
    int32_t st;
    if (ocean_abi_version() != OCEAN_ABI_VERSION) return 10;
    if (sizeof(pl) != 12 || sizeof(cl) != 4) return 11;
    st = ocean_context_create(0, 500, &ctx);               /* not a power of two */
    if (st != OCEAN_E_UNSUPPORTED_N || ctx != 0) return 12;
    if (strlen(ocean_last_error((const OceanContext*)0)) == 0) return 13;
    st = ocean_context_create(0, 512, &ctx);               /* no GPU in the CPU tier: must fail, not crash */
    if (st == OCEAN_OK) { ocean_context_destroy(ctx); printf("gpu present\n"); return 0; }
    if (st >= 0 || ctx != 0) return 14;
    if (ocean_frame((OceanContext*)0, 0.0f, (void*)0) != OCEAN_E_INVALID_ARG) return 15;
    if (ocean_set_quirks((OceanContext*)0, OCEAN_QUIRKS_REFERENCE) != OCEAN_E_INVALID_ARG) return 16;
    if (ocean_quirks((const OceanContext*)0) != 0u) return 17;
    ocean_context_destroy((OceanContext*)0);
    (void)pl; (void)cl;
    printf("ok: %s\n", ocean_last_error((const OceanContext*)0));
    return 0;
}

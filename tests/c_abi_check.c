/* Plain C99 consumer of include/ocean_hip.h: the header must be valid C and the library must link from C.
 *
 *   c_abi_check                                   error paths (every call fails cleanly without a GPU, never crashes)
 *   c_abi_check spectrum.bin omega.bin crop.f32   GPU tier (tests/test_gpu_native.py): decode the reference's bincode
 *       inputs, ocean_upload_spectrum, ocean_frame(t = 1), ocean_read_displacement at N = 512 and compare the
 *       top-left 64 x 64 texels with the golden crop (raw float32 [64][64][4]); exit 0 = within 1e-4. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ocean_hip.h"

#define N 512
#define CROP 64

static float* read_bincode_f32(const char* path, size_t floats) {   /* u64-LE count + f32-LE payload */
    FILE* f = fopen(path, "rb");
    unsigned char head[8];
    float* v;
    if (!f) return NULL;
    v = (float*)malloc(floats * sizeof(float));
    if (!v || fread(head, 1, 8, f) != 8 || fread(v, sizeof(float), floats, f) != floats) { fclose(f); free(v); return NULL; }
    fclose(f);
    return v;
}

static int gpu_frame(const char* spec_path, const char* omega_path, const char* crop_path) {
    OceanContext* ctx = NULL;
    float* h0 = read_bincode_f32(spec_path, (size_t)N * N * 2);
    float* om = read_bincode_f32(omega_path, (size_t)N * N);
    float* gold = (float*)malloc((size_t)CROP * CROP * 4 * sizeof(float));
    float* img = (float*)malloc((size_t)N * N * 4 * sizeof(float));
    FILE* g = fopen(crop_path, "rb");
    double worst = 0.0;
    int c, x, y, rc = 0;
    if (!h0 || !om || !gold || !img || !g || fread(gold, sizeof(float), (size_t)CROP * CROP * 4, g) != (size_t)CROP * CROP * 4) return 20;
    fclose(g);
    if (ocean_context_create(0, N, &ctx) != OCEAN_OK) { fprintf(stderr, "%s\n", ocean_last_error(NULL)); return 21; }
    if (ocean_frame(ctx, 1.0f, NULL) != OCEAN_E_STATE) rc = 22;            /* frame before upload is a state error */
    if (!rc && ocean_upload_spectrum(ctx, h0, om) != OCEAN_OK) rc = 23;
    if (!rc && ocean_frame(ctx, 1.0f, NULL) != OCEAN_OK) rc = 24;
    if (!rc && ocean_read_displacement(ctx, img) != OCEAN_OK) rc = 25;
    if (rc) { fprintf(stderr, "%s\n", ocean_last_error(ctx)); ocean_context_destroy(ctx); return rc; }
    for (c = 0; c < 3; ++c) {
        double num = 0.0, den = 0.0;
        for (y = 0; y < CROP; ++y)
            for (x = 0; x < CROP; ++x) {
                const double a = img[((size_t)y * N + x) * 4 + c], b = gold[((size_t)y * CROP + x) * 4 + c];
                if (fabs(a - b) > num) num = fabs(a - b);
                if (fabs(b) > den) den = fabs(b);
            }
        if (num / den > worst) worst = num / den;
    }
    ocean_context_destroy(ctx);
    if (ocean_frame(ctx, 1.0f, NULL) != OCEAN_E_INVALID_ARG) return 26;    /* a destroyed handle is rejected, not dereferenced */
    ocean_context_destroy(ctx);                                            /* and a second destroy is a no-op */
    printf("native c: fused %.3e (normalised max on the 64x64 crop, tolerance 1e-4)\n", worst);
    free(h0); free(om); free(gold); free(img);
    return worst <= 1e-4 ? 0 : 27;
}

int main(int argc, char** argv) {
    OceanContext* ctx = (OceanContext*)0;
    OceanPropagateLocals pl = {0.0f, 512, 1000.0f};
    OceanCorrectionLocals cl = {512u};
    int32_t st;
    if (ocean_abi_version() != OCEAN_ABI_VERSION) return 10;
    if (sizeof(pl) != 12 || sizeof(cl) != 4) return 11;
    st = ocean_context_create(0, 500, &ctx);               /* not a power of two */
    if (st != OCEAN_E_UNSUPPORTED_N || ctx != 0) return 12;
    if (strlen(ocean_last_error((const OceanContext*)0)) == 0) return 13;
    if (ocean_frame((OceanContext*)0, 0.0f, (void*)0) != OCEAN_E_INVALID_ARG) return 15;
    if (ocean_set_quirks((OceanContext*)0, OCEAN_QUIRKS_REFERENCE) != OCEAN_E_INVALID_ARG) return 16;
    if (ocean_quirks((const OceanContext*)0) != 0u) return 17;
    {   /* round 3's entry points: NULL handles are rejected, never dereferenced */
        uint64_t sum = 0;
        if (ocean_checksum_displacement((OceanContext*)0, (void*)0, &sum) != OCEAN_E_INVALID_ARG) return 30;
        if (ocean_pack_displacement((OceanContext*)0, OCEAN_PACK_RGB32F, (void*)16, (void*)0) != OCEAN_E_INVALID_ARG) return 31;
        if (ocean_packed_bytes((const OceanContext*)0, OCEAN_PACK_HEIGHT32F) != OCEAN_E_INVALID_ARG) return 32;
        if (ocean_set_intermediate((OceanContext*)0, OCEAN_INTER_BFP16) != OCEAN_E_INVALID_ARG) return 33;
        if (ocean_intermediate((const OceanContext*)0) != OCEAN_E_INVALID_ARG) return 34;
        if (ocean_tile_exchange_bytes((const OceanContext*)0, 2) != OCEAN_E_INVALID_ARG) return 35;
        if (ocean_tile_pass1((OceanContext*)0, &pl, 0, 2, 0, 1, (void*)16, (void*)0) != OCEAN_E_INVALID_ARG) return 36;
        if (ocean_tile_pass2((OceanContext*)0, 0, 2, 1, (const void*)16, (void*)16, (void*)0) != OCEAN_E_INVALID_ARG) return 37;
    }
    ocean_context_destroy((OceanContext*)0);
    if (argc >= 4) return gpu_frame(argv[1], argv[2], argv[3]);
    st = ocean_context_create(0, 512, &ctx);               /* no GPU in the CPU tier: must fail, not crash */
    if (st == OCEAN_OK) { ocean_context_destroy(ctx); printf("gpu present\n"); return 0; }
    if (st >= 0 || ctx != 0) return 14;
    (void)pl; (void)cl;
    printf("ok: %s\n", ocean_last_error((const OceanContext*)0));
    return 0;
}

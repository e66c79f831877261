/*
 * ocean_oracle.c -- C restatement of the gfx-ocean compute path, used as
 * (1) the second CPU oracle and (2) the timed "reference path on host cores"
 * baseline (bench.py cpu_baseline, kind "port").
 *
 * TEST INFRASTRUCTURE ONLY: nothing under gfx_ocean_amd/ links or loads this.
 * PARITY UNPINNED by the reference's own tests (it has none and cannot be built
 * here: Rust + gfx-hal + Vulkan absent) -- see oracle/ocean_oracle.py header
 * for what pins it instead.
 *
 * Algorithm-faithful to the shaders: fp32, radix-2 Stockham with one
 * cos/sin per butterfly, three separate fields, separate propagate / row /
 * column / correction passes, quirks Q1 (uint wrap) and Q2 (mirror N-1-g, no
 * conjugate) included.  Threaded over lines with OpenMP.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* shader/propagate.comp:6, shader/fft_row.comp:5 */
static const float PI_F = 3.1415926f;

typedef struct { float x, y; } vec2;

/* complex_mul -- shader/propagate.comp:12-14 */
static inline vec2 cmul(vec2 a, vec2 b) {
    vec2 r;
    r.x = a.x * b.x - a.y * b.y;
    r.y = a.y * b.x + a.x * b.y;
    return r;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* a1: shader/propagate.comp:42-72.  Outputs: height (binding 3), disp_x (4), disp_z (5). */
void oracle_propagate(const vec2* h0, const float* omega, int32_t n, float time,
                      float domain_size, vec2* height, vec2* disp_x, vec2* disp_z) {
    const uint32_t res = (uint32_t)n;
#pragma omp parallel for schedule(static)
    for (int32_t gy_i = 0; gy_i < n; ++gy_i) {
        const uint32_t gy = (uint32_t)gy_i;
        for (uint32_t gx = 0; gx < res; ++gx) {
            const uint32_t index = gx + res * gy;                        /* :43 */
            const uint32_t x = 2u * gx - res - 1u;                       /* :45 (uint wrap, Q1) */
            const uint32_t y = 2u * gy - res - 1u;                       /* :46 */
            const uint32_t index_neg = (res - gy - 1u) * res + res - gx - 1u; /* :48 */
            vec2 k;
            k.x = PI_F * (float)x / domain_size;                         /* :50-53 */
            k.y = PI_F * (float)y / domain_size;
            const float disp = omega[index] * time;                      /* :55 */
            const float c = cosf(disp), s = sinf(disp);
            const vec2 disp_pos = { c, s }, disp_neg = { c, -s };        /* :56-57 */
            const vec2 a = cmul(h0[index], disp_pos);
            const vec2 b = cmul(h0[index_neg], disp_neg);                /* no conjugate (Q2) */
            const vec2 h = { a.x + b.x, a.y + b.y };                     /* :59-62 */
            vec2 kn = { 0.0f, 0.0f };
            const float len = sqrtf(k.x * k.x + k.y * k.y);
            if (len > 1.0e-10f) { kn.x = k.x / len; kn.y = k.y / len; }  /* :64-67 */
            const vec2 mx = { 0.0f, -kn.x }, mz = { 0.0f, -kn.y };
            height[index] = h;                                           /* :69 */
            disp_x[index] = cmul(mx, h);                                 /* :70 */
            disp_z[index] = cmul(mz, h);                                 /* :71 */
        }
    }
}

/* a2: butterfly(), shader/fft_row.comp:25-40, one whole line in `buf` (2*n vec2 ping-pong).
 * Returns the index (0/1) of the half that holds the result (= log2(n) % 2). */
static int stockham_line(vec2* buf, uint32_t n) {
    const uint32_t half = n >> 1;
    uint32_t stages = 0;
    while ((1u << stages) < n) ++stages;
    for (uint32_t i = 0; i < stages; ++i) {
        const uint32_t bs = 1u << i;
        vec2* src = buf + (i & 1u) * n;
        vec2* dst = buf + ((i + 1u) & 1u) * n;
        for (uint32_t j = 0; j < half; ++j) {
            const uint32_t k = j & (bs - 1u);
            const vec2 in0 = src[j], in1 = src[j + half];
            const float theta = PI_F * (float)k / (float)bs;             /* :32 */
            const vec2 w = { cosf(theta), sinf(theta) };
            const vec2 t = cmul(in1, w);
            const uint32_t dest = (j << 1) - k;                          /* :36 */
            dst[dest].x = in0.x + t.x;       dst[dest].y = in0.y + t.y;
            dst[dest + bs].x = in0.x - t.x;  dst[dest + bs].y = in0.y - t.y;
        }
    }
    return (int)(stages & 1u);
}

/* a3: shader/fft_row.comp:44-63 -- in place, line y = data[y*n .. y*n+n) */
void oracle_fft_rows(vec2* data, int32_t n) {
#pragma omp parallel
    {
        vec2* buf = (vec2*)malloc(sizeof(vec2) * 2u * (size_t)n);
#pragma omp for schedule(static)
        for (int32_t y = 0; y < n; ++y) {
            memcpy(buf, data + (size_t)y * n, sizeof(vec2) * (size_t)n);
            const int r = stockham_line(buf, (uint32_t)n);
            memcpy(data + (size_t)y * n, buf + (size_t)r * n, sizeof(vec2) * (size_t)n);
        }
        free(buf);
    }
}

/* a4: shader/fft_col.comp:44-63 -- in place, element m of line x is data[x + n*m] */
void oracle_fft_cols(vec2* data, int32_t n) {
#pragma omp parallel
    {
        vec2* buf = (vec2*)malloc(sizeof(vec2) * 2u * (size_t)n);
#pragma omp for schedule(static)
        for (int32_t x = 0; x < n; ++x) {
            for (int32_t m = 0; m < n; ++m) buf[m] = data[(size_t)x + (size_t)n * m];
            const int r = stockham_line(buf, (uint32_t)n);
            const vec2* res = buf + (size_t)r * n;
            for (int32_t m = 0; m < n; ++m) data[(size_t)x + (size_t)n * m] = res[m];
        }
        free(buf);
    }
}

/* a5: shader/correction.comp:24-35 -- out[(y*n+x)*4 + c] = (dx.x, h.x, dz.x, 0) * sign */
void oracle_correct(const vec2* height, const vec2* disp_x, const vec2* disp_z, int32_t n,
                    float* out_rgba) {
#pragma omp parallel for schedule(static)
    for (int32_t y = 0; y < n; ++y) {
        for (int32_t x = 0; x < n; ++x) {
            const size_t index = (size_t)x + (size_t)n * y;
            const float sign_mul = (((uint32_t)x + (uint32_t)y) % 2u == 0u) ? -1.0f : 1.0f;
            float* o = out_rgba + index * 4u;
            o[0] = disp_x[index].x * sign_mul;
            o[1] = height[index].x * sign_mul;
            o[2] = disp_z[index].x * sign_mul;
            o[3] = 0.0f;              /* vec4(displacement, 0.0): w is not multiplied by the sign */
        }
    }
}

/* a8: frame order of src/render.rs:1122-1310.  work: 3*n*n vec2 scratch (height, dx, dz). */
void oracle_frame(const vec2* h0, const float* omega, int32_t n, float time, float domain_size,
                  vec2* work, float* out_rgba) {
    vec2* h = work;
    vec2* dx = work + (size_t)n * n;
    vec2* dz = work + 2u * (size_t)n * n;
    oracle_propagate(h0, omega, n, time, domain_size, h, dx, dz);
    oracle_fft_rows(dx, n); oracle_fft_rows(h, n); oracle_fft_rows(dz, n);   /* render.rs:1158-1179 */
    oracle_fft_cols(dx, n); oracle_fft_cols(h, n); oracle_fft_cols(dz, n);   /* render.rs:1210-1231 */
    oracle_correct(h, dx, dz, n, out_rgba);
}

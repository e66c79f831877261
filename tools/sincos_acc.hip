// Accuracy of the gfx950 hardware sin/cos (v_sin_f32 / v_cos_f32, argument in revolutions) on [-0.5, 0.5].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float* x, float* s, float* c, float* s2, float* c2, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    s[i] = __builtin_amdgcn_sinf(x[i]);
    c[i] = __builtin_amdgcn_cosf(x[i]);
    sincospif(2.0f * x[i], &s2[i], &c2[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n), s(n), c(n), s2(n), c2(n);
    for (int i = 0; i < n; ++i) x[i] = -0.5f + (float)i / (float)(n - 1);
    float *dx, *ds, *dc, *ds2, *dc2;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&ds2, n * 4); hipMalloc(&dc2, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, ds2, dc2, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(s2.data(), ds2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c2.data(), dc2, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, es2 = 0, ec2 = 0;
    for (int i = 0; i < n; ++i) {
        const double a = 2.0 * M_PI * (double)x[i];
        es = fmax(es, fabs(s[i] - sin(a))); ec = fmax(ec, fabs(c[i] - cos(a)));
        es2 = fmax(es2, fabs(s2[i] - sin(a))); ec2 = fmax(ec2, fabs(c2[i] - cos(a)));
    }
    printf("{\"hw_sin_maxabs\":%.3e,\"hw_cos_maxabs\":%.3e,\"sincospif_sin_maxabs\":%.3e,\"sincospif_cos_maxabs\":%.3e}\n", es, ec, es2, ec2);
    return 0;
}

// Accuracy of a Cody-Waite reduction + v_sin_f32 / v_cos_f32 (revolution-based hardware ops) against ocml sincosf,
// both measured against double-precision sin / cos of the same fp32 angle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ __forceinline__ void fast_sincos(float ang, float *sn, float *cs)
{
    const float inv2pi = 0.15915494309189535f;
    const float k = rintf(ang * inv2pi);
    // 2 pi = hi + mid + lo (Cody-Waite): hi has 11 significant bits so k * hi is exact for |k| < 2^13
    const float hi = 6.28125f, mid = 1.9350051879882812e-3f, lo = 3.0199159819567e-7f;
    float f = fmaf(-k, hi, ang);
    f = fmaf(-k, mid, f);
    f = fmaf(-k, lo, f);
    const float rev = f * inv2pi;
    *sn = __builtin_amdgcn_sinf(rev);
    *cs = __builtin_amdgcn_cosf(rev);
}
__global__ void k(const float *a, float *o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c, s2, c2;
    fast_sincos(a[i], &s, &c);
    sincosf(a[i], &s2, &c2);
    o[4 * i] = s; o[4 * i + 1] = c; o[4 * i + 2] = s2; o[4 * i + 3] = c2;
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) {
        double u = (double)((unsigned)(i * 2654435761u) >> 8) / (1 << 24);
        double range = i % 3 == 0 ? 8.0 : (i % 3 == 1 ? 300.0 : 6000.0);
        h[i] = (float)((2 * u - 1) * range);
    }
    float *da, *dout;
    hipMalloc(&da, n * 4); hipMalloc(&dout, n * 16);
    hipMemcpy(da, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, da, dout, n);
    std::vector<float> o(4 * (size_t)n);
    hipMemcpy(o.data(), dout, (size_t)n * 16, hipMemcpyDeviceToHost);
    double ef[3] = {0, 0, 0}, eo[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const double s = sin((double)h[i]), c = cos((double)h[i]);
        const int b = i % 3;
        ef[b] = fmax(ef[b], fmax(fabs(o[4 * (size_t)i] - s), fabs(o[4 * (size_t)i + 1] - c)));
        eo[b] = fmax(eo[b], fmax(fabs(o[4 * (size_t)i + 2] - s), fabs(o[4 * (size_t)i + 3] - c)));
    }
    const char *names[3] = {"|x|<8", "|x|<300", "|x|<6000"};
    for (int b = 0; b < 3; ++b) printf("%-9s max abs err: fast %.3e   ocml sincosf %.3e\n", names[b], ef[b], eo[b]);
    return 0;
}

#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float head_sum(float v)
{
    v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); return v;
}
__device__ __forceinline__ float fold32(float x, float y)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float fold16(float x, float y)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
__global__ void k(float* out, float* stages)
{
    int lane = threadIdx.x;
    float p[8];
    for (int h = 0; h < 8; ++h) p[h] = (float)((h + 1) * 1000 + lane);   // expected sum over lanes: 64*(h+1)*1000 + 2016
    const float t0 = fold32(p[0], p[4]), t1 = fold32(p[1], p[5]);
    const float t2 = fold32(p[2], p[6]), t3 = fold32(p[3], p[7]);
    const float w0 = fold16(t0, t2), w1 = fold16(t1, t3);
    const bool hi = (lane & 8) != 0;
    const float keep = hi ? w1 : w0, give = hi ? w0 : w1;
    float part = keep + dpp_mov<0x128>(give);
    stages[lane] = t0; stages[64 + lane] = w0; stages[128 + lane] = part;
    part = head_sum(part);
    out[lane] = part;
}
int main()
{
    float *d, *s; hipMalloc(&d, 256); hipMalloc(&s, 3 * 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, s);
    float h[64], st[192]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost); hipMemcpy(st, s, 768, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d got %9.0f expect %9.0f | t0 %8.0f w0 %8.0f part %8.0f\n", l, h[l], 64.0f * ((l >> 3) + 1) * 1000 + 2016, st[l], st[64 + l], st[128 + l]);
    return 0;
}

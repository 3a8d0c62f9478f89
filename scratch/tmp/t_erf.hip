#include <hip/hip_runtime.h>
__global__ void k1(const float* x, float* y) { float v = x[threadIdx.x]; y[threadIdx.x] = 0.5f * v * (1.0f + erff(v * 0.70710678f)); }
__device__ __forceinline__ float gelu_as(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float erfc_z = p * e;                       // erfc(|x|/sqrt2)
    // 1 + erf(x/sqrt2) = x >= 0 ? 2 - erfc : erfc
    const float c = x >= 0.f ? 2.0f - erfc_z : erfc_z;
    return 0.5f * x * c;
}
__global__ void k2(const float* x, float* y) { y[threadIdx.x] = gelu_as(x[threadIdx.x]); }

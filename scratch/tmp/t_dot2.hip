#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(2))) __bf16 v2bf;
__global__ void k(const unsigned* x, float* y) {
    unsigned w = x[threadIdx.x];
    v2bf a = __builtin_bit_cast(v2bf, w);
    y[threadIdx.x] = __builtin_amdgcn_fdot2_f32_bf16(a, a, 0.0f, false);
}

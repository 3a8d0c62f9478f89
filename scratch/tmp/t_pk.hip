#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__global__ void k(const float* x, unsigned* y) {
    float a = x[threadIdx.x], b = x[threadIdx.x + 64], c = x[threadIdx.x+128], d = x[threadIdx.x+192];
    y[threadIdx.x] = pack2(a, b);
    y[threadIdx.x + 64] = pack2(c, d);
}

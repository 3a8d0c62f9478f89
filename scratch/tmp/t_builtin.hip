#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) short v4s;
typedef __attribute__((ext_vector_type(2))) __bf16 v2bf;
__global__ void k(const float* x, unsigned* y, short* z) {
    __shared__ __attribute__((aligned(16))) short s[1024];
    s[threadIdx.x] = (short)x[threadIdx.x];
    __syncthreads();
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(&s[threadIdx.x * 4]));
    z[threadIdx.x] = r[0] + r[1] + r[2] + r[3];
    float a = x[threadIdx.x], b = x[threadIdx.x + 64];
    v2bf p = {(__bf16)a, (__bf16)b};
    unsigned u = __builtin_bit_cast(unsigned, p);
    auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    y[threadIdx.x] = sw[0] ^ sw[1];
}

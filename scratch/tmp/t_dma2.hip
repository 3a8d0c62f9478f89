#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// C: p8-like: MFMA consumer
__global__ __launch_bounds__(512) void kC(const char* __restrict__ src, float* __restrict__ out, int n) {
    __shared__ __attribute__((aligned(1024))) char dsm[65536];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int r = 0; r < 8; ++r)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + r * 1024 + lane * 16), (lds_ptr_t)&dsm[wave * 8192 + r * 1024], 16, 0, 0);
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < n; ++s) {
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        const mfma_bf16x8 v = *reinterpret_cast<const mfma_bf16x8*>(&dsm[wave * 8192 + (s & 7) * 1024 + lane * 16]);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (long)(s + 8) * 1024 + lane * 16), (lds_ptr_t)&dsm[wave * 8192 + (s & 7) * 1024], 16, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v, v, acc, 0, 0, 0);
    }
    out[threadIdx.x] = acc[0];
}
// D: uint4 read but no float math
__global__ __launch_bounds__(512) void kD(const char* __restrict__ src, uint4* __restrict__ out, int n) {
    __shared__ __attribute__((aligned(1024))) char dsm[65536];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int r = 0; r < 8; ++r)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + r * 1024 + lane * 16), (lds_ptr_t)&dsm[wave * 8192 + r * 1024], 16, 0, 0);
    uint4 acc = {0, 0, 0, 0};
    for (int s = 0; s < n; ++s) {
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        const uint4 v = *reinterpret_cast<const uint4*>(&dsm[wave * 8192 + (s & 7) * 1024 + lane * 16]);
        acc.x ^= v.x; acc.y ^= v.y; acc.z += v.z; acc.w += v.w;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (long)(s + 8) * 1024 + lane * 16), (lds_ptr_t)&dsm[wave * 8192 + (s & 7) * 1024], 16, 0, 0);
    }
    out[threadIdx.x] = acc;
}

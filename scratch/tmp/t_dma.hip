#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
// A: like the LM head: wave-private ring, read own 16 bytes
__global__ __launch_bounds__(512) void kA(const char* __restrict__ src, float* __restrict__ out, int n) {
    __shared__ __attribute__((aligned(1024))) char dsm[65536];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* ring = dsm + wave * 8192;
    for (int r = 0; r < 8; ++r)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + r * 1024 + lane * 16), (lds_ptr_t)(ring + r * 1024), 16, 0, 0);
    float acc = 0;
    for (int s = 0; s < n; ++s) {
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        const float4 v = *reinterpret_cast<const float4*>(ring + (s & 7) * 1024 + lane * 16);
        acc += v.x + v.y + v.z + v.w;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (long)(s + 8) * 1024 + lane * 16), (lds_ptr_t)(ring + (s & 7) * 1024), 16, 0, 0);
    }
    out[threadIdx.x] = acc;
}
// B: same but indices into the array directly
__global__ __launch_bounds__(512) void kB(const char* __restrict__ src, float* __restrict__ out, int n) {
    __shared__ __attribute__((aligned(1024))) char dsm[65536];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int r = 0; r < 8; ++r)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + r * 1024 + lane * 16), (lds_ptr_t)&dsm[wave * 8192 + r * 1024], 16, 0, 0);
    float acc = 0;
    for (int s = 0; s < n; ++s) {
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        const float4 v = *reinterpret_cast<const float4*>(&dsm[wave * 8192 + (s & 7) * 1024 + lane * 16]);
        acc += v.x + v.y + v.z + v.w;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + (long)(s + 8) * 1024 + lane * 16), (lds_ptr_t)&dsm[wave * 8192 + (s & 7) * 1024], 16, 0, 0);
    }
    out[threadIdx.x] = acc;
}

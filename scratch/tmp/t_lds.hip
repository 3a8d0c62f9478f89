#include <hip/hip_runtime.h>
__global__ __launch_bounds__(512) void k(float* y) {
    __shared__ __attribute__((aligned(1024))) char smem[131072];
    smem[threadIdx.x * 200] = (char)threadIdx.x;
    __syncthreads();
    y[threadIdx.x] = smem[(threadIdx.x * 37) % 131072];
}

// launchchain.hip -- what a DEPENDENT kernel launch costs inside a hipGraph on this part (the decode step is 142 of them).
// Chains of N launches captured into one graph, replayed R times, time per launch = wall / (R * N):
//   empty      : kernels that do nothing (grid G x 512 threads)
//   touch      : every workgroup reads 64 KiB the previous launch wrote (one dependent round trip) and writes its 256 B slice
//   touch2     : ... then reads a second buffer at an address taken from the first (two dependent round trips), like a decode
//                GEMV (activations, then the bytes they select)
// build: hipcc --offload-arch=gfx950 -O3 -o launchchain launchchain.hip ; run: ./launchchain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void k_empty(int* p) { if (p == nullptr && threadIdx.x == 1000) p[0] = 1; }

__global__ __launch_bounds__(512) void k_touch(const float4* __restrict__ in, float4* __restrict__ out, const float4* __restrict__ big, int two) {
    float4 a = make_float4(0, 0, 0, 0);
    for (int i = threadIdx.x; i < 4096; i += 512) { const float4 v = in[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }   // 64 KiB
    if (two) {
        const int idx = ((int)a.x & 1023) * 512 + threadIdx.x;     // address depends on the first read
        const float4 w = big[(size_t)blockIdx.x * 1024 * 512 % (1u << 22) + idx];
        a.x += w.x; a.y += w.y;
    }
    if (threadIdx.x < 16) out[blockIdx.x * 16 + threadIdx.x] = a;   // 256 B per workgroup
}

int main() {
    const int N = 142, R = 50;
    float4 *a, *b, *big;
    CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMalloc(&big, (size_t)(1u << 22) * 16 + (1 << 24)));
    CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20)); CK(hipMemset(big, 0, (size_t)(1u << 22) * 16 + (1 << 24)));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int mode = 0; mode < 3; ++mode)
        for (int grid : {8, 64, 256, 512}) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < N; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(512), 0, s, (int*)a);
                else hipLaunchKernelGGL(k_touch, dim3(grid), dim3(512), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, big, mode == 2);
            }
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("%-7s grid %3d x 512: %.2f us per dependent launch\n", mode == 0 ? "empty" : mode == 1 ? "touch" : "touch2", grid, us / (R * N));
            fflush(stdout);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    return 0;
}

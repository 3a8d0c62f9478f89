// lanechain.hip -- the AGGREGATE price of dependent launches when several chains run side by side on one MI355X (the lanes of
// qasr_dp_submit: one decode chain per pass in flight).  L streams, each replaying its own graph of N dependent launches on its own
// buffers, all started together; aggregate time per launch = wall / (R * N * L).  Kernels:
//   empty : nothing (grid G x 512 threads)
//   touch : every workgroup reads 64 KiB the previous launch of ITS chain wrote and writes its 256 B slice (one dependent round trip)
//   gemv  : touch + 32 KiB per workgroup of weights nobody read for > 1 GB (HBM): the decode GEMV's shape, 8 MB per launch at G = 256
// build: hipcc --offload-arch=gfx950 -O3 -o lanechain lanechain.hip ; run: ./lanechain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void k_empty(int* p) { if (p == nullptr && threadIdx.x == 1000) p[0] = 1; }

__global__ __launch_bounds__(512) void k_touch(const float4* __restrict__ in, float4* __restrict__ out, const float4* __restrict__ W) {
    float4 a = make_float4(0, 0, 0, 0);
    for (int i = threadIdx.x; i < 4096; i += 512) { const float4 v = in[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }   // 64 KiB
    if (W) {
        const float4* wp = W + (size_t)blockIdx.x * 2048 + threadIdx.x;      // 32 KiB per workgroup
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float4 w = wp[j * 512]; a.x += w.x; a.y += w.y; a.z += w.z; a.w += w.w; }
    }
    if (threadIdx.x < 16) out[blockIdx.x * 16 + threadIdx.x] = a;           // 256 B per workgroup
}

int main() {
    const int N = 142, R = 30, LMAX = 4;
    const size_t slice = (size_t)256 * 32768;                     // 8 MB of weights per gemv-shaped launch
    hipStream_t st[LMAX];
    float4 *a[LMAX], *b[LMAX];
    char* W[LMAX];
    for (int l = 0; l < LMAX; ++l) {
        CK(hipStreamCreateWithFlags(&st[l], hipStreamNonBlocking));
        CK(hipMalloc(&a[l], 1 << 20)); CK(hipMalloc(&b[l], 1 << 20)); CK(hipMemset(a[l], 0, 1 << 20)); CK(hipMemset(b[l], 0, 1 << 20));
        CK(hipMalloc(&W[l], slice * N)); CK(hipMemset(W[l], 0, slice * N));
    }
    for (int mode = 0; mode < 3; ++mode)
        for (int grid : {8, 64, 256, 512}) {
            if (mode == 2 && grid != 256) continue;
            double base = 0;
            for (int L = 1; L <= LMAX; ++L) {
                hipGraph_t g[LMAX]; hipGraphExec_t ge[LMAX];
                for (int l = 0; l < L; ++l) {
                    CK(hipStreamBeginCapture(st[l], hipStreamCaptureModeThreadLocal));
                    for (int i = 0; i < N; ++i) {
                        if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(512), 0, st[l], (int*)a[l]);
                        else hipLaunchKernelGGL(k_touch, dim3(grid), dim3(512), 0, st[l], (i & 1) ? b[l] : a[l], (i & 1) ? a[l] : b[l],
                                                mode == 2 ? (const float4*)(W[l] + slice * i) : (const float4*)nullptr);
                    }
                    CK(hipStreamEndCapture(st[l], &g[l]));
                    CK(hipGraphInstantiate(&ge[l], g[l], nullptr, nullptr, 0));
                }
                for (int w = 0; w < 2; ++w) for (int l = 0; l < L; ++l) CK(hipGraphLaunch(ge[l], st[l]));
                for (int l = 0; l < L; ++l) CK(hipStreamSynchronize(st[l]));
                const auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < R; ++r) for (int l = 0; l < L; ++l) CK(hipGraphLaunch(ge[l], st[l]));
                for (int l = 0; l < L; ++l) CK(hipStreamSynchronize(st[l]));
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                const double per = us / ((double)R * N * L);
                if (L == 1) base = per;
                printf("%-5s grid %3d x 512, %d chain%s side by side: %.2f us per launch in aggregate (%.2f x one chain's rate)\n",
                       mode == 0 ? "empty" : mode == 1 ? "touch" : "gemv", grid, L, L > 1 ? "s" : " ", per, base / per);
                fflush(stdout);
                for (int l = 0; l < L; ++l) { CK(hipGraphExecDestroy(ge[l])); CK(hipGraphDestroy(g[l])); }
            }
        }
    return 0;
}

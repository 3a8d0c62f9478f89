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


// gemv-like launch: 64 KiB of the predecessor's output (dependent), then this launch's own 32 KiB-per-workgroup weight slice from a
// region no launch has read for > 1 GB (HBM, not the caches).  pf: a ninth wave touches one dword per 128 B line of the NEXT launch's
// slice of the same workgroup index (same XCD -> same L2); wfirst: issue the own weight loads together with the X loads instead of
// after X was consumed.
template <int pf, int wfirst>
__global__ __launch_bounds__(576) void k_gemv(const float4* in, float4* out, const float4* W, const int* Wnext) {   // no __restrict__: the asm memory barriers must order the loads
    const int tid = threadIdx.x;
    if (tid >= 512) {      // ninth wave: the only one whose loads may stay outstanding for an HBM round trip (vmcnt is in-order per wave)
        if (pf) {
            const int* pp = Wnext + (size_t)blockIdx.x * 8192 + (tid - 512) * 32;
            const int p = pp[0] ^ pp[64 * 32] ^ pp[128 * 32] ^ pp[192 * 32];
            if (p == 0x7fffffff) out[0].x = 1.f;
        }
        return;
    }
    float4 x[8], w[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = in[tid + i * 512];
    const float4* wp = W + (size_t)blockIdx.x * 2048 + tid;
    if (wfirst) {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = wp[j * 512];
    }
    asm volatile("" ::: "memory");
    float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) { a.x += x[i].x; a.y += x[i].y; a.z += x[i].z; a.w += x[i].w; }
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w) :: "memory");     // X is consumed before anything below is issued
    if (!wfirst) {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = wp[j * 512];
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) { a.x += w[j].x; a.y += w[j].y; a.z += w[j].z; a.w += w[j].w; }
    if (tid < 16) out[blockIdx.x * 16 + tid] = a;
}


// ---- does warming a K/V-sized stream into the Infinity Cache from the idle time of GEMV-shaped launches pay? ----------------------------
// One "layer" = 4 GEMV-shaped launches (k_gemv2: as k_gemv, plus a ninth wave that touches a quarter of the NEXT stream's 61.5 MB, one dword
// per 128-B line, after sleeping `delay` x 64 cycles so that the X reads go first) + 1 streaming launch that reads its 61.5 MB once
// (256 workgroups x 240 KiB).  28 layers per graph, all buffers distinct (1.7 GB of streams, 0.9 GB of weights).
template <int pf>
__global__ __launch_bounds__(576) void k_gemv2(const float4* in, float4* out, const float4* W, const int* S, int part, int delay) {
    const int tid = threadIdx.x;
    if (tid >= 512) {
        if (pf) {
            if (delay > 0) for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(1);
            // quarter `part` of the stream: 61.5 MB / 4 = 15.4 MB = 120 K lines; workgroup b touches lines b*64 + lane, stride gridDim.x*64
            const int lane = tid - 512;
            const long lines = (long)256 * 245760 / 128 / 4, base = (long)part * lines;
            int acc = 0;
            for (long l = (long)blockIdx.x * 64 + lane; l < lines; l += (long)gridDim.x * 64) acc ^= S[(base + l) * 32];
            if (acc == 0x7fffffff) out[0].x = 1.f;
        }
        return;
    }
    float4 x[8], w[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = in[tid + i * 512];
    asm volatile("" ::: "memory");
    float4 a = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) { a.x += x[i].x; a.y += x[i].y; a.z += x[i].z; a.w += x[i].w; }
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w) :: "memory");
    const float4* wp = W + (size_t)blockIdx.x * 2048 + tid;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = wp[j * 512];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) { a.x += w[j].x; a.y += w[j].y; a.z += w[j].z; a.w += w[j].w; }
    if (tid < 16) out[blockIdx.x * 16 + tid] = a;
}

__global__ __launch_bounds__(512) void k_stream(const float4* in, float4* out, const float4* S) {
    const int tid = threadIdx.x;
    float4 a = in[tid];                                                    // dependent on the predecessor
    const float4* p = S + (size_t)blockIdx.x * (245760 / 16) + tid;        // 240 KiB per workgroup = 15360 float4 = 30 per thread
#pragma unroll 10
    for (int i = 0; i < 30; ++i) { const float4 v = p[i * 512]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (tid < 16) out[blockIdx.x * 16 + tid] = a;
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

    {
        const size_t slice = (size_t)256 * 32768;                 // 8 MB per launch
        char* Wb; CK(hipMalloc(&Wb, slice * (N + 1))); CK(hipMemset(Wb, 0, slice * (N + 1)));
        for (int wfirst = 0; wfirst < 2; ++wfirst)
            for (int pf = 0; pf < 2; ++pf) {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                for (int i = 0; i < N; ++i)
                {
                    auto kf = wfirst ? (pf == 0 ? k_gemv<0, 1> : k_gemv<1, 1>) : (pf == 0 ? k_gemv<0, 0> : k_gemv<1, 0>);
                    hipLaunchKernelGGL(kf, dim3(256), dim3(576), 0, s, (i & 1) ? b : a, (i & 1) ? a : b,
                                       (const float4*)(Wb + slice * i), (const int*)(Wb + slice * ((i + 1) % N)));
                }
                CK(hipStreamEndCapture(s, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                const auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
                CK(hipStreamSynchronize(s));
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                printf("gemv    grid 256 x 512, 8 MB of cold weights per launch, weights %s X, next-slice touch %s: %.2f us per launch\n",
                       wfirst ? "with " : "after", pf == 0 ? "none" : "by a ninth wave", us / (R * N));
                fflush(stdout);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
    }

    {
        const int L = 28;
        const size_t slice = (size_t)256 * 32768, stream = (size_t)256 * 245760;
        char *Wb, *Sb;
        CK(hipMalloc(&Wb, slice * 4 * L)); CK(hipMemset(Wb, 0, slice * 4 * L));
        CK(hipMalloc(&Sb, stream * L)); CK(hipMemset(Sb, 0, stream * L));
        for (int mode = 0; mode < 4; ++mode) {            // 0 no prefetch | 1 prefetch, no delay | 2 delay 16 x 64 cycles | 3 delay 32 x 64
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            int flip = 0;
            for (int l = 0; l < L; ++l) {
                for (int j = 0; j < 4; ++j) {
                    auto kf = mode == 0 ? k_gemv2<0> : k_gemv2<1>;
                    hipLaunchKernelGGL(kf, dim3(256), dim3(576), 0, s, flip ? b : a, flip ? a : b, (const float4*)(Wb + slice * (4 * l + j)),
                                       (const int*)(Sb + stream * l), j, mode == 2 ? 16 : mode == 3 ? 32 : 0);
                    flip ^= 1;
                }
                hipLaunchKernelGGL(k_stream, dim3(256), dim3(512), 0, s, flip ? b : a, flip ? a : b, (const float4*)(Sb + stream * l));
                flip ^= 1;
            }
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            const char* names[4] = {"no warm-up", "ninth wave warms the next stream, at once", "... after 16 x 64 cycles", "... after 32 x 64 cycles"};
            printf("layer   4 gemv-shaped launches + one 61.5 MB stream, %s: %.2f us per layer\n", names[mode], us / (R * L));
            fflush(stdout);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
        // the streaming launch alone, back to back on distinct buffers
        {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_stream, dim3(256), dim3(512), 0, s, (l & 1) ? b : a, (l & 1) ? a : b, (const float4*)(Sb + stream * l));
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("stream  61.5 MB per launch from HBM, alone: %.2f us per launch\n", us / (R * L));
        }
    }
    return 0;
}

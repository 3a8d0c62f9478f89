// microbenchmark: cost of a software grid barrier (256 workgroups, one per CU) with cross-XCD data visibility
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);   // agent scope by default for global atomics
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { ok = false; atomicExch(err, 1); break; }
        }
    }
    __syncthreads();
    return ok;
}

// variant: arrive on a per-XCD counter (workgroup id % 8 = XCD), the last arriver of an XCD arrives on the root, the
// last of the root publishes the generation on a separate flag line that everyone polls
__device__ __forceinline__ bool grid_barrier_tree(unsigned* c, unsigned gen, int nwg, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const int x = blockIdx.x & 7;
        const unsigned per = (nwg + 7 - x) / 8;                 // workgroups in this XCD group
        unsigned* leaf = c + 64 + x * 32;                       // 128-byte separated lines
        unsigned* root = c + 32;
        unsigned* flag = c + 64 + 8 * 32;
        const unsigned old = __hip_atomic_fetch_add(leaf, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == gen * per) {
            const unsigned r = __hip_atomic_fetch_add(root, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (r + 1 == gen * 8) __hip_atomic_store(flag, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { ok = false; atomicExch(err, 1); break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    return ok;
}

template <int MODE>
__global__ void bar_kernel(unsigned* counter, int* buf, int iters, int* err, int* mismatches) {
    const int nwg = gridDim.x, wg = blockIdx.x;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {
            // every thread writes one element of this workgroup's 2 KiB slice
            buf[wg * blockDim.x + threadIdx.x] = it * 1000 + wg;
        }
        if (*(volatile int*)err) return;
        if (MODE == 2) { if (!grid_barrier_tree(counter, (unsigned)(2 * it + 1), nwg, err)) return; }
        else if (!grid_barrier(counter, (unsigned)(it + 1) * nwg, err)) return;
        if (MODE >= 1) {
            const int src = (wg + 37 + it) % nwg;
            const int v = buf[src * blockDim.x + threadIdx.x];
            if (v != it * 1000 + src) ++bad;
        }
        if (MODE >= 1) {   // second barrier so the next iteration's writes don't race with these reads
            if (MODE == 2) { if (!grid_barrier_tree(counter, (unsigned)(2 * it + 2), nwg, err)) return; }
            else if (!grid_barrier(counter + 32, (unsigned)(it + 1) * nwg, err)) return;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

int main(int argc, char** argv) {
    int nwg = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 512, iters = 2000;
    unsigned* counter; int *buf, *err, *mm;
    CK(hipMalloc(&counter, 4096)); CK(hipMalloc(&buf, nwg * threads * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&mm, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(counter, 0, 4096)); CK(hipMemset(err, 0, 4)); CK(hipMemset(mm, 0, 4));
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(bar_kernel<0>, dim3(nwg), dim3(threads), 0, 0, counter, buf, iters, err, mm);
            else if (mode == 2) hipLaunchKernelGGL(bar_kernel<2>, dim3(nwg), dim3(threads), 0, 0, counter, buf, iters, err, mm);
            else hipLaunchKernelGGL(bar_kernel<1>, dim3(nwg), dim3(threads), 0, 0, counter, buf, iters, err, mm);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int herr, hmm; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hmm, mm, 4, hipMemcpyDeviceToHost));
            printf("mode %d (%s) nwg %d x %d thr: %.3f us per iteration (%d barriers/iter), timeout=%d mismatches=%d\n", mode,
                   mode == 2 ? "tree: write+barrier+remote read+barrier" : mode ? "write+barrier+remote read+barrier" : "barrier only", nwg, threads, ms * 1e3 / iters, mode ? 2 : 1, herr, hmm);
        }
    }
    return 0;
}

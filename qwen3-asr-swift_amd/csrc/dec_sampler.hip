// dec_sampler.hip -- device-side pickNextToken (Qwen3ASR.swift:449-520): repetition penalty, no-repeat n-gram, Gumbel-max temperature.
// The host twin is csrc/sampler.cpp (qasr_pick_next_token); tests/test_gpu_api.py holds the two against each other.
#include "dec_sampler.h"

namespace qasr {

// ------------------------------------------------------------------------------------------------
// device-side pickNextToken of the non-default decoding options (see dec_sampler.h): one workgroup per batch row
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64_at(unsigned long long s0, unsigned long long call) {
    // csrc/sampler.cpp's sequential generator is a counter: the state before call k is s0 + k * gamma
    unsigned long long z = s0 + (call + 1ull) * 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(1024) void sampler_pick_kernel(float* __restrict__ logits, int V, GreedyState st, float penalty, int ngram,
                                                            float temperature, unsigned long long seed, float* __restrict__ part_val,
                                                            int* __restrict__ part_idx) {
    __shared__ float s_v[1024], s_l[1024];
    __shared__ int s_i[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    float* row = logits + (long)b * V;
    const int n_gen = st.lens[b];
    const int* gen = st.tokens + (long)b * (st.max_new + 1);
    if (st.finished[b]) {                              // the row's picks are over (greedy_finalize ignores it)
        if (tid == 0) { part_val[b] = 0.0f; part_idx[b] = 0; }
        return;
    }
    // :469-480 -- every DISTINCT generated id once: positive logits divide, negative multiply
    if (penalty > 1.0f && n_gen > 0) {
        for (int i = tid; i < n_gen; i += 1024) {
            const int t = gen[i];
            if (t < 0 || t >= V) continue;
            bool first = true;
            for (int j = 0; j < i; ++j)
                if (gen[j] == t) { first = false; break; }
            if (first) {
                const float v = row[t];
                row[t] = v > 0.0f ? v / penalty : v * penalty;
            }
        }
    }
    __syncthreads();
    // :484-500 -- forbid the token that completed an earlier occurrence of the last (n-1)-gram
    if (ngram > 0 && n_gen >= ngram) {
        const int* last = gen + n_gen - (ngram - 1);
        for (int i = tid; i + ngram <= n_gen; i += 1024) {
            bool same = true;
            for (int j = 0; j < ngram - 1; ++j)
                if (gen[i + j] != last[j]) { same = false; break; }
            if (!same) continue;
            const int f = gen[i + ngram - 1];
            if (f >= 0 && f < V) row[f] = -INFINITY;
        }
    }
    __syncthreads();
    // :504-520 -- argmax(logits / T + Gumbel(0,1)), u in [1e-6, 1]; the first maximum wins
    const unsigned long long s0 = seed * 0x9e3779b97f4a7c15ull + (unsigned long long)b + 1ull;
    const unsigned long long call0 = (unsigned long long)n_gen * (unsigned long long)V;
    float bv = -INFINITY, bl = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 1024) {
        const float l = row[i];
        float v = l;
        if (temperature > 0.0f) {
            const double r = (double)(splitmix64_at(s0, call0 + (unsigned long long)i) >> 11) * (1.0 / 9007199254740992.0);
            const float u = (float)(1e-6 + r * (1.0 - 1e-6));
            v = l / temperature - logf(-logf(u));
        }
        if (v > bv) { bv = v; bi = i; bl = l; }       // ascending i per thread: strict '>' keeps the lowest index
    }
    s_v[tid] = bv;
    s_i[tid] = bi;
    s_l[tid] = bl;
    __syncthreads();
    for (int ofs = 512; ofs > 0; ofs >>= 1) {
        if (tid < ofs) {
            const float ov = s_v[tid + ofs];
            const int oi = s_i[tid + ofs];
            if (ov > s_v[tid] || (ov == s_v[tid] && oi < s_i[tid])) { s_v[tid] = ov; s_i[tid] = oi; s_l[tid] = s_l[tid + ofs]; }
        }
        __syncthreads();
    }
    // the partial carries the winner's LOGIT: u = 1 (the reference draws from the closed range) makes the Gumbel term +inf, which is
    // a legitimate pick, while greedy_finalize's non-finite check is about broken logits
    if (tid == 0) { part_val[b] = s_l[0]; part_idx[b] = s_i[0]; }
}

void sampler_pick_launch(float* logits, int V, GreedyState st, int B, float repetition_penalty, int ngram, float temperature,
                         unsigned long long seed, float* part_val, int* part_idx, hipStream_t s) {
    if (B <= 0) return;
    hipLaunchKernelGGL(sampler_pick_kernel, dim3(B), dim3(1024), 0, s, logits, V, st, repetition_penalty, ngram, temperature, seed,
                       part_val, part_idx);
}

}  // namespace qasr

// dec_chain_dev.h -- device-side pieces shared by the persistent decode launches (dec_chain.hip, dec_qa.hip): write-through stores / sc1 loads,
// the sharded arrival counters of a hand-off seam with bounded waits, and one GEMV unit on register-resident weight fragments.
// Protocol and its source: see the header of dec_chain.hip.
#pragma once
#include "dec_chain.h"
#include "dec_epilogue.h"

namespace qasr {
namespace chain_dev {

constexpr int CT = 512, CWAVES = 8;
constexpr int CH_H = 1024, CH_NQ = 2048, CH_I = 3072, CH_NQKV = 4096;
constexpr int CH_GRID = 256;
constexpr unsigned long long CH_SPIN_TICKS = 20000000ull;          // 200 ms of the 100 MHz wall clock

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// dynamic LDS carve (bytes, all multiples of 16)
constexpr int L_NORM = 0;                                          // ln2 | next ln1: 2 x 2 KiB
constexpr int L_FLAG = 4096;
constexpr int L_X = 4224;                                          // activation image [rows][2 K + 16]
constexpr int L_XMAX = 16 * (2 * CH_I + 16);                       // down: 98,560 B (gate|up / q|k|v with 32 rows: 66,048)
constexpr int L_RED = L_X + L_XMAX;                                // cross-wave partial sums [7][NT * NBU][256] f32
constexpr int L_TOTAL = L_RED + 7 * 4 * 1024;                      // 131,456 B: one workgroup per CU

template <bool NTW>
__device__ __forceinline__ uint4 ld_weight(const bf16_t* p) {
    if constexpr (NTW) {
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *reinterpret_cast<const uint4*>(p);
    }
}

__device__ __forceinline__ uint4 ld16_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);       // aux 16 = sc1
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 ld8_sc1(const bf16_t* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((unsigned)v, (unsigned)(v >> 32));
}
__device__ __forceinline__ void st8_sc1(bf16_t* p, uint2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// wave 0, after its write-through stores: drain, then one lane arrives
__device__ __forceinline__ void seam_signal(unsigned* ctr, int seam, int unit) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add(ctr + (seam * CHAIN_SHARDS + (unit & (CHAIN_SHARDS - 1))) * CHAIN_SHARD_WORDS, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}

// every thread of the workgroup calls this; wave 0 polls `nshards` counters (lines of their own) starting at c0.  false: the wait gave
// up (uniform over the workgroup)
__device__ __forceinline__ bool seam_wait_n(const unsigned* c0, int nshards, unsigned target, int* err, int* s_flag) {
    if (threadIdx.x < 64) {
        // lanes 0 .. nshards - 1 poll the counters; lane 63 polls the error word in the same instruction: once ANY wait of the step has given
        // up, every later wait fails at its first poll (a step that lost one arrival would otherwise spend the budget in each of its launches)
        const bool on_ctr = (int)threadIdx.x < nshards, on_err = threadIdx.x == 63;
        const unsigned* c = on_err ? reinterpret_cast<const unsigned*>(err) : c0 + (threadIdx.x & (CHAIN_SHARDS - 1)) * CHAIN_SHARD_WORDS;
        const unsigned long long t0 = wall_clock64();
        bool ok;
        for (;;) {
            const unsigned v = (on_ctr || on_err) ? __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const bool dead = __builtin_amdgcn_ballot_w64(on_err && (v & CHAIN_ERR_TIMEOUT)) != 0;
            ok = !dead && __builtin_amdgcn_ballot_w64(on_ctr && v < target) == 0;
            if (ok || dead || wall_clock64() - t0 > CH_SPIN_TICKS) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (threadIdx.x == 0) {
            *s_flag = ok ? 1 : 0;
            if (!ok) atomicOr(err, CHAIN_ERR_TIMEOUT);
        }
    }
    __syncthreads();
    return *s_flag != 0;
}
__device__ __forceinline__ bool seam_wait(const unsigned* ctr, int seam, unsigned target, int* err, int* s_flag) {
    return seam_wait_n(ctr + seam * CHAIN_SHARDS * CHAIN_SHARD_WORDS, CHAIN_SHARDS, target, err, s_flag);
}
// Replicated form of a seam's counter (MI355X_MICROARCH.md, "Valid forms", second table row): every producer adds to ALL 8 replicas with one
// wave instruction (8 active lanes, 8 lines), every consumer polls ONE replica -- 32 pollers per line instead of 256 pollers on each of 8 lines.
__device__ __forceinline__ void seam_signal_r(unsigned* ctr, int seam) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x < CHAIN_SHARDS)
        __hip_atomic_fetch_add(ctr + (seam * CHAIN_SHARDS + threadIdx.x) * CHAIN_SHARD_WORDS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool seam_wait_r(const unsigned* ctr, int seam, unsigned target, int* err, int* s_flag) {
    return seam_wait_n(ctr + (seam * CHAIN_SHARDS + (blockIdx.x & (CHAIN_SHARDS - 1))) * CHAIN_SHARD_WORDS, 1, target, err, s_flag);
}
__device__ __forceinline__ unsigned ld4_sc1(const bf16_t* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One unit with its weight fragments already in registers: acc[t][b] (valid on wave 0 afterwards) = [rmsnorm](X rows) . W^T for NT weight
// tiles and NBU batch tiles.  xr[p][i] = chunk (scol + 32 i) of row (16 p + srow), zero for rows past the batch.
// ST (diagnostic instantiations only): thread 0 stamps the 100 MHz clock into st[0] once the image is staged and st[1] once the sums are in.
struct ChainNoHook { __device__ __forceinline__ void operator()() const {} };
// after_stage(): called by every thread once its activation chunks are in LDS (their registers are free), before the barrier
template <int NT, int NBU, int KSW, bool NORM, bool ST = false, class Hook = ChainNoHook>
__device__ __forceinline__ void chain_mma(const uint4 (&w)[NT][KSW], const uint4 (&xr)[NBU][KSW], const char* s_normw, float eps, char* s_x,
                                          float* s_red, f32x4 (&acc)[NT][NBU], unsigned long long* st = nullptr, Hook after_stage = Hook()) {
    constexpr int K = KSW * CWAVES * 32, XSTRIDE = 2 * K + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fc = lane >> 4;
    const int srow = tid >> 5, scol = tid & 31;
#pragma unroll
    for (int p = 0; p < NBU; ++p) {
        char* xrow = s_x + (size_t)(p * 16 + srow) * XSTRIDE + scol * 16;
        if constexpr (NORM) {
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < KSW; ++i) {
                const bf16_t* e = reinterpret_cast<const bf16_t*>(&xr[p][i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = bf16_to_f32(e[j]); ss = fmaf(f, f, ss); }
            }
            ss = lane_sum<32>(ss);
            const float inv = rsqrtf(ss / (float)K + eps);
#pragma unroll
            for (int i = 0; i < KSW; ++i) {
                const uint4 nw = *reinterpret_cast<const uint4*>(s_normw + (scol + i * 32) * 16);
                const bf16_t* e = reinterpret_cast<const bf16_t*>(&xr[p][i]);
                const bf16_t* we = reinterpret_cast<const bf16_t*>(&nw);
                uint4 o;
                bf16_t* oe = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
                for (int j = 0; j < 8; ++j) oe[j] = f32_to_bf16(bf16_to_f32(we[j]) * bf16_round(bf16_to_f32(e[j]) * inv));
                *reinterpret_cast<uint4*>(xrow + i * 32 * 16) = o;
            }
        } else {
#pragma unroll
            for (int i = 0; i < KSW; ++i) *reinterpret_cast<uint4*>(xrow + i * 32 * 16) = xr[p][i];
        }
    }
    after_stage();
    __syncthreads();
    if constexpr (ST) { if (tid == 0) st[0] = wall_clock64(); }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < NBU; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KSW; ++i) {
        const int kb = ((wave + CWAVES * i) * 32 + fc * 8) * 2;
#pragma unroll
        for (int b = 0; b < NBU; ++b) {
            const uint4 xf = *reinterpret_cast<const uint4*>(s_x + (size_t)(b * 16 + fr) * XSTRIDE + kb);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, w[t][i]), __builtin_bit_cast(mfma_bf16x8, xf),
                                                                    acc[t][b], 0, 0, 0);
        }
    }
    // cross-wave sums in the fixed order wave 0 + 1 + ... + 7
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int b = 0; b < NBU; ++b)
                *reinterpret_cast<f32x4*>(&s_red[((size_t)(wave - 1) * NT * NBU + t * NBU + b) * 256 + lane * 4]) = acc[t][b];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int b = 0; b < NBU; ++b)
#pragma unroll
                for (int wv = 0; wv < CWAVES - 1; ++wv)
                    acc[t][b] += *reinterpret_cast<const f32x4*>(&s_red[((size_t)wv * NT * NBU + t * NBU + b) * 256 + lane * 4]);
    }
    if constexpr (ST) { if (tid == 0) st[1] = wall_clock64(); }
}

}  // namespace chain_dev
}  // namespace qasr

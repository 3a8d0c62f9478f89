// common.h -- shared device/host helpers for libqasr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <set>
#include <string>
#include <stdexcept>
#include <utility>

namespace qasr {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define QASR_HIP(expr)                                                                        \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            throw ::qasr::HipError(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + \
                                   __FILE__ + ":" + std::to_string(__LINE__) + ")");          \
    } while (0)

struct NotLoaded : std::runtime_error {      // -> QASR_ERR_NOT_LOADED at the C ABI
    using std::runtime_error::runtime_error;
};

typedef unsigned short bf16_t;  // storage type; arithmetic is always float32

__host__ __device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    union { uint32_t u; float f; } x;
    x.u = (uint32_t)v << 16;
    return x.f;
}

// round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs NaN
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

inline bf16_t f32_to_bf16_host(float f) {
    union { uint32_t u; float f; } x;
    x.f = f;
    if ((x.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((x.u >> 16) | 0x40);  // NaN
    uint32_t r = x.u + 0x7fffu + ((x.u >> 16) & 1u);
    return (bf16_t)(r >> 16);
}

inline float bf16_to_f32_host(bf16_t b) {
    union { uint32_t u; float f; } x;
    x.u = (uint32_t)b << 16;
    return x.f;
}

__device__ __forceinline__ float bf16_round(float f) { return bf16_to_f32(f32_to_bf16(f)); }
// two floats -> one word of two bf16 (lo in bits 0-15): ONE v_cvt_pk_bf16_f32.  Packing two separately converted values
// costs a conversion each plus a shift and an or (found in the attention kernels' P packing: 64 instead of 16 instructions per
// 64-key tile; the GEMM epilogues packed the same way).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_native;
typedef __attribute__((ext_vector_type(2))) float f32x2_native;
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    const f32x2_native v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_native));
}

// one word of two bf16 -> its two floats
__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
// RMSNorm of a bf16 pair at the reference's rounding points: bf16(weight * bf16(x * inv_rms))  (MLXFast.rmsNorm on bf16)
__device__ __forceinline__ unsigned rmsnorm_pair_bf16(unsigned xw, unsigned ww, float inv) {
    const unsigned t = pack_bf16x2(bf16_lo(xw) * inv, bf16_hi(xw) * inv);
    return pack_bf16x2(bf16_lo(ww) * bf16_lo(t), bf16_hi(ww) * bf16_hi(t));
}

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator

// 16-byte LDS read through a NATIVE vector type.  hipcc puts s_waitcnt vmcnt(0) in front of an LDS read whose pointee is a
// HIP_vector_type (float4 / uint4 / uint2: no type-based alias info) whenever global_load_lds copies or global stores are outstanding --
// under a direct-to-LDS prefetch that ends the prefetch at the read instead of at the barrier meant for it.
__device__ __forceinline__ float4 lds_read_f4(const float* p) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    return make_float4(v[0], v[1], v[2], v[3]);
}

// "These registers are read here": an empty asm that uses every 32-bit word of a loaded value.  hipcc places the s_waitcnt for a load at
// its first use; when that use sits in a conditional block (a guarded store), the wait state is merged conservatively at every join and
// each later guarded block gets its own s_waitcnt vmcnt(0) -- which then also waits for the stores issued in between.  A use in
// unconditional code ahead of the guarded blocks leaves ONE wait.
template <class T>
__device__ __forceinline__ void reg_use(const T& t) {
    static_assert(sizeof(T) % 4 == 0, "reg_use: whole 32-bit words");
    if constexpr (sizeof(T) >= 4) {
        unsigned w[sizeof(T) / 4];
        __builtin_memcpy(w, &t, sizeof(T));
#pragma unroll
        for (unsigned i = 0; i < sizeof(T) / 4; ++i) asm volatile("" ::"v"(w[i]));
    }
}

// Cross-lane sums on the DPP path (v_add_f32 with a lane-permuting source operand, a few cycles) instead of __shfl_xor,
// which hipcc lowers to ds_bpermute_b32: an LDS round trip (~100 cycles) per step, each behind its own s_waitcnt.
//   quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E: sums of 4;  row_half_mirror 0x141: lane i <- 7 - i, completes 8;
//   row_mirror 0x140: lane i <- 15 - i, completes 16.  Every lane of the aligned group ends with the group's sum.
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f32(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_sum8(float p) {
    p += dpp_mov_f32<0xB1>(p);
    p += dpp_mov_f32<0x4E>(p);
    p += dpp_mov_f32<0x141>(p);
    return p;
}
// sum over aligned groups of N adjacent lanes (N = 8, 16, 32 or 64), result on every lane of the group
template <int N>
__device__ __forceinline__ float lane_sum(float p) {
    static_assert(N == 8 || N == 16 || N == 32 || N == 64, "group size");
    p = lane_sum8(p);
    if constexpr (N >= 16) p += dpp_mov_f32<0x140>(p);
    if constexpr (N >= 32) {                   // v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of the second
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(p), false, false);
        p = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    if constexpr (N >= 64) {                   // v_permlane32_swap: upper half of the first operand <-> lower half of the second
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);
        p = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return p;
}

// max / sum over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} (one per 16-lane row), on every lane: two permlane swaps instead of the two
// ds_bpermute round trips that __shfl_xor(., 16) and __shfl_xor(., 32) cost (the decode attention's per-chunk score maximum).  Exact for max;
// the sum adds the same pairs as the shuffle form (x + x^16, then + the other half), so it is bit-identical to it.
__device__ __forceinline__ float rows4_max(float p) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(p), false, false);
    p = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);
    return fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
}
__device__ __forceinline__ float rows4_sum(float p) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(p), false, false);
    p = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(p), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

// maximum over aligned groups of 16 adjacent lanes, on every lane of the group (same DPP steps)
__device__ __forceinline__ float lane_max16(float p) {
    p = fmaxf(p, dpp_mov_f32<0xB1>(p));
    p = fmaxf(p, dpp_mov_f32<0x4E>(p));
    p = fmaxf(p, dpp_mov_f32<0x141>(p));
    p = fmaxf(p, dpp_mov_f32<0x140>(p));
    return p;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// erf-form GELU, as MLXNN.gelu / torch F.gelu: x * Phi(x) with Phi(x) = erfc(-x / sqrt 2) / 2.  erfc(z), z >= 0, by
// Abramowitz & Stegun 7.1.26 (t = 1 / (1 + p z), five-term polynomial times exp(-z^2); |error| <= 1.5e-7, i.e. f32 rounding
// class) -- 16 vector instructions where the library erff costs 34 with both of its branches live in a wave; the GELU
// epilogues of the FFN GEMMs are vector-ALU-bound, not MFMA-bound.  Absolute error of the result <= 0.5 |x| 1.5e-7.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float erfc_z = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    return 0.5f * x * (x >= 0.f ? 2.0f - erfc_z : erfc_z);
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize once per (device, kernel): a kernel's attributes live with the device's code object, and one
// process may drive an engine per GPU (qasr_dp_*) from several threads
inline void ensure_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    QASR_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return;
    QASR_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({dev, kernel});
}

}  // namespace qasr

// common.h -- shared device/host helpers for libqasr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <stdexcept>

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

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator

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

// exact (erf) GELU, as MLXNN.gelu / torch F.gelu
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace qasr

// dec_quant_dev.h -- device helpers shared by the kernels that multiply by MLX affine-quantised weights (dec_quant.hip, dec_qa.hip):
// scale / bias access and the exact q -> bf16 MFMA fragment conversions.
#pragma once
#include "common.h"
#include "gemm.h"      // mfma_bf16x8

namespace qasr {

template <bool F32>
__device__ __forceinline__ float sb_at(const void* p, long i) {
    if constexpr (F32) return reinterpret_cast<const float*>(p)[i];
    else return bf16_to_f32(reinterpret_cast<const bf16_t*>(p)[i]);
}

// Scale / bias image of a matrix (quant_pack_sb_kernel): per 16-row tile [row 16][group G][2 = scale, bias], so that the GPB consecutive groups
// of one k-block of one weight row are ONE aligned load of 2 GPB elements (4 - 16 bytes) instead of 2 GPB scalar ones -- a wave-level load
// instruction costs the CU's address path the same ~16 clocks whether it fetches 2 or 16 bytes per lane, and the decode GEMVs issued up to 16 of
// the 2-byte kind per lane behind their weights.  first = element index of the first group's scale: ((tile * 16 + row) * G + g) * 2.
template <bool F32, int GPB>
__device__ __forceinline__ void sb_load(const void* img, long first, float (&sc)[GPB], float (&bi)[GPB]) {
    static_assert(GPB == 1 || GPB == 2, "groups per k-block");
    if constexpr (F32) {
        if constexpr (GPB == 2) {
            const uint4 v = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(img) + first);
            sc[0] = __uint_as_float(v.x); bi[0] = __uint_as_float(v.y); sc[1] = __uint_as_float(v.z); bi[1] = __uint_as_float(v.w);
        } else {
            const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const float*>(img) + first);
            sc[0] = __uint_as_float(v.x); bi[0] = __uint_as_float(v.y);
        }
    } else {
        if constexpr (GPB == 2) {
            const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(img) + first);
            sc[0] = __uint_as_float(v.x << 16); bi[0] = __uint_as_float(v.x & 0xffff0000u);
            sc[1] = __uint_as_float(v.y << 16); bi[1] = __uint_as_float(v.y & 0xffff0000u);
        } else {
            const unsigned v = *reinterpret_cast<const unsigned*>(reinterpret_cast<const bf16_t*>(img) + first);
            sc[0] = __uint_as_float(v << 16); bi[0] = __uint_as_float(v & 0xffff0000u);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// q (integers < 2^bits) -> bf16 MFMA fragment, exactly: v_cvt_f32_ubyteN then a truncating pack (an integer < 256 has at
// most 8 significant bits, so its f32 image already is a bf16 value).
// 4 bit: one word = the lane's 8 elements of a k-step (element j in bits [4j, 4j + 4)).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) {
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);      // {hi[31:16], lo[31:16]}
}

template <int B>
__device__ __forceinline__ float ubyte_f32(unsigned w) { return (float)((w >> (8 * B)) & 0xFFu); }      // v_cvt_f32_ubyteB

// 4 bit.  The q image stores a k-step's 8 elements of a lane in the nibble order (e0 e2 e4 e6 | e1 e3 e5 e7): element
// pair (2k, 2k+1) sits in nibbles k and k + 4 (quant_pack_q_kernel).  A nibble placed in the top four mantissa bits of a
// bf16 with exponent 2^4 is the bf16 value 16 + q, exactly: bits 0x4180 | q << 3.  So one shift and one and-or give a
// packed pair, 8 vector instructions per fragment instead of 20 for the convert route, and the kernels multiply by
// (16 + q): sum (16 + q) x = sum q x + 16 sum x, folded into the bias term (bias' = bias - 16 scale, Q4_OFFSET).
constexpr float Q4_OFFSET = 16.0f;
__device__ __forceinline__ mfma_bf16x8 frag_q4(unsigned w) {
    uint4 o;
    o.x = ((w << 3) & 0x00780078u) | 0x41804180u;
    o.y = ((w >> 1) & 0x00780078u) | 0x41804180u;
    o.z = ((w >> 5) & 0x00780078u) | 0x41804180u;
    o.w = ((w >> 9) & 0x00780078u) | 0x41804180u;
    return __builtin_bit_cast(mfma_bf16x8, o);
}
// bias of a group as the kernels use it: the 4-bit fragments carry 16 + q
template <int BITS>
__device__ __forceinline__ float eff_bias(float scale, float bias) {
    if constexpr (BITS == 4) return fmaf(-Q4_OFFSET, scale, bias);
    else return bias;
}

__device__ __forceinline__ mfma_bf16x8 frag_q8(unsigned w0, unsigned w1) {      // w0 = elements 0..3, w1 = 4..7
    uint4 o;
    o.x = pack_hi16(ubyte_f32<0>(w0), ubyte_f32<1>(w0));
    o.y = pack_hi16(ubyte_f32<2>(w0), ubyte_f32<3>(w0));
    o.z = pack_hi16(ubyte_f32<0>(w1), ubyte_f32<1>(w1));
    o.w = pack_hi16(ubyte_f32<2>(w1), ubyte_f32<3>(w1));
    return __builtin_bit_cast(mfma_bf16x8, o);
}

// word i of a 16-byte register block; i is a compile-time constant after unrolling (never an address computation:
// indexing a register array through a pointer cast sends it to scratch)
__device__ __forceinline__ unsigned u4_word(const uint4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// k-step `ks` (0 .. BLK/32 - 1) of a lane's 16-byte block
template <int BITS>
__device__ __forceinline__ mfma_bf16x8 frag_of(const uint4& blk, int ks) {
    if constexpr (BITS == 4) return frag_q4(u4_word(blk, ks));
    else return frag_q8(u4_word(blk, 2 * ks), u4_word(blk, 2 * ks + 1));
}

}  // namespace qasr

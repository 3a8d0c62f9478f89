// dec_epilogue.h -- epilogues of the decode-step skinny GEMMs, shared by the bf16 (dec_gemv.hip, dec_gemv_wide.hip) and the MLX-quantised
// (dec_quant.hip) kernels.  A lane holds, per (row tile t, batch tile b), four consecutive outputs n of one batch row.
#pragma once
#include "dec_kernels.h"

namespace qasr {

// accumulator layout: acc[t][b][j] = out[batch b*16 + fr][n0 + t*16 + fc*4 + j]
template <int NT, int NB, int EPI>
__device__ __forceinline__ void dec_epilogue(const DecGemvArgs& a, f32x4 (&acc)[NT][NB], int n0, int fr, int fc,
                                             const uint2 (*resid)[NB] = nullptr) {
    if (EPI == DEC_EPI_BF16 || EPI == DEC_EPI_RESID) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int row = b * 16 + fr;
                if (row < a.B) {
                    bf16_t* p = a.out + (long)row * a.N + n0 + t * 16 + fc * 4;
                    float4 v = make_float4(acc[t][b][0], acc[t][b][1], acc[t][b][2], acc[t][b][3]);
                    if (EPI == DEC_EPI_RESID) {
                        float4 r;
                        if (resid) {      // residual fetched at kernel start (saves a memory round trip)
                            const uint2 u = resid[t][b];
                            r = make_float4(bf16_to_f32((bf16_t)(u.x & 0xffff)), bf16_to_f32((bf16_t)(u.x >> 16)),
                                            bf16_to_f32((bf16_t)(u.y & 0xffff)), bf16_to_f32((bf16_t)(u.y >> 16)));
                        } else {
                            r = load_bf16x4(p);
                        }
                        v.x = r.x + bf16_round(v.x); v.y = r.y + bf16_round(v.y);
                        v.z = r.z + bf16_round(v.z); v.w = r.w + bf16_round(v.w);
                    }
                    *reinterpret_cast<uint2*>(p) = pack_bf16x4(v);
                }
            }
    } else if (EPI == DEC_EPI_SWIGLU) {
        // rows come in blocks of 32: 16 gate rows then the 16 matching up rows -> tile pairs (2i, 2i+1)
#pragma unroll
        for (int t = 0; t + 1 < NT; t += 2)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int row = b * 16 + fr;
                if (row < a.B) {
                    float4 v;
                    v.x = swiglu_bf16(acc[t][b][0], acc[t + 1][b][0]);
                    v.y = swiglu_bf16(acc[t][b][1], acc[t + 1][b][1]);
                    v.z = swiglu_bf16(acc[t][b][2], acc[t + 1][b][2]);
                    v.w = swiglu_bf16(acc[t][b][3], acc[t + 1][b][3]);
                    bf16_t* p = a.out + (long)row * (a.N / 2) + (n0 + t * 16) / 2 + fc * 4;
                    *reinterpret_cast<uint2*>(p) = pack_bf16x4(v);
                }
            }
    } else {   // DEC_EPI_LOGITS: bf16-rounded logits, per-block argmax with lowest-index ties
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int row = b * 16 + fr;
            float best = -INFINITY;
            int bidx = 0x7fffffff;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + t * 16 + fc * 4 + j;
                    const float v = bf16_round(acc[t][b][j]);
                    if (a.logits && row < a.B) a.logits[(long)row * a.N + n] = v;
                    if (v > best || (v == best && n < bidx)) { best = v; bidx = n; }
                }
#pragma unroll
            for (int ofs = 16; ofs < 64; ofs <<= 1) {
                float ov = __shfl_xor(best, ofs, 64);
                int oi = __shfl_xor(bidx, ofs, 64);
                if (ov > best || (ov == best && oi < bidx)) { best = ov; bidx = oi; }
            }
            if (fc == 0 && row < a.B) {
                a.part_val[(long)row * gridDim.x + blockIdx.x] = best;
                a.part_idx[(long)row * gridDim.x + blockIdx.x] = bidx;
            }
        }
    }
}

}  // namespace qasr

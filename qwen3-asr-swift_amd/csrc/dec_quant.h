// dec_quant.h -- decode-step kernels on MLX affine-quantised weights (4 / 8 bit, group 64), SURVEY.md section 8f N1 / R6.
//
// Reference: `QuantizedLinear` / `quantizedMatmul` (Sources/Qwen3ASR/QuantizedTextDecoder.swift:33-44,111-136) and
// `PreQuantizedEmbedding` (Sources/MLXCommon/PreQuantizedEmbedding.swift:35-49).  The checkpoint holds, per Linear and
// for the tied embedding, `weight` uint32 [N][K * bits / 32] (element i of a row in word i / (32 / bits), LSB first),
// `scales`, `biases` [N][K / 64]; the effective weight is scale * q + bias per 64-element group.
//
// With ONE row of x per sequence (every decode step) the reference's backend evaluates
//     y[n] = sum_g ( scale[n][g] * sum_{k in g} q[n][k] x[k]  +  bias[n][g] * sum_{k in g} x[k] )        (mlx qmv)
// in f32 and never rounds the dequantised weight.  The kernels here compute exactly that form: q (an integer < 256, exact
// in bf16) goes through the bf16 MFMA against x, per 64-element group the f32 partial sum is scaled and the bias term
// added on the vector unit.  The packed weights stay packed in HBM (3.6x / 1.9x fewer bytes per step than bf16).
//
// HBM images built once by qasr_finalize (quant_pack_launch):
//   q image   uint32, fragment-major: block (16-row tile, BLK columns) = 64 lanes x 16 bytes, lane l = (row l & 15,
//             k offset 8 (l >> 4)); BLK = 128 columns (4 bit: word i of the lane's uint4 = k-step i of the block, its
//             nibbles re-ordered to e0 e2 e4 e6 e1 e3 e5 e7 so that a shift + and-or yields a packed bf16 pair -- see
//             frag_q4) or 64 columns (8 bit: words 2i, 2i+1 = k-step i).  One wave instruction reads 1 KiB contiguous.
//   sb image  [tile][scales | biases][row 16][group G] in the checkpoint's dtype (bf16) or f32 (f16 / f32 checkpoints):
//             a lane reads the G values of its row as 16-byte loads.
#pragma once
#include "dec_kernels.h"

namespace qasr {

struct QuantImg {            // decode-step images of one [N][K] matrix
    const uint32_t* qp = nullptr;
    const void* sb = nullptr;
    int sb_f32 = 0, bits = 0;
    QuantRaw raw{};          // kept for the generic (untuned-shape) kernel
};

// bytes of the two images
size_t quant_q_bytes(int N, int K, int bits);
size_t quant_sb_bytes(int N, int K, int sb_f32);
void quant_pack_launch(const QuantRaw& src, uint32_t* qp, void* sb, hipStream_t s);

// out[n][k] = bf16(scale * q + bias): the weight the reference's prompt-pass kernel (qmm_t) multiplies by, and the row
// `dequantized()` returns for the embedding lookup.  rows [r0, r0 + nrows) of src -> out rows [0, nrows), ld = K; with block > 0,
// source row i lands at out row (i / block) * block_stride + block_off + i % block (gate | up interleaved in 16-row blocks).
void quant_dequant_rows_launch(const QuantRaw& src, int r0, int nrows, bf16_t* out, hipStream_t s, int block = 0, int block_stride = 0,
                               int block_off = 0);

// the same for up to eight matrices in one launch (block == 0: rows back to back)
struct DequantJob {
    QuantRaw q;
    int r0 = 0, nrows = 0;
    bf16_t* out = nullptr;
    int block = 0, block_stride = 0, block_off = 0;
};
struct DequantJobs {
    enum { MAX = 8 };
    DequantJob job[MAX];
    int first_block[MAX];
    int n;
};
void quant_dequant_multi_launch(const DequantJob* jobs, int n, hipStream_t s);

// x[p] = audio_src[p] >= 0 ? audio[audio_src[p]] : dequantized(embed row ids[p])      (Qwen3ASR.swift:236-244)
void embed_splice_q_launch(const int* ids, const int* audio_src, const QuantRaw& embed, const bf16_t* audio, bf16_t* x,
                           int n_pos, int H, hipStream_t s);
// dst[i] = dequantized(embed row idx[i])
void gather_rows_q_launch(const QuantRaw& embed, const int* row_idx, bf16_t* dst, int n, hipStream_t s);

// Decode-step skinny GEMM on a quantised matrix, fused RMSNorm prologue (norm_w != null) and the epilogues of
// dec_kernels.h (BF16 | RESID | SWIGLU).  a.W / a.Wp are ignored.  Falls back to a generic kernel for shapes without a
// tuned instantiation (norm_scratch [B][K] is used then).
void decode_gemv_q_launch(DecEpi epi, const DecGemvArgs& a, const QuantImg& w, const bf16_t* norm_w, float eps,
                          bf16_t* norm_scratch, hipStream_t s);

// Final RMSNorm + tied LM head on the quantised embedding + per-workgroup argmax partials (layout of lm_head_launch).
int lm_head_q_parts(int N, int K, int bits);
int lm_head_q_launch(const QuantImg& w, const bf16_t* X, const bf16_t* norm_w, float eps, int B, int N, int K, float* logits,
                     float* part_val, int* part_idx, bf16_t* norm_scratch, hipStream_t s);

// one 8-element chunk of dequantized(row): out[j] = bf16(scale * q + bias), chunk c covers elements 8c .. 8c+7
__device__ __forceinline__ uint4 quant_dequant_chunk(const QuantRaw& q, long row, int c) {
    const int G = q.K / 64, g = c >> 3;
    const float s = q.sb_f32 ? reinterpret_cast<const float*>(q.scales)[row * G + g]
                             : bf16_to_f32(reinterpret_cast<const bf16_t*>(q.scales)[row * G + g]);
    const float b = q.sb_f32 ? reinterpret_cast<const float*>(q.biases)[row * G + g]
                             : bf16_to_f32(reinterpret_cast<const bf16_t*>(q.biases)[row * G + g]);
    unsigned e[8];
    if (q.bits == 4) {
        const uint32_t w = q.wq[row * (q.K / 8) + c];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = (w >> (4 * j)) & 0xFu;
    } else {
        const uint2 w = *reinterpret_cast<const uint2*>(q.wq + row * (q.K / 4) + 2 * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) { e[j] = (w.x >> (8 * j)) & 0xFFu; e[4 + j] = (w.y >> (8 * j)) & 0xFFu; }
    }
    uint4 o;
    bf16_t* oe = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
    for (int j = 0; j < 8; ++j) oe[j] = f32_to_bf16(fmaf(s, (float)e[j], b));      // one rounding of scale * q + bias
    return o;
}

}  // namespace qasr

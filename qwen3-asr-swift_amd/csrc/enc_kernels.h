// enc_kernels.h -- audio-encoder kernels other than the GEMM: conv2d1 (C_in = 1), LayerNorm,
// block-diagonal window attention.  Reference: Sources/Qwen3ASR/AudioEncoder.swift:362-511.
#pragma once
#include "common.h"
#include "gemm.h"

namespace qasr {

// One conv "image" = one 100-frame mel chunk of one clip (AudioEncoder.swift:382-406).
struct ChunkMeta {
    int clip;        // batch slot
    int t0;          // first mel frame
    int clen;        // valid mel frames in this chunk (beyond: zero pad, :392-398)
    int w0;          // conv input width of this clip's chunks (100, or clen for a single short chunk)
    int w1, w2, w3;  // widths after conv1/2/3 = conv_len^k(w0); outputs beyond are written as zeros
    int tok_off;     // first packed token of this chunk
    int n_tok;       // valid tokens = conv_len^3(clen) (:443-449)
};

// ---- conv2d1: [img][128][w0] f32 (from the per-clip [128][stride] mel) -> NHWC bf16 [img][H1][W1][C]
// out = gelu(bias + sum_{kh,kw} w[c][kh][kw] * in(2oh-1+kh, 2ow-1+kw)), f32 arithmetic.
void conv1_launch(const float* mel, int mel_stride, int n_mels, const ChunkMeta* chunks, int n_img,
                  const bf16_t* w /*[C][3][3][1]*/, const float* bias, bf16_t* out, int H1, int W1, int C,
                  hipStream_t s);

// ---- LayerNorm over the last dim: x f32 [T][D] -> y bf16 [T][D] (eps, affine bf16 params)
void layernorm_launch(const float* x, const bf16_t* gamma, const bf16_t* beta, bf16_t* y, int T, int D, float eps,
                      hipStream_t s);

// ---- window attention: qkv bf16 [T][3D] (q | k | v), windows given by cu_seqlens; out bf16 [T][D].
// softmax(q k^T / sqrt(hd)) v inside each window (== the additive -1e9 block mask of :337-357).
// P is rounded to bf16 before the PV product (MFMA operands).  Window length <= 128.
void window_attention_launch(const bf16_t* qkv, const int* cu_seqlens, int n_windows, int heads, int head_dim,
                             bf16_t* out, hipStream_t s);

// ---- epilogues used only by the encoder ---------------------------------------------------------
// conv2/conv3: out = gelu(acc + bias) as bf16, zero beyond the image's valid output width
struct EpiConvGelu {
    bf16_t* out; long ldo; const float* bias; const ChunkMeta* chunks;
    int OH, OW; bool hw_major; int level;      // level 2 -> w2, 3 -> w3
    struct Pre { float4 b; int wv; };
    __device__ __forceinline__ Pre prefetch(int m, int n) const {
        const int img = m / (OH * OW);
        return {*reinterpret_cast<const float4*>(bias + n), level == 2 ? chunks[img].w2 : chunks[img].w3};
    }
    __device__ __forceinline__ void apply(int m, int n, float4 v, const Pre& p) const {
        const int img = m / (OH * OW), rem = m - img * (OH * OW);
        const int ow = hw_major ? rem % OW : rem / OH;
        const float4 b = p.b;
        v.x = gelu_erf(v.x + b.x); v.y = gelu_erf(v.y + b.y); v.z = gelu_erf(v.z + b.z); v.w = gelu_erf(v.w + b.w);
        if (ow >= p.wv) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<uint2*>(out + (long)m * ldo + n) = pack_bf16x4(v);
    }
    __device__ __forceinline__ void operator()(int m, int n, float4 v) const { apply(m, n, v, prefetch(m, n)); }
};

// conv_out: x_f32[token][n] = acc + pe[t_in_chunk(token)][n]   (AudioEncoder.swift:427-439)
struct EpiPosF32 {
    float* x; long ldx; const float* pe; const int* tok_t;
    __device__ __forceinline__ void operator()(int m, int n, float4 v) const {
        float4 p = *reinterpret_cast<const float4*>(pe + (long)tok_t[m] * ldx + n);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        *reinterpret_cast<float4*>(x + (long)m * ldx + n) = v;
    }
};

}  // namespace qasr

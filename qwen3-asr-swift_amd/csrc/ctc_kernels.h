// ctc_kernels.h -- kernels of the Omnilingual ASR path (wav2vec2 encoder + CTC head) that the Qwen3 path does not have.
// Reference: Sources/OmnilingualASR/MLX/{Wav2Vec2Frontend,Wav2Vec2EncoderLayer,Wav2Vec2Encoder,OmnilingualMLXModel}.swift.
// The reference widens every float tensor to f32 at load (OmnilingualMLXWeightLoader.swift:26-37): LayerNorm / bias
// parameters are f32 here too, the contractions run on the bf16 MFMA (operands rounded to bf16, f32 accumulate) -- the same
// stated deviation as the Qwen3 audio encoder (oracle policy DEVICE).
#pragma once
#include "common.h"
#include "gemm.h"

namespace qasr {

// Per clip: layerNormalize statistics (OmnilingualASR.swift:305-325): mean and 1 / sqrt(max(0, E[x^2] - mean^2) + eps)
// stats[2b] = mean, stats[2b + 1] = inv_std.  One workgroup per clip.
void wave_stats_launch(const float* pcm, const long* pcm_off, const int* n_samples, int B, float eps, float* stats,
                       hipStream_t s);

// Feature-extractor layer 0: Conv1d(1 -> C, k = 10, s = 5) on the normalised samples -> LayerNorm(C) -> GELU -> bf16
// [frames][C].  frame_off[b] = first packed output row of clip b; n_out[b] = its output length.
void w2v_conv0_launch(const float* pcm, const long* pcm_off, const float* stats, const int* frame_off, const int* n_out, int B,
                      int max_out, const float* w /*[C][10]*/, const float* bias, const float* ln_g, const float* ln_b, float eps,
                      bf16_t* out, int C, hipStream_t s);

// rows of a strided Conv1d over channel-last activations as an implicit GEMM: output frame m of clip b reads the
// contiguous slice in[(in_off[b] + stride * t) * C .. + k * C).  Fills row_off[m] (element offsets) for ARowTable.
void w2v_conv_rows_launch(const int* in_off, const int* out_off, const int* n_out, int B, int total_out, int stride, int C,
                          long* row_off, hipStream_t s);

// y_bf16[row] = gelu(LayerNorm(x_f32[row]))  (ACT = 1)  or  LayerNorm(x_f32[row])  (ACT = 0); f32 affine parameters
void layernorm_f32p_launch(const float* x, const float* gamma, const float* beta, bf16_t* y, int rows, int D, float eps, int act,
                           hipStream_t s);

// y_f32[row] = gelu(LayerNorm(x_f32[row])): the last feature-extractor layer, whose output feeds another LayerNorm in f32
void layernorm_gelu_f32_launch(const float* x, const float* gamma, const float* beta, float* y, int rows, int D, float eps,
                               hipStream_t s);

void cast_f32_bf16_launch(const float* x, bf16_t* y, long n, hipStream_t s);

// per packed frame m: (t, L) of its clip -> valid tap range of the positional conv; filled from the clip tables
void w2v_frame_info_launch(const int* frame_off, const int* n_frames, int B, int total, int2* info, hipStream_t s);

// Positional encoder (Wav2Vec2Frontend.swift:88-122): grouped Conv1d, kernel KP, padding KP / 2, trailing frame trimmed.
// As an implicit GEMM per group: row m = packed frame, K index = tap * cpg + ci.
struct AGroupConv1d {
    const bf16_t* x;        // [frames][D] bf16
    const int2* info;       // per frame (t, L)
    int D, cpg, KP, g, M;
    static constexpr bool no_p8 = true;   // N = D / 16 columns per launch: a 256-wide tile would be mostly padding anyway
    struct Row { const bf16_t* base; int kmin, kmax; };
    __device__ __forceinline__ AGroupConv1d for_group(int grp) const { AGroupConv1d a = *this; a.g = grp; return a; }
    __device__ __forceinline__ Row row_init(int m) const {
        if (m >= M) return {nullptr, 0, 0};
        const int2 tl = info[m];
        const int pad = KP / 2;
        const int kmin = pad - tl.x > 0 ? pad - tl.x : 0;                 // source frame t - pad + tap >= 0
        const int kmax = tl.y - tl.x + pad < KP ? tl.y - tl.x + pad : KP;  // ... < L
        return {x + ((long)m - pad) * D + g * cpg, kmin, kmax};
    }
    __device__ __forceinline__ const bf16_t* addr(const Row& r, int k) const {
        if (!r.base) return nullptr;
        // k / cpg without an integer division (cpg is a run-time value, 64 / 80 / 128 in the published variants): exact for
        // k < 2^20 with the rounded-up reciprocal, like AConv3x3s2::addr
        const int tap = (int)__umulhi((unsigned)k, (0xffffffffu / (unsigned)cpg) + 1u), ci = k - tap * cpg;
        if (tap < r.kmin || tap >= r.kmax) return nullptr;
        return r.base + (long)tap * D + ci;
    }
    __device__ __forceinline__ uint4 load(const Row& r, int k) const {
        const bf16_t* p = addr(r, k);
        return p ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
    }
    struct KT { int k; };                                                  // K-tile form of addr (gemm.h)
    __device__ __forceinline__ KT ktile(int k0, int c8) const { return {k0 + c8}; }
    __device__ __forceinline__ const bf16_t* addr_kt(const Row& r, const KT& t) const { return addr(r, t.k); }
};

// ---- epilogues with f32 parameters ------------------------------------------------------------------
// out_f32[m][n] = acc + bias[n]
struct EpiBiasF32 {
    float* out; long ldo; const float* bias;
    struct Pre { float4 b; };
    __device__ __forceinline__ Pre prefetch(int m, int n) const { return {*reinterpret_cast<const float4*>(bias + n)}; }
    __device__ __forceinline__ void apply(int m, int n, float4 v, const Pre& p) const {
        v.x += p.b.x; v.y += p.b.y; v.z += p.b.z; v.w += p.b.w;
        *reinterpret_cast<float4*>(out + (long)m * ldo + n) = v;
    }
    __device__ __forceinline__ void operator()(int m, int n, float4 v) const { apply(m, n, v, prefetch(m, n)); }
};
// out_bf16[m][n] = act(acc + bias[n]); ACT 0 none, 1 exact GELU
template <int ACT>
struct EpiBiasActBf16F {
    bf16_t* out; long ldo; const float* bias;
    struct Pre { float4 b; };
    __device__ __forceinline__ Pre prefetch(int m, int n) const { return {*reinterpret_cast<const float4*>(bias + n)}; }
    __device__ __forceinline__ void apply(int m, int n, float4 v, const Pre& p) const {
        v.x += p.b.x; v.y += p.b.y; v.z += p.b.z; v.w += p.b.w;
        if (ACT == 1) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        *reinterpret_cast<uint2*>(out + (long)m * ldo + n) = pack_bf16x4(v);
    }
    __device__ __forceinline__ void operator()(int m, int n, float4 v) const { apply(m, n, v, prefetch(m, n)); }
};
// x_f32[m][n] += acc + bias[n]
struct EpiResidF32F {
    float* x; long ldx; const float* bias;
    struct Pre { float4 b, r; };
    __device__ __forceinline__ Pre prefetch(int m, int n) const {
        return {*reinterpret_cast<const float4*>(bias + n), *reinterpret_cast<const float4*>(x + (long)m * ldx + n)};
    }
    __device__ __forceinline__ void apply(int m, int n, float4 v, const Pre& p) const {
        float4 r = p.r;
        r.x += v.x + p.b.x; r.y += v.y + p.b.y; r.z += v.z + p.b.z; r.w += v.w + p.b.w;
        *reinterpret_cast<float4*>(x + (long)m * ldx + n) = r;
    }
    __device__ __forceinline__ void operator()(int m, int n, float4 v) const { apply(m, n, v, prefetch(m, n)); }
};
// positional encoder: y[m][col0 + n] = gelu(acc + bias[col0 + n]) + x[m][col0 + n]
struct EpiPosConv {
    float* y; const float* x; long ld; const float* bias; int col0;
    int cpg = 0;                                            // columns per group (for_group)
    __device__ __forceinline__ EpiPosConv for_group(int grp) const { EpiPosConv e = *this; e.col0 = grp * cpg; return e; }
    __device__ __forceinline__ void operator()(int m, int n, float4 v) const {
        const float4 b = *reinterpret_cast<const float4*>(bias + col0 + n);
        const float4 r = *reinterpret_cast<const float4*>(x + (long)m * ld + col0 + n);
        v.x = gelu_erf(v.x + b.x) + r.x; v.y = gelu_erf(v.y + b.y) + r.y;
        v.z = gelu_erf(v.z + b.z) + r.z; v.w = gelu_erf(v.w + b.w) + r.w;
        *reinterpret_cast<float4*>(y + (long)m * ld + col0 + n) = v;
    }
};

// Full (unmasked) multi-head self-attention over packed clips: qkv bf16 [frames][3D] (q | k | v), clip c = rows
// [cu[c], cu[c+1]); out bf16 [frames][D].  Online softmax over 64-key tiles, P rounded to bf16 for the P V product
// (SDPA.multiHead with mask nil, MLXCommon/SDPA.swift:18-37).  head_dim 64 (every Omnilingual variant) or 16 (tests).
void mha_attention_launch(const bf16_t* qkv, const int* cu, int n_clips, int max_len, int heads, int head_dim, bf16_t* out,
                          hipStream_t s);

// ids[r] = first index of the maximum of x[r][0 .. n)   (CTCGreedyDecoder.swift:39-48: strict '>' keeps the first maximum).
// *err |= 1 if any logit is NaN / infinite (the reference would silently emit whatever its comparisons leave; the engine
// turns this into an error status, like the Qwen3 path's non-finite check).
void argmax_f32_launch(const float* x, long ld, int rows, int n, int* ids, int* err, hipStream_t s);

}  // namespace qasr

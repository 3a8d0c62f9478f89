// dec_kernels.h -- text-decoder kernels (Qwen3 decoder-only LM with GQA, q/k RMSNorm, split-half
// RoPE, SwiGLU, tied LM head).  Reference: Sources/Qwen3ASR/QuantizedTextDecoder.swift:56-251,
// FloatTextDecoder.swift:35-226, Qwen3ASR.swift:236-256,317-390.
//
// bf16 at every op boundary like the reference's MLX tensors (see oracle/precision.py), f32 inside.
#pragma once
#include "common.h"
#include "gemm.h"
#include "dec_rope.h"

namespace qasr {

// A checkpoint triplet of an MLX affine-quantised [N][K] matrix as uploaded (row-major); see dec_quant.h
struct QuantRaw {
    const uint32_t* wq = nullptr;   // [N][K * bits / 32], element i of a row in word i / (32 / bits), LSB first
    const void* scales = nullptr;   // [N][K / 64]
    const void* biases = nullptr;
    int sb_f32 = 0;                 // 0: bf16 elements, 1: f32
    int N = 0, K = 0, bits = 0;
};

// ---- shared small kernels -------------------------------------------------------------------
// y = bf16(w * bf16(x * rsqrt(mean(x^2) + eps)))  over rows of width H (bf16 in / bf16 out)
void rmsnorm_rows_launch(const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int H, float eps, hipStream_t s);

// x[p] = audio_src[p] >= 0 ? audio[audio_src[p]] : embed[ids[p]]      (Qwen3ASR.swift:236-244)
void embed_splice_launch(const int* ids, const int* audio_src, const bf16_t* embed, const bf16_t* audio, bf16_t* x,
                         int n_pos, int H, hipStream_t s);

// gather rows: dst[i] = src[row_idx[i]]
void gather_rows_launch(const bf16_t* src, const int* row_idx, bf16_t* dst, int n, int H, hipStream_t s);

// ---- prefill -------------------------------------------------------------------------------------
struct KVLayout {          // one layer's cache, bf16, per (slot, kv head) a block of max_ctx * hd elements
    bf16_t* k;             // keys [slot][kv_head][max_ctx][hd]
    bf16_t* v;             // unused since round 3 (was a row-major V scratch of the prompt pass; v_transpose now reads V from qkv): null
    int max_ctx, kv_heads, hd;
    // the values the decode sweep reads: MFMA-fragment order [key/32][d/16][lane 64][8], see vfrag_index() in
    // dec_attention.hip; written by v_transpose (prompt) and decode_attention (append)
    bf16_t* vf = nullptr;
    __host__ __device__ long off(int slot, int kvh, int pos) const {
        return (((long)slot * kv_heads + kvh) * max_ctx + pos) * hd;
    }
};

// Per packed prompt position p: q/k RMSNorm over head_dim, RoPE at position pos[p], then
//   q  -> qr[p][head][hd]; k -> cache.k[slot][kvh][pos]; v (no arithmetic) -> cache.vf fragments and, where the prompt attention reads V^T, vt[slot][kvh][d][pos]
// qkv: [n_pos][(heads + 2 kv) * hd]  (q | k | v).   rope tables: [max_pos][hd/2] f32.
void qk_norm_rope_launch(const bf16_t* qkv, const int* slot, const int* pos, int n_pos, int heads, int kv_heads,
                         int hd, const bf16_t* qn_w, const bf16_t* kn_w, float eps, const float* rope_cos,
                         const float* rope_sin, bf16_t* qr, KVLayout cache, bf16_t* vt, int vt_stride,
                         const int* cu, const int* slot_of_clip, int n_clips, int max_len, hipStream_t s, bool v_only = false);
// true when the q|k|v projection may run as a head-tile GEMM with EpiQkHeads (then qk_norm_rope_launch(..., v_only = true) writes the V images)
bool qk_norm_rope_fusable(int heads, int kv_heads, int hd);

// Causal flash attention over packed prompts.  clip c occupies packed rows [cu[c], cu[c+1]).
// Online softmax in key tiles of 64 with unnormalised P rounded to bf16 (restated in
// oracle/decoder.py: flash_prefill_attention).  out: [n_pos][heads*hd] bf16.
void prefill_attention_launch(const bf16_t* qr, KVLayout cache, const bf16_t* vt, int vt_stride, const int* cu,
                              const int* slot_of_clip, int n_clips, int max_len, int heads, bf16_t* out,
                              hipStream_t s, unsigned long long* dbg = nullptr);

// ---- decode step (M = batch rows) -----------------------------------------------------------------
enum DecEpi { DEC_EPI_BF16 = 0, DEC_EPI_RESID = 1, DEC_EPI_SWIGLU = 2, DEC_EPI_LOGITS = 3 };

struct DecGemvArgs {
    const bf16_t* W;       // [N][K] row-major
    const bf16_t* Wp;      // same weight, fragment-major (pack_mfma_a_launch) for the tuned kernels; may be null
    const bf16_t* X;       // [B][K] bf16 activations
    int B, N, K;
    bf16_t* out;           // BF16: [B][N]; RESID: x in place [B][N]; SWIGLU: [B][N/2]
    float* logits;         // LOGITS: optional [B][N] f32 (bf16-rounded values), may be null
    float* part_val;       // LOGITS: [B][n_blocks]
    int* part_idx;         // LOGITS: [B][n_blocks]
};
// Weight-streaming skinny GEMM: every weight byte is read once, straight into MFMA A fragments.
// Returns the number of workgroups (LOGITS: partial count per row).
int decode_gemv_launch(DecEpi epi, const DecGemvArgs& a, hipStream_t s);
int decode_gemv_blocks(DecEpi epi, int N);
// Decode-step form with the RMSNorm of the input fused into the activation staging (norm_w != null):
// out = epi( rmsnorm(X) . W^T ).  `norm_scratch` [B][K] is only used by the generic fallback.
// out[r] = lowest index of the maximum of x[r][0..n) (bf16 logits, MLX argMax tie rule)
void argmax_rows_launch(const bf16_t* x, long ld, int rows, int n, int* out, hipStream_t s);

// diagnostic: the tuned kernels stamp their phases into dbg[(workgroup*16 + wave)*8 + i] (100 MHz clock); null = off
void decode_gemv_set_debug(unsigned long long* dbg);
int decode_gemv_fused_launch(DecEpi epi, const DecGemvArgs& a, const bf16_t* norm_w, float eps, bf16_t* norm_scratch,
                             hipStream_t s);

// Final RMSNorm + tied LM head + argmax partials (persistent kernel where the shape allows, else the fused
// GEMV).  lm_head_parts = partials per row the launch will produce (fixed per (N, K)); partial layout
// [row][parts].  NOTE: the persistent form needs every row group of a step to use the same `parts`.
int lm_head_parts(int N, int K);
int lm_head_launch(const bf16_t* W, const bf16_t* Wp, const bf16_t* X, const bf16_t* norm_w, float eps, int B, int N, int K,
                   float* logits, float* part_val, int* part_idx, bf16_t* norm_scratch, hipStream_t s);
// Fragment-major repack of a row-major [N][K] weight (MFMA 16x16x32 A operand, 1 KiB per (row tile, k-step))
void pack_mfma_a_launch(const bf16_t* src, bf16_t* dst, int N, int K, hipStream_t s);

// One new token per batch row: q/k norm + RoPE at pos = ctx_len[b] (rope_cos/rope_sin are the per-row
// RopeRows of that position), append K/V to the cache,
// attention of the rep = heads/kv_heads query heads over the cache (f32 softmax), out [B][heads*hd].
void decode_attention_launch(const bf16_t* qkv, const int* ctx_len, int B, int heads, int kv_heads, int hd,
                             const bf16_t* qn_w, const bf16_t* kn_w, float eps, const float* rope_cos,
                             const float* rope_sin, KVLayout cache, bf16_t* out, hipStream_t s,
                             unsigned long long* dbg = nullptr);   // dbg: diagnostic phase stamps, null in product launches

// Greedy bookkeeping after the LM head (Qwen3ASR.swift:336-388): argmax over the per-block partials
// (lowest index wins ties, like MLX argMax), append the token unless the row is finished, mark EOS /
// length-cap, advance ctx_len (decode steps), and gather the next input embedding.
struct GreedyState {
    int* tokens;        // [B][max_new + 1]
    int* lens;          // [B]
    int* finished;      // [B]
    int* ctx_len;       // [B]
    int* n_active;      // [1] rows not finished
    int* err;           // [1] set to 1 when a row's best logit is not finite (NaN / inf logits): qasr_batch_tokens reports it
    int max_new;        // row stride - 1
    int max_tokens;     // cap for this batch
    int eos;
    int ignore_eos;
    int vocab;          // ids outside [0, vocab) (all-NaN logits) are clamped to 0 so the gather cannot fault, and *err is set
    unsigned* clear = nullptr;   // (the word 32 behind them is the step sequence word, incremented instead)  words zeroed by every finalize launch: the arrival counters of the NEXT step's fused launches (dec_chain.h) --
    int clear_words = 0;         // a greedy step is always preceded by a finalize, so its graph needs no memset node (4.7 us per step)
};
// rope rows of the next decode position, one per batch row (cos_rows/sin_rows [B][half]), copied from the
// position-indexed tables by greedy_finalize so decode attention does not chain ctx_len -> table lookup
struct RopeRows {
    const float* cos_table; const float* sin_table;
    float* cos_rows; float* sin_rows;
    int half;
};
// qembed != null: the next-token embedding is `dequantized(row)` of the quantised table (PreQuantizedEmbedding.swift:35-42)
void greedy_finalize_launch(const float* part_val, const int* part_idx, int n_parts, GreedyState st, int B,
                            int advance_ctx, const bf16_t* embed, bf16_t* x, int H, RopeRows rr, hipStream_t s,
                            const QuantRaw* qembed = nullptr);

// Epilogue of the prompt pass's q|k|v projection as a head-tile GEMM (gemm.h MODE 2, head_dim 128): per packed position and head, q/k RMSNorm
// over the head, RoPE at pos[m], q -> qr[m][head][128], k -> cache.k[slot[m]][kvh][pos[m]] -- the arithmetic of qk_norm_rope_wide_kernel
// (dec_prefill.hip), element for element and in the same order, on the tile while it is still in LDS instead of a second pass over the
// projection's output (160 MB of traffic and a launch per layer).  v heads are stored as before (qkv buffer) for the V image kernel.
struct EpiQkHeads {
    bf16_t* qkv; long ldo;                 // v tiles: plain bf16 store at [m][n] like EpiStoreBf16
    bf16_t* qr;
    KVLayout cache;
    const int* slot; const int* pos;
    const bf16_t* qn_w; const bf16_t* kn_w;
    float eps;
    const float* rope_cos; const float* rope_sin;
    int heads, kv_heads;
    __device__ __forceinline__ bool head_tile(int n0) const { return n0 < (heads + kv_heads) * 128; }
    __device__ __forceinline__ void operator()(int m, int n, float4 v) const {
        *reinterpret_cast<uint2*>(qkv + (long)m * ldo + n) = pack_bf16x4(v);
    }
    __device__ __forceinline__ void rows(int m0, int n0, int M, const unsigned short* T, int tid) const {
        constexpr int HD = 128, HALF = 64;
        const int h = n0 / HD, j = tid & 7;
        const bf16_t* nw = h < heads ? qn_w : kn_w;
        const uint4 w1 = *reinterpret_cast<const uint4*>(nw + 8 * j), w2 = *reinterpret_cast<const uint4*>(nw + HALF + 8 * j);
        const bf16_t* w1e = reinterpret_cast<const bf16_t*>(&w1);
        const bf16_t* w2e = reinterpret_cast<const bf16_t*>(&w2);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 32 + (tid >> 3);
            const bool live = m0 + row < M;
            const int m = live ? m0 + row : M - 1;
            const int sw = (row >> 2) & 7;
            const uint4 a = *reinterpret_cast<const uint4*>(T + row * 128 + ((j ^ sw) << 3));
            const uint4 b = *reinterpret_cast<const uint4*>(T + row * 128 + (((8 + j) ^ sw) << 3));
            const bf16_t* ae = reinterpret_cast<const bf16_t*>(&a);
            const bf16_t* be = reinterpret_cast<const bf16_t*>(&b);
            const int sl = slot[m], ps = pos[m];
            float x1[8], x2[8], ss = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                x1[e] = bf16_to_f32(ae[e]);
                x2[e] = bf16_to_f32(be[e]);
                ss += x1[e] * x1[e] + x2[e] * x2[e];
            }
#pragma unroll
            for (int ofs = 1; ofs < 8; ofs <<= 1) ss += __shfl_xor(ss, ofs, 64);
            const float inv = rsqrtf(ss / (float)HD + eps);
            const float4* cp = reinterpret_cast<const float4*>(rope_cos + (long)ps * HALF + 8 * j);
            const float4* sp = reinterpret_cast<const float4*>(rope_sin + (long)ps * HALF + 8 * j);
            const float4 c0 = cp[0], c1 = cp[1], s0 = sp[0], s1 = sp[1];
            const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            uint4 o1, o2;
            bf16_t* o1e = reinterpret_cast<bf16_t*>(&o1);
            bf16_t* o2e = reinterpret_cast<bf16_t*>(&o2);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float r1, r2;
                norm_rope_pair(x1[e], x2[e], bf16_to_f32(w1e[e]), bf16_to_f32(w2e[e]), inv, cs[e], sn[e], r1, r2);
                o1e[e] = f32_to_bf16(r1);
                o2e[e] = f32_to_bf16(r2);
            }
            if (live) {
                bf16_t* dst = h < heads ? qr + ((long)m * heads + h) * HD : cache.k + cache.off(sl, h - heads, ps);
                *reinterpret_cast<uint4*>(dst + 8 * j) = o1;
                *reinterpret_cast<uint4*>(dst + HALF + 8 * j) = o2;
            }
        }
    }
};

// ---- epilogues for the prefill GEMMs ----------------------------------------------------------------
// x_bf16[m][n] = bf16(x + bf16(acc))   (residual add in the decoder dtype)
struct EpiResidBf16 {
    bf16_t* x; long ldx;
    struct Pre { uint2 r; };
    __device__ __forceinline__ Pre prefetch(int m, int n) const { return {*reinterpret_cast<const uint2*>(x + (long)m * ldx + n)}; }
    __device__ __forceinline__ void apply(int m, int n, float4 v, const Pre& p) const {
        float4 r = unpack_bf16x4(p.r);
        r.x += bf16_round(v.x); r.y += bf16_round(v.y); r.z += bf16_round(v.z); r.w += bf16_round(v.w);
        *reinterpret_cast<uint2*>(x + (long)m * ldx + n) = pack_bf16x4(r);
    }
    __device__ __forceinline__ void operator()(int m, int n, float4 v) const { apply(m, n, v, prefetch(m, n)); }
};

// QuantizedTextDecoder.swift:134-136 with bf16 tensors: silu(gate) -> bf16, * up -> bf16
__device__ __forceinline__ float swiglu_bf16(float g_acc, float u_acc) { return gemm_swiglu(g_acc, u_acc); }

}  // namespace qasr

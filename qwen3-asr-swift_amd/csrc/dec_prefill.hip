// dec_prefill.hip -- prompt pass of the text decoder: RMSNorm rows, embedding splice / row gather, q/k norm + RoPE + cache fill, V transpose, causal prompt attention (declarations: dec_kernels.h).
#include "dec_kernels.h"
#include "dec_rope.h"
#include <cstdlib>
#include <cstdio>

namespace qasr {

// ------------------------------------------------------------------------------------------------
// RMSNorm rows: one wavefront per row.  y = bf16(w * bf16(x * inv)), inv = rsqrt(mean(x^2) + eps)
// ------------------------------------------------------------------------------------------------
constexpr int RMS_MAXV = 4;   // 8-element chunks per lane: H <= 2048

__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                           bf16_t* __restrict__ y, int rows, int H, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nch = H / 8;
    const uint4* xr = reinterpret_cast<const uint4*>(x + (long)row * H);
    uint4 v[RMS_MAXV];
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < RMS_MAXV; ++i) {
        int c = lane + 64 * i;
        v[i] = c < nch ? xr[c] : make_uint4(0, 0, 0, 0);
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&v[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) { float f = bf16_to_f32(e[j]); ss = fmaf(f, f, ss); }
    }
    const float inv = rsqrtf(wave_sum(ss) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < RMS_MAXV; ++i) {
        int c = lane + 64 * i;
        if (c < nch) {
            uint4 wv = reinterpret_cast<const uint4*>(w)[c];
            const uint4 o = make_uint4(rmsnorm_pair_bf16(v[i].x, wv.x, inv), rmsnorm_pair_bf16(v[i].y, wv.y, inv),
                                       rmsnorm_pair_bf16(v[i].z, wv.z, inv), rmsnorm_pair_bf16(v[i].w, wv.w, inv));
            reinterpret_cast<uint4*>(y + (long)row * H)[c] = o;
        }
    }
}

void rmsnorm_rows_launch(const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int H, float eps, hipStream_t s) {
    if (rows <= 0) return;
    if (H % 8 != 0 || H > 512 * RMS_MAXV) throw std::invalid_argument("rmsnorm: unsupported width");
    hipLaunchKernelGGL(rmsnorm_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, w, y, rows, H, eps);
}

// ------------------------------------------------------------------------------------------------
// embedding lookup with audio splice; row gather
// ------------------------------------------------------------------------------------------------
__global__ void embed_splice_kernel(const int* __restrict__ ids, const int* __restrict__ audio_src,
                                    const bf16_t* __restrict__ embed, const bf16_t* __restrict__ audio,
                                    bf16_t* __restrict__ x, int H) {
    const int p = blockIdx.x;
    const int a = audio_src[p];
    const uint4* src = reinterpret_cast<const uint4*>(a >= 0 ? audio + (long)a * H : embed + (long)ids[p] * H);
    uint4* dst = reinterpret_cast<uint4*>(x + (long)p * H);
    for (int i = threadIdx.x; i < H / 8; i += blockDim.x) dst[i] = src[i];
}

void embed_splice_launch(const int* ids, const int* audio_src, const bf16_t* embed, const bf16_t* audio, bf16_t* x,
                         int n_pos, int H, hipStream_t s) {
    if (n_pos <= 0) return;
    hipLaunchKernelGGL(embed_splice_kernel, dim3(n_pos), dim3(128), 0, s, ids, audio_src, embed, audio, x, H);
}

__global__ void gather_rows_kernel(const bf16_t* __restrict__ src, const int* __restrict__ idx, bf16_t* __restrict__ dst, int H) {
    const uint4* s = reinterpret_cast<const uint4*>(src + (long)idx[blockIdx.x] * H);
    uint4* d = reinterpret_cast<uint4*>(dst + (long)blockIdx.x * H);
    for (int i = threadIdx.x; i < H / 8; i += blockDim.x) d[i] = s[i];
}

void gather_rows_launch(const bf16_t* src, const int* row_idx, bf16_t* dst, int n, int H, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(128), 0, s, src, row_idx, dst, H);
}

// ------------------------------------------------------------------------------------------------
// q/k RMSNorm + RoPE + cache write for packed prompt positions.  One wavefront per (position, head).
// Lane l owns the rotation pair (l, l + hd/2).
// ------------------------------------------------------------------------------------------------
// HPW = 256 / HD heads per wavefront; a head lives on HD/4 lanes, lane j of a head owns elements
// (2j, 2j+1) of the first half and the matching pair of the second half (4-byte accesses).
template <int HD>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ slot,
                                                           const int* __restrict__ pos, int n_pos, int heads,
                                                           int kv_heads, const bf16_t* __restrict__ qn_w,
                                                           const bf16_t* __restrict__ kn_w, float eps,
                                                           const float* __restrict__ rope_cos,
                                                           const float* __restrict__ rope_sin, bf16_t* __restrict__ qr,
                                                           KVLayout cache) {
    constexpr int LPH = HD / 4, HPW = 64 / LPH, HALF = HD / 2;
    const int nh = heads + 2 * kv_heads;
    const int groups = nh / HPW;                        // head groups per position
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wid >= (long)n_pos * groups) return;
    const int p = (int)(wid / groups), h = (int)(wid - (long)p * groups) * HPW + lane / LPH;
    const int j = lane % LPH;
    const bf16_t* src = qkv + (long)p * nh * HD + (long)h * HD;
    const int sl = slot[p], ps = pos[p];
    const unsigned a = *reinterpret_cast<const unsigned*>(src + 2 * j);
    const unsigned bb = *reinterpret_cast<const unsigned*>(src + HALF + 2 * j);
    if (h >= heads + kv_heads) return;                  // V needs no arithmetic: v_transpose_kernel reads it from qkv
    const float x1a = bf16_to_f32((bf16_t)(a & 0xffff)), x1b = bf16_to_f32((bf16_t)(a >> 16));
    const float x2a = bf16_to_f32((bf16_t)(bb & 0xffff)), x2b = bf16_to_f32((bf16_t)(bb >> 16));
    float ss = (x1a * x1a + x1b * x1b) + (x2a * x2a + x2b * x2b);
#pragma unroll
    for (int ofs = 1; ofs < LPH; ofs <<= 1) ss += __shfl_xor(ss, ofs, 64);
    const float inv = rsqrtf(ss / (float)HD + eps);
    const bf16_t* nw = h < heads ? qn_w : kn_w;
    const unsigned w1 = *reinterpret_cast<const unsigned*>(nw + 2 * j), w2 = *reinterpret_cast<const unsigned*>(nw + HALF + 2 * j);
    const float2 cs = *reinterpret_cast<const float2*>(rope_cos + (long)ps * HALF + 2 * j);
    const float2 sn = *reinterpret_cast<const float2*>(rope_sin + (long)ps * HALF + 2 * j);
    float o1a, o2a, o1b, o2b;
    norm_rope_pair(x1a, x2a, bf16_to_f32((bf16_t)(w1 & 0xffff)), bf16_to_f32((bf16_t)(w2 & 0xffff)), inv, cs.x, sn.x, o1a, o2a);
    norm_rope_pair(x1b, x2b, bf16_to_f32((bf16_t)(w1 >> 16)), bf16_to_f32((bf16_t)(w2 >> 16)), inv, cs.y, sn.y, o1b, o2b);
    bf16_t* dst = h < heads ? qr + ((long)p * heads + h) * HD : cache.k + cache.off(sl, h - heads, ps);
    *reinterpret_cast<unsigned*>(dst + 2 * j) = pack_bf16x2(o1a, o1b);
    *reinterpret_cast<unsigned*>(dst + HALF + 2 * j) = pack_bf16x2(o2a, o2b);
}

// Wide form: a head lives on HD/16 lanes, each owning 8 consecutive elements of the first half and the matching 8 of the
// second half (16-byte accesses); a wave covers 64 / (HD/16) heads.  Same arithmetic per element as the form above; a
// quarter of the waves (the narrow form's 208 000 four-byte-per-lane waves per launch were bound by wave launch rate).
template <int HD>
__global__ __launch_bounds__(256) void qk_norm_rope_wide_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ slot,
                                                                const int* __restrict__ pos, int n_pos, int heads,
                                                                int kv_heads, const bf16_t* __restrict__ qn_w,
                                                                const bf16_t* __restrict__ kn_w, float eps,
                                                                const float* __restrict__ rope_cos,
                                                                const float* __restrict__ rope_sin, bf16_t* __restrict__ qr,
                                                                KVLayout cache) {
    constexpr int LPH = HD / 16, HPW = 64 / LPH, HALF = HD / 2;
    const int nh = heads + 2 * kv_heads;
    // head groups per position: the q and k heads only (host checks (heads + kv_heads) % HPW == 0) -- V needs no arithmetic, the
    // transposing kernel below reads it from qkv directly
    const int groups = (heads + kv_heads) / HPW;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wid >= (long)n_pos * groups) return;
    const int p = (int)(wid / groups), h = (int)(wid - (long)p * groups) * HPW + lane / LPH;
    const int j = lane % LPH;
    const bf16_t* src = qkv + (long)p * nh * HD + (long)h * HD;
    const int sl = slot[p], ps = pos[p];
    const uint4 a = *reinterpret_cast<const uint4*>(src + 8 * j);
    const uint4 b = *reinterpret_cast<const uint4*>(src + HALF + 8 * j);
    const bf16_t* ae = reinterpret_cast<const bf16_t*>(&a);
    const bf16_t* be = reinterpret_cast<const bf16_t*>(&b);
    float x1[8], x2[8], ss = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        x1[e] = bf16_to_f32(ae[e]);
        x2[e] = bf16_to_f32(be[e]);
        ss += x1[e] * x1[e] + x2[e] * x2[e];
    }
#pragma unroll
    for (int ofs = 1; ofs < LPH; ofs <<= 1) ss += __shfl_xor(ss, ofs, 64);
    const float inv = rsqrtf(ss / (float)HD + eps);
    const bf16_t* nw = h < heads ? qn_w : kn_w;
    const uint4 w1 = *reinterpret_cast<const uint4*>(nw + 8 * j), w2 = *reinterpret_cast<const uint4*>(nw + HALF + 8 * j);
    const bf16_t* w1e = reinterpret_cast<const bf16_t*>(&w1);
    const bf16_t* w2e = reinterpret_cast<const bf16_t*>(&w2);
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
    bf16_t* dst = h < heads ? qr + ((long)p * heads + h) * HD : cache.k + cache.off(sl, h - heads, ps);
    *reinterpret_cast<uint4*>(dst + 8 * j) = o1;
    *reinterpret_cast<uint4*>(dst + HALF + 8 * j) = o2;
}

// V images for the prompt pass and the decode sweep, 64 positions per workgroup, transposed through LDS so both sides move 128-byte
// rows: the V third of qkv (packed prompt rows cu[clip] + pos, head heads + kv_heads + kvh) -> vt[slot][kvh][d][pos] (only when the
// prompt attention reads V^T: engines without the fragment image, or the pa_vfrag = 0 A/B) and -> cache.vf (fragment-major).
template <int HD>
__global__ __launch_bounds__(256) void v_transpose_kernel(KVLayout cache, const bf16_t* __restrict__ qkv, int nh, int vhead0,
                                                          const int* __restrict__ cu, const int* __restrict__ slot_of_clip,
                                                          bf16_t* __restrict__ vt, int vt_stride) {
    __shared__ bf16_t tile[64][HD + 2];
    const int clip = blockIdx.z, kvh = blockIdx.y, p0 = blockIdx.x * 64;
    const int T = cu[clip + 1] - cu[clip];
    if (p0 >= T) return;
    const int sl = slot_of_clip[clip], tid = threadIdx.x;
    const bf16_t* src = qkv + ((long)(cu[clip] + p0) * nh + vhead0 + kvh) * HD;
    constexpr int CH = HD / 8;
    for (int i = tid; i < 64 * CH; i += 256) {
        const int r = i / CH, ch = i - r * CH;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (p0 + r < T) u = *reinterpret_cast<const uint4*>(src + (long)r * nh * HD + ch * 8);
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&u);
#pragma unroll
        for (int q = 0; q < 8; ++q) tile[r][ch * 8 + q] = e[q];
    }
    __syncthreads();
    bf16_t* dst = vt ? vt + ((long)sl * cache.kv_heads + kvh) * HD * vt_stride + p0 : nullptr;
    for (int i = tid; dst && i < HD * 8; i += 256) {     // 8 chunks of 8 positions per d row
        const int d = i >> 3, ch = i & 7;
        uint4 o;
        bf16_t* oe = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
        for (int q = 0; q < 8; ++q) oe[q] = tile[ch * 8 + q][d];
        *reinterpret_cast<uint4*>(dst + (long)d * vt_stride + ch * 8) = o;
    }
    if (!cache.vf) return;
    // the decode sweep's fragment-major image (vfrag_index): 2 chunks of 32 keys, one 16-byte fragment per thread
    constexpr int DT = HD / 16;
    bf16_t* vf = cache.vf + cache.off(sl, kvh, 0) + (long)(p0 / 32) * DT * 512;
    for (int i = tid; i < 2 * DT * 64; i += 256) {
        const int kbl = i / (DT * 64), rem = i - kbl * DT * 64, dt = rem >> 6, ln = rem & 63, d = dt * 16 + (ln & 15), g = ln >> 4;
        uint4 o;
        bf16_t* oe = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
        for (int e = 0; e < 8; ++e) oe[e] = tile[kbl * 32 + (e >> 2) * 16 + g * 4 + (e & 3)][d];
        *reinterpret_cast<uint4*>(vf + (long)i * 8) = o;
    }
}

void qk_norm_rope_launch(const bf16_t* qkv, const int* slot, const int* pos, int n_pos, int heads, int kv_heads, int hd,
                         const bf16_t* qn_w, const bf16_t* kn_w, float eps, const float* rope_cos,
                         const float* rope_sin, bf16_t* qr, KVLayout cache, bf16_t* vt, int vt_stride, const int* cu,
                         const int* slot_of_clip, int n_clips, int max_len, hipStream_t s, bool v_only) {
    if (n_pos <= 0) return;
    const int nh = heads + 2 * kv_heads;
    if (v_only) {            // q and k were handled in the projection's epilogue (EpiQkHeads): only the V images are left
        if (hd != 128) throw std::invalid_argument("qk_norm_rope: v_only needs head_dim 128");
        bf16_t* vt_o = (cache.vf == nullptr || tuning().pa_vfrag == 0) ? vt : nullptr;
        if (!vt_o && !cache.vf) throw std::invalid_argument("qk_norm_rope: no V image to write");
        hipLaunchKernelGGL(v_transpose_kernel<128>, dim3(cdiv(max_len, 64), kv_heads, n_clips), dim3(256), 0, s, cache, qkv, nh, heads + kv_heads, cu,
                           slot_of_clip, vt_o, vt_stride);
        return;
    }
    const int wide = tuning().qknr_wide;      // A/B knob
    // the prompt attention reads V^T only without the fragment image (forced aligner) or under the pa_vfrag = 0 A/B
    bf16_t* vt_out = (cache.vf == nullptr || tuning().pa_vfrag == 0) ? vt : nullptr;
    if (!vt_out && !cache.vf) throw std::invalid_argument("qk_norm_rope: no V image to write");
    const dim3 tgrid(cdiv(max_len, 64), kv_heads, n_clips);
    if (hd == 128 && (heads + kv_heads) % 8 == 0 && wide) {
        long waves = (long)n_pos * ((heads + kv_heads) / 8);
        hipLaunchKernelGGL(qk_norm_rope_wide_kernel<128>, dim3(cdiv(waves, 4)), dim3(256), 0, s, qkv, slot, pos, n_pos, heads,
                           kv_heads, qn_w, kn_w, eps, rope_cos, rope_sin, qr, cache);
        hipLaunchKernelGGL(v_transpose_kernel<128>, tgrid, dim3(256), 0, s, cache, qkv, nh, heads + kv_heads, cu, slot_of_clip, vt_out, vt_stride);
    } else if (hd == 128 && nh % 2 == 0) {
        long waves = (long)n_pos * (nh / 2);
        hipLaunchKernelGGL(qk_norm_rope_kernel<128>, dim3(cdiv(waves, 4)), dim3(256), 0, s, qkv, slot, pos, n_pos, heads,
                           kv_heads, qn_w, kn_w, eps, rope_cos, rope_sin, qr, cache);
        hipLaunchKernelGGL(v_transpose_kernel<128>, tgrid, dim3(256), 0, s, cache, qkv, nh, heads + kv_heads, cu, slot_of_clip, vt_out, vt_stride);
    } else if (hd == 32 && nh % 8 == 0) {
        long waves = (long)n_pos * (nh / 8);
        hipLaunchKernelGGL(qk_norm_rope_kernel<32>, dim3(cdiv(waves, 4)), dim3(256), 0, s, qkv, slot, pos, n_pos, heads,
                           kv_heads, qn_w, kn_w, eps, rope_cos, rope_sin, qr, cache);
        hipLaunchKernelGGL(v_transpose_kernel<32>, tgrid, dim3(256), 0, s, cache, qkv, nh, heads + kv_heads, cu, slot_of_clip, vt_out, vt_stride);
    } else {
        throw std::invalid_argument("qk_norm_rope: unsupported (head_dim, head count)");
    }
}

bool qk_norm_rope_fusable(int heads, int kv_heads, int hd) {
    return hd == 128 && (heads + kv_heads) % 8 == 0 && tuning().qknr_wide != 0 && tuning().pp_fuse_qk != 0;
}

// ------------------------------------------------------------------------------------------------
// Causal flash attention for the prompt pass.  Workgroup = 128 query rows of one (clip, head):
// 4 waves x 32 rows (2 MFMA row tiles).  Key tiles of 64: K rows and V^T rows are staged in LDS with
// an XOR chunk swizzle, S = Q K^T (16x16x32 MFMA), online softmax on the accumulator layout (a row
// lives on 16 lanes), P -> bf16 through a wave-private LDS image -> A operand of P V.
// ------------------------------------------------------------------------------------------------
template <int HD, int MT>
__global__ __launch_bounds__(256) void prefill_attention_kernel(const bf16_t* __restrict__ qr, KVLayout cache,
                                                                const bf16_t* __restrict__ vt, int vt_stride,
                                                                const int* __restrict__ cu,
                                                                const int* __restrict__ slot_of_clip, int heads,
                                                                bf16_t* __restrict__ out, float scale) {
    constexpr int KT = 64;                 // keys per tile
    constexpr int KCH = HD / 8;            // 16-byte chunks per K row
    constexpr int KS = HD / 32;            // k-steps of Q K^T
    constexpr int DT = HD / 16;            // output d tiles
    constexpr int PLD = KT + 8;
    constexpr int NKL = KT * KCH / 256;    // K-tile chunks staged per thread
    constexpr int NVL = HD * (KT / 8) / 256;   // V^T-tile chunks staged per thread
    static_assert(KT * KCH % 256 == 0 && HD * (KT / 8) % 256 == 0, "tile staging must divide evenly");
    // one LDS block: K tile | V^T tile (re-used as the output staging image at the end) | P images
    __shared__ __attribute__((aligned(16))) bf16_t s_kv[2 * KT * HD];
    __shared__ __attribute__((aligned(16))) bf16_t s_p[4][16 * MT][PLD];
    bf16_t* s_k = s_kv;
    bf16_t* s_v = s_kv + KT * HD;
    constexpr int QW = 16 * MT, QB = 4 * QW;   // query rows per wave / per workgroup
    const int clip = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB;
    const int row0 = cu[clip], T = cu[clip + 1] - row0;
    if (q0 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fc = lane >> 4;
    const int sl = slot_of_clip[clip];
    const int kvh = h / (heads / cache.kv_heads);
    const bf16_t* kbase = cache.k + cache.off(sl, kvh, 0);
    const bf16_t* vbase = vt + ((long)sl * cache.kv_heads + kvh) * HD * vt_stride;
    const int qw = q0 + wave * QW;         // first query row of this wave

    // tile staging global -> registers -> LDS inside the tile step (a register prefetch of tile kt+1 across the
    // MFMAs costs 32 more VGPRs, which drops this kernel from 2 waves/SIMD to 1 with spills: measured slower)
    auto stage = [&](int k0) {
        uint4 kreg[NKL], vreg[NVL];
#pragma unroll
        for (int i = 0; i < NKL; ++i) {
            const int idx = tid + i * 256, key = idx / KCH, ch = idx - key * KCH;
            const int kc = k0 + key < T ? k0 + key : T - 1;              // clamped row, zeroed below
            kreg[i] = *reinterpret_cast<const uint4*>(kbase + (long)kc * HD + ch * 8);
            if (k0 + key >= T) kreg[i] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NVL; ++i) {
            const int idx = tid + i * 256, d = idx / (KT / 8), ch = idx - d * (KT / 8);
            vreg[i] = *reinterpret_cast<const uint4*>(vbase + (long)d * vt_stride + k0 + ch * 8);   // V^T is zero past T
        }
#pragma unroll
        for (int i = 0; i < NKL; ++i) {
            const int idx = tid + i * 256, key = idx / KCH, ch = idx - key * KCH;
            *reinterpret_cast<uint4*>(&s_k[key * HD + ((ch ^ (key & (KCH - 1))) << 3)]) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < NVL; ++i) {
            const int idx = tid + i * 256, d = idx / (KT / 8), ch = idx - d * (KT / 8);
            *reinterpret_cast<uint4*>(&s_v[d * KT + ((ch ^ (d & 7)) << 3)]) = vreg[i];
        }
    };

    mfma_bf16x8 qf[MT][KS];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int r = qw + mi * 16 + fr;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int rc = r < T ? r : T - 1;
            uint4 u = *reinterpret_cast<const uint4*>(qr + ((long)(row0 + rc) * heads + h) * HD + s * 32 + fc * 8);
            if (r >= T) u = make_uint4(0, 0, 0, 0);
            qf[mi][s] = __builtin_bit_cast(mfma_bf16x8, u);
        }
    }
    f32x4 o[MT][DT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int d = 0; d < DT; ++d) o[mi][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[MT][4], l_run[MT][4];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j) { m_run[mi][j] = -INFINITY; l_run[mi][j] = 0.0f; }

    const int q_hi = min(q0 + QB, T);                 // causal: keys < q_hi
    const int n_tiles = (q_hi + KT - 1) / KT;
    for (int kt = 0; kt < n_tiles; ++kt) {
        const int k0 = kt * KT;
        __syncthreads();                               // previous tile's LDS reads are done
        stage(k0);
        __syncthreads();
        if (k0 <= qw + QW - 1 && qw < T) {                 // this wave has unmasked keys in the tile
            f32x4 sc[MT][4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                mfma_bf16x8 kf[KS];
                const int key = nt * 16 + fr;
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    kf[s] = *reinterpret_cast<const mfma_bf16x8*>(&s_k[key * HD + (((s * 4 + fc) ^ (key & (KCH - 1))) << 3)]);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[mi][s], kf[s], acc, 0, 0, 0);
                    sc[mi][nt] = acc;
                }
            }
            // online softmax; lane holds rows mi*16 + fc*4 + j, key column nt*16 + fr
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                float alpha[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int qpos = qw + mi * 16 + fc * 4 + j;
                    float mx = -INFINITY;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int key = k0 + nt * 16 + fr;
                        float v = (key <= qpos && key < T) ? sc[mi][nt][j] * scale : -INFINITY;
                        sc[mi][nt][j] = v;
                        mx = fmaxf(mx, v);
                    }
#pragma unroll
                    for (int ofs = 1; ofs < 16; ofs <<= 1) mx = fmaxf(mx, __shfl_xor(mx, ofs, 64));
                    const float m_new = fmaxf(m_run[mi][j], mx);
                    // rows with no visible key yet keep m = -inf: use 0 as the reference to avoid inf - inf
                    const float m_ref = m_new == -INFINITY ? 0.0f : m_new;
                    alpha[j] = expf(m_run[mi][j] - m_ref);
                    float rs = 0.0f;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        bf16_t pb = f32_to_bf16(expf(sc[mi][nt][j] - m_ref));
                        s_p[wave][mi * 16 + fc * 4 + j][nt * 16 + fr] = pb;
                        rs += bf16_to_f32(pb);
                    }
#pragma unroll
                    for (int ofs = 1; ofs < 16; ofs <<= 1) rs += __shfl_xor(rs, ofs, 64);
                    l_run[mi][j] = l_run[mi][j] * alpha[j] + rs;
                    m_run[mi][j] = m_new;
                }
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[mi][d][j] *= alpha[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ks = 0; ks < KT / 32; ++ks) {
                mfma_bf16x8 pa[MT];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    pa[mi] = *reinterpret_cast<const mfma_bf16x8*>(&s_p[wave][mi * 16 + fr][ks * 32 + fc * 8]);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const int dr = d * 16 + fr;
                    mfma_bf16x8 vf = *reinterpret_cast<const mfma_bf16x8*>(&s_v[dr * KT + (((ks * 4 + fc) ^ (dr & 7)) << 3)]);
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) o[mi][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[mi], vf, o[mi][d], 0, 0, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- output: normalise, stage the wave's 32 x HD tile in LDS (K/V images are dead), 16-byte row stores ----
    __syncthreads();
    bf16_t* s_o = s_kv + wave * (QW * HD);                      // 4 waves x QW x HD bf16 <= 2*KT*HD elements
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float invl = 1.0f / l_run[mi][j];
            const int r = mi * 16 + fc * 4 + j;
#pragma unroll
            for (int d = 0; d < DT; ++d) s_o[r * HD + d * 16 + fr] = f32_to_bf16(o[mi][d][j] * invl);
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < QW * KCH; i += 64) {
        const int r = i / KCH, ch = i - r * KCH;
        if (qw + r < T)
            *reinterpret_cast<uint4*>(out + ((long)(row0 + qw + r) * heads + h) * HD + ch * 8) =
                *reinterpret_cast<const uint4*>(&s_o[r * HD + ch * 8]);
    }
}

// ------------------------------------------------------------------------------------------------
// Prompt attention, second form.  Workgroup = 64 query rows of one (clip, KV head) -- BOTH query heads of the GQA
// pair share every staged K / V^T tile; 4 waves x 16 rows.  Key tiles of 64 are double-buffered in LDS by
// direct-to-LDS loads (global_load_lds, lane-linear image, XOR chunk swizzle on the source address), so tile t+1
// streams in under the MFMAs of tile t and no staging registers are live.  As in the decode kernel the scores are
// produced TRANSPOSED, S^T = K Q^T: a lane then owns ONE query row (lane & 15) and 16 of the tile's keys, so the row
// maximum needs 2 cross-lane steps instead of 8, the row sum none until the end, and the accumulator registers ARE
// the B operand of O^T = V^T P^T -- P never goes through LDS, and O^T has the query on the same lane as its
// softmax statistics (rescale = one multiply, no shuffles).  Rounding points are those of the first form
// (flash_prefill_attention in oracle/decoder.py): per 64-key tile, P rounded to bf16, the row sum over rounded P.
// ------------------------------------------------------------------------------------------------
// amdgpu_waves_per_eu(2, 2): without it hipcc spreads the accumulators over 202 VGPRs + 78 AGPRs = 280 registers, which
// leaves ONE wave per SIMD (hipOccupancyMaxActiveBlocksPerMultiprocessor = 1; SQ_WAVE_CYCLES showed 0.8 waves per SIMD);
// capped at 256 it needs 204 VGPRs, no spills, two workgroups per CU.
//
// VFRAG (round 3): V comes from the decode sweep's fragment-major image (cache.vf, vfrag_index: per 32 keys and 16 head dims one 1 KiB
// block holding, lane by lane, exactly the 16-byte A fragment of O^T = V^T P^T -- keys {4g .. 4g+3, 16+4g .. 16+4g+3} of row d).  A 64-key
// tile is 16 KiB contiguous in HBM (linear direct-to-LDS copy), a fragment is ONE lane-linear ds_read_b128 at a constant offset: no
// address arithmetic, no bank conflicts, none of the register moves the two-8-byte-reads form needed (hipcc paired the reads of
// neighbouring d tiles into ds_read2st64 and moved the halves together: 48 v_mov per tile).  The transposed V^T image stays for
// engines without the fragment image (forced aligner).  Staging and read addresses are per-lane constants + a wave-uniform base.
// STAMPS (qasr_kernel_probe 5 with the pa_stamps diagnostic knob, `make DIAG=1`): per wave, 100 MHz wall-clock sums of the loop's phases.
template <int HD, bool VFRAG, bool STAMPS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void prefill_attention2_kernel(const bf16_t* __restrict__ qr, KVLayout cache,
                                                                 const bf16_t* __restrict__ vt, int vt_stride,
                                                                 const int* __restrict__ cu,
                                                                 const int* __restrict__ slot_of_clip, int heads,
                                                                 bf16_t* __restrict__ out, float scale, int heavy_first,
                                                                 unsigned long long* __restrict__ dbg = nullptr) {
    [[maybe_unused]] unsigned long long st_t0 = 0, st_prev = 0, st_sum[6] = {0, 0, 0, 0, 0, 0};
    // phase p ends here: the time since the previous mark goes to st_sum[p]
#define PA_MARK(p) do { if constexpr (STAMPS) { const unsigned long long t_ = wall_clock64(); st_sum[p] += t_ - st_prev; st_prev = t_; } } while (0)
    if constexpr (STAMPS) { st_t0 = wall_clock64(); st_prev = st_t0; }
    constexpr int KT = 64, KS = HD / 32, DT = HD / 16, KCH = HD / 8, REP = 2;
    constexpr int TILE_BYTES = KT * HD * 2;                       // K tile and V^T tile have the same size
    constexpr int K_RPI = 64 / KCH, K_IPW = KT / K_RPI / 4;       // rows per wave instruction, instructions per wave
    constexpr int V_IPW = HD / 8 / 4;                             // V^T rows are 128 B: 8 rows per instruction
    static_assert(K_IPW >= 1 && V_IPW >= 1, "tile staging geometry");
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    __shared__ __attribute__((aligned(16))) char smem[2][2][TILE_BYTES];   // [buffer][K | V^T]
    // causal: query tile i sweeps i + 1 key tiles.  heavy_first: grid (kv head, clip, query tile) with the LAST query tile dispatched
    // first -- the long workgroups start at once and the short ones fill the tail (the other order interleaves 1..7-tile workgroups and
    // ends on whichever long one started last); a kv head then also stays on one XCD (workgroup id % 8), with its K / V^T tiles in that L2
    const int clip = heavy_first ? blockIdx.y : blockIdx.z, kvh = heavy_first ? blockIdx.x : blockIdx.y;
    const int q0 = (heavy_first ? gridDim.z - 1 - blockIdx.z : blockIdx.x) * 64;
    const int row0 = cu[clip], T = cu[clip + 1] - row0;
    if (q0 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
    const int sl = slot_of_clip[clip];
    const bf16_t* kbase = cache.k + cache.off(sl, kvh, 0);
    const bf16_t* vbase = vt + ((long)sl * cache.kv_heads + kvh) * HD * vt_stride;
    const int qw = q0 + wave * 16, qpos = qw + fr;

    // per-lane constants of the tile copies (bytes from the tile's wave-uniform base)
    unsigned koff[K_IPW], voff[V_IPW];
#pragma unroll
    for (int i = 0; i < K_IPW; ++i) {
        const int r = (wave * K_IPW + i) * K_RPI + lane / KCH, c = lane % KCH;
        koff[i] = (unsigned)(r * HD + ((c ^ (r & (KCH - 1))) << 3)) * 2u;
    }
#pragma unroll
    for (int i = 0; i < V_IPW; ++i) {
        const int inst = wave * V_IPW + i;
        if (VFRAG) voff[i] = (unsigned)(inst * 1024 + lane * 16);
        else {
            const int d = inst * 8 + (lane >> 3), c = lane & 7;
            voff[i] = (unsigned)(d * vt_stride + ((c ^ (d & 7)) << 3)) * 2u;           // V^T is zero past the prompt
        }
    }
    const char* vsrc = VFRAG ? reinterpret_cast<const char*>(cache.vf + cache.off(sl, kvh, 0)) : reinterpret_cast<const char*>(vbase);
    auto stage_k = [&](int buf, int k0) {
        const char* kb = reinterpret_cast<const char*>(kbase) + (long)k0 * (HD * 2);
        if (k0 + KT <= cache.max_ctx) {
#pragma unroll
            for (int i = 0; i < K_IPW; ++i)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(kb + koff[i]), (lds_ptr_t)&smem[buf][0][(wave * K_IPW + i) * 1024], 16, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < K_IPW; ++i) {
                const int inst = wave * K_IPW + i;
                const int r = inst * K_RPI + lane / KCH, c = lane % KCH;
                int key = k0 + r;
                key = key < cache.max_ctx ? key : cache.max_ctx - 1;      // rows past the prompt are masked, not read as data
                const bf16_t* src = kbase + (long)key * HD + ((c ^ (r & (KCH - 1))) << 3);
                __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&smem[buf][0][inst * 1024], 16, 0, 0);
            }
        }
    };
    auto stage_v = [&](int buf, int k0) {
        // fragment image: 32 keys = DT KiB; transposed image: 2 bytes per key along a row
        const char* vb = vsrc + (VFRAG ? (long)(k0 / 32) * (DT * 1024) : (long)k0 * 2);
#pragma unroll
        for (int i = 0; i < V_IPW; ++i)
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(vb + voff[i]), (lds_ptr_t)&smem[buf][1][(wave * V_IPW + i) * 1024], 16, 0, 0);
    };
    auto stage = [&](int buf, int k0) { stage_k(buf, k0); stage_v(buf, k0); };

    const int q_hi = min(q0 + 64, T);                     // causal: keys < q_hi
    const int n_tiles = (q_hi + KT - 1) / KT;
    stage(0, 0);
    // query fragments: B operand of S^T (column = query row fr, k = head dims), both heads of the pair
    mfma_bf16x8 qf[REP][KS];
#pragma unroll
    for (int mi = 0; mi < REP; ++mi) {
        const int h = kvh * REP + mi;
        const int rc = qpos < T ? qpos : T - 1;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint4 u = *reinterpret_cast<const uint4*>(qr + ((long)(row0 + rc) * heads + h) * HD + s * 32 + g * 8);
            qf[mi][s] = __builtin_bit_cast(mfma_bf16x8, u);
        }
    }
    f32x4 o[REP][DT];                                     // O^T: rows d = dt*16 + g*4 + j, column = query row fr
#pragma unroll
    for (int mi = 0; mi < REP; ++mi)
#pragma unroll
        for (int d = 0; d < DT; ++d) o[mi][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[REP], l_run[REP];
#pragma unroll
    for (int mi = 0; mi < REP; ++mi) { m_run[mi] = -INFINITY; l_run[mi] = 0.0f; }
    // LDS read offsets inside a tile: K fragment (key = nb * 16 + fr, k-step s) = kaddr[s] + nb * 16 rows; V fragment at a lane-linear
    // offset (VFRAG) or the two 8-byte halves of the swizzled V^T row
    unsigned kaddr[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) kaddr[s] = (unsigned)(fr * (HD * 2) + (((s * 4 + g) ^ (fr & (KCH - 1))) << 4));
    const bf16x2_native ones2 = __builtin_bit_cast(bf16x2_native, 0x3f803f80u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // LDS-DMA completion is tracked by vmcnt only (also covers qf)
    __syncthreads();
    PA_MARK(0);                                         // prologue: cu / slot reads, first tile's copy, query fragments

    for (int kt = 0; kt < n_tiles; ++kt) {
        const int cur = kt & 1, k0 = kt * KT;
        // the next tile streams in under this tile's MFMAs.  (Requesting its V half after the S^T block instead -- so that the eight waves of
        // a CU do not push 64 KB into the 64 B/clk L1 fill path in one burst -- measured +-0: profiles/r03_stamps_prompt_attention.txt)
        if (kt + 1 < n_tiles) stage(cur ^ 1, k0 + KT);
        PA_MARK(1);                                     // copy requests issued
        if (k0 <= qw + 15 && qw < T) {                    // this wave has unmasked keys in the tile (wave-uniform)
            const char* s_k = smem[cur][0];
            const char* s_v = smem[cur][1];
            f32x4 sc[REP][4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                mfma_bf16x8 kf[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    kf[s] = *reinterpret_cast<const mfma_bf16x8*>(s_k + nb * (16 * HD * 2) + kaddr[s]);
#pragma unroll
                for (int mi = 0; mi < REP; ++mi) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[s], qf[mi][s], acc, 0, 0, 0);
                    sc[mi][nb] = acc;
                }
            }
            if constexpr (STAMPS) { asm volatile("s_nop 0" ::"v"(sc[REP - 1][3][3])); }    // the last score is in its register
            PA_MARK(2);                                 // S^T
            unsigned pk[REP][2][4];
            // only the tile on the diagonal (or the prompt's last tile) needs the per-key mask (wave-uniform)
            const bool full = k0 + KT - 1 <= qw && k0 + KT <= T;
            const float c2 = scale * 1.44269504088896341f;                 // exp(s * scale - m * scale) = 2^((s - m) * c2)
#pragma unroll
            for (int mi = 0; mi < REP; ++mi) {
                float mx = -INFINITY;
                if (!full) {
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int key = k0 + nb * 16 + g * 4 + j;
                            if (!(key <= qpos && key < T)) sc[mi][nb][j] = -INFINITY;     // select: stale rows may be NaN
                        }
                }
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mx = fmaxf(mx, sc[mi][nb][j]);
                {   // the row's other keys live on the lanes fr + 16 g': v_permlane16_swap / v_permlane32_swap instead of two LDS round trips
                    const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                    mx = fmaxf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
                    const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                    mx = fmaxf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
                }
                const float m_new = fmaxf(m_run[mi], mx);                  // raw score units (scale > 0 keeps the order)
                const float m_ref = m_new == -INFINITY ? 0.0f : m_new;     // rows past the prompt keep m = -inf
                const float alpha = __builtin_amdgcn_exp2f((m_run[mi] - m_ref) * c2);
                const float mc = -m_ref * c2;
                float rs = 0.0f;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const unsigned pw = pack_bf16x2(__builtin_amdgcn_exp2f(fmaf(sc[mi][nb][j], c2, mc)),
                                                        __builtin_amdgcn_exp2f(fmaf(sc[mi][nb][j + 1], c2, mc)));
                        rs = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_native, pw), ones2, rs, false);   // sum of the ROUNDED P
                        // k-slot order of the P V^T product: slots 0-3 <- keys 4g+j of the even 16-key block, 4-7 <- the odd one
                        pk[mi][nb >> 1][(nb & 1) * 2 + j / 2] = pw;
                    }
                l_run[mi] = l_run[mi] * alpha + rs;
                m_run[mi] = m_new;
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[mi][d][j] *= alpha;
            }
            PA_MARK(3);                                 // softmax, rescale
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                mfma_bf16x8 pb[REP];
#pragma unroll
                for (int mi = 0; mi < REP; ++mi)
                    pb[mi] = __builtin_bit_cast(mfma_bf16x8, make_uint4(pk[mi][p][0], pk[mi][p][1], pk[mi][p][2], pk[mi][p][3]));
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    mfma_bf16x8 vf;
                    if constexpr (VFRAG) {
                        vf = *reinterpret_cast<const mfma_bf16x8*>(s_v + (p * DT + d) * 1024 + lane * 16);
                    } else {
                        const int dr = d * 16 + fr;
                        // A operand rows = d; k-slots 8g..8g+7 <- keys {32p + 4g + j, 32p + 16 + 4g + j}: two 8-byte reads.  Native vector
                        // types: a read through HIP_vector_type (uint2) carries no alias info and hipcc puts s_waitcnt vmcnt(0) in front of
                        // it while the next tile's global_load_lds are in flight -- the prefetch would end here, not at the barrier
                        typedef unsigned __attribute__((ext_vector_type(2))) u32x2_n;
                        typedef unsigned __attribute__((ext_vector_type(4))) u32x4_n;
                        const char* vrow = s_v + dr * 128 + (g & 1) * 8;
                        const u32x2_n lo = *reinterpret_cast<const u32x2_n*>(vrow + (((4 * p + (g >> 1)) ^ (dr & 7)) << 4));
                        const u32x2_n hi = *reinterpret_cast<const u32x2_n*>(vrow + (((4 * p + 2 + (g >> 1)) ^ (dr & 7)) << 4));
                        vf = __builtin_bit_cast(mfma_bf16x8, u32x4_n{lo.x, lo.y, hi.x, hi.y});
                    }
#pragma unroll
                    for (int mi = 0; mi < REP; ++mi) o[mi][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[mi], o[mi][d], 0, 0, 0);
                }
            }
        }
        if constexpr (STAMPS) { asm volatile("s_nop 0" ::"v"(o[REP - 1][DT - 1][3])); }
        PA_MARK(4);                                         // P V^T issued and its last accumulator written
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // tile kt+1 has landed (this wave's part) ...
        __syncthreads();                                // ... and every wave's part after the barrier
        PA_MARK(5);                                         // wait for the copies + barrier
    }
    if constexpr (STAMPS) {
        if (dbg && lane == 0) {
            unsigned long long* d = dbg + ((((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave) * 8;
            d[0] = st_t0; d[1] = wall_clock64();
#pragma unroll
            for (int i = 0; i < 6; ++i) d[2 + i] = st_sum[i];
        }
    }
#undef PA_MARK
    // ---- output: lane = query row fr, head dims dt*16 + g*4 .. +3 -> 8-byte stores ------------------------------
    if (qpos < T) {
#pragma unroll
        for (int mi = 0; mi < REP; ++mi) {
            float l = l_run[mi];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            const float invl = 1.0f / l;
            bf16_t* dst = out + ((long)(row0 + qpos) * heads + kvh * REP + mi) * HD + g * 4;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const uint2 v = make_uint2(pack_bf16x2(o[mi][d][0] * invl, o[mi][d][1] * invl), pack_bf16x2(o[mi][d][2] * invl, o[mi][d][3] * invl));
                *reinterpret_cast<uint2*>(dst + d * 16) = v;
            }
        }
    }
}

void prefill_attention_launch(const bf16_t* qr, KVLayout cache, const bf16_t* vt, int vt_stride, const int* cu,
                              const int* slot_of_clip, int n_clips, int max_len, int heads, bf16_t* out,
                              hipStream_t s, unsigned long long* dbg) {
    if (n_clips <= 0 || max_len <= 0) return;
    const int mt = tuning().pa_mt;          // A/B knob: row tiles per wave
    const int form = tuning().pa_form;      // A/B knob: 2 = transposed-score form
    const float scale = 1.0f / sqrtf((float)cache.hd);
    if (form >= 2 && heads == 2 * cache.kv_heads && (cache.hd == 128 || cache.hd == 32)) {
        const int hf = tuning().pa_order != 0 && cache.kv_heads == 8 ? 1 : 0;     // A/B knob; the XCD argument holds for 8 kv heads
        const dim3 grid = hf ? dim3(cache.kv_heads, n_clips, cdiv(max_len, 64)) : dim3(cdiv(max_len, 64), cache.kv_heads, n_clips);
        const bool vfrag = cache.vf != nullptr && tuning().pa_vfrag != 0;      // A/B knob; engines without the fragment image: V^T
        if (dbg) {     // diagnostic build only (qasr_kernel_probe 5): [workgroup][wave][8] stamps
            if (cache.hd != 128 || !vfrag) throw std::invalid_argument("prompt attention stamps: head_dim 128 with the fragment image only");
            hipLaunchKernelGGL((prefill_attention2_kernel<128, true, true>), grid, dim3(256), 0, s, qr, cache, vt, vt_stride, cu, slot_of_clip, heads, out, scale, hf, dbg);
            return;
        }
        if (cache.hd == 128) {
            if (vfrag) hipLaunchKernelGGL((prefill_attention2_kernel<128, true>), grid, dim3(256), 0, s, qr, cache, vt, vt_stride, cu, slot_of_clip, heads, out, scale, hf);
            else hipLaunchKernelGGL((prefill_attention2_kernel<128, false>), grid, dim3(256), 0, s, qr, cache, vt, vt_stride, cu, slot_of_clip, heads, out, scale, hf);
        } else {
            if (vfrag) hipLaunchKernelGGL((prefill_attention2_kernel<32, true>), grid, dim3(256), 0, s, qr, cache, vt, vt_stride, cu, slot_of_clip, heads, out, scale, hf);
            else hipLaunchKernelGGL((prefill_attention2_kernel<32, false>), grid, dim3(256), 0, s, qr, cache, vt, vt_stride, cu, slot_of_clip, heads, out, scale, hf);
        }
        return;
    }
    if (cache.hd == 128 && mt == 2)
        hipLaunchKernelGGL((prefill_attention_kernel<128, 2>), dim3(cdiv(max_len, 128), heads, n_clips), dim3(256), 0, s,
                           qr, cache, vt, vt_stride, cu, slot_of_clip, heads, out, scale);
    else if (cache.hd == 128)
        hipLaunchKernelGGL((prefill_attention_kernel<128, 1>), dim3(cdiv(max_len, 64), heads, n_clips), dim3(256), 0, s,
                           qr, cache, vt, vt_stride, cu, slot_of_clip, heads, out, scale);
    else if (cache.hd == 32)
        hipLaunchKernelGGL((prefill_attention_kernel<32, 2>), dim3(cdiv(max_len, 128), heads, n_clips), dim3(256), 0, s,
                           qr, cache, vt, vt_stride, cu, slot_of_clip, heads, out, scale);
    else
        throw std::invalid_argument("prefill attention: head_dim must be 32 or 128");
}

}  // namespace qasr

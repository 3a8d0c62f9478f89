// dec_kernels.hip -- text-decoder kernels (see dec_kernels.h).
#include "dec_kernels.h"
#include "dec_epilogue.h"
#include "dec_quant.h"
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
__device__ __forceinline__ void norm_rope_pair(float x1, float x2, float w1, float w2, float inv, float c, float sn,
                                               float& o1, float& o2) {
    // bf16(w * bf16(x * inv))  then  bf16(x1*cos - x2*sin), bf16(x1*sin + x2*cos)
    float y1 = bf16_round(w1 * bf16_round(x1 * inv));
    float y2 = bf16_round(w2 * bf16_round(x2 * inv));
    o1 = bf16_round(y1 * c - y2 * sn);
    o2 = bf16_round(y1 * sn + y2 * c);
}

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
    if (h >= heads + kv_heads) {                        // V: plain copy (the transposed image is built separately)
        bf16_t* dv = cache.v + cache.off(sl, h - heads - kv_heads, ps);
        *reinterpret_cast<unsigned*>(dv + 2 * j) = a;
        *reinterpret_cast<unsigned*>(dv + HALF + 2 * j) = bb;
        return;
    }
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
    const int groups = nh / HPW;                        // head groups per position (host checks nh % HPW == 0)
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wid >= (long)n_pos * groups) return;
    const int p = (int)(wid / groups), h = (int)(wid - (long)p * groups) * HPW + lane / LPH;
    const int j = lane % LPH;
    const bf16_t* src = qkv + (long)p * nh * HD + (long)h * HD;
    const int sl = slot[p], ps = pos[p];
    const uint4 a = *reinterpret_cast<const uint4*>(src + 8 * j);
    const uint4 b = *reinterpret_cast<const uint4*>(src + HALF + 8 * j);
    if (h >= heads + kv_heads) {                        // V: plain copy (the transposed image is built separately)
        bf16_t* dv = cache.v + cache.off(sl, h - heads - kv_heads, ps);
        *reinterpret_cast<uint4*>(dv + 8 * j) = a;
        *reinterpret_cast<uint4*>(dv + HALF + 8 * j) = b;
        return;
    }
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

// V^T image for the prompt pass: cache.v rows [pos][HD] -> vt[slot][kvh][d][pos], 64 positions per workgroup,
// transposed through LDS so both sides move 128-byte rows.
template <int HD>
__global__ __launch_bounds__(256) void v_transpose_kernel(KVLayout cache, const int* __restrict__ cu,
                                                          const int* __restrict__ slot_of_clip,
                                                          bf16_t* __restrict__ vt, int vt_stride) {
    __shared__ bf16_t tile[64][HD + 2];
    const int clip = blockIdx.z, kvh = blockIdx.y, p0 = blockIdx.x * 64;
    const int T = cu[clip + 1] - cu[clip];
    if (p0 >= T) return;
    const int sl = slot_of_clip[clip], tid = threadIdx.x;
    const bf16_t* src = cache.v + cache.off(sl, kvh, p0);
    constexpr int CH = HD / 8;
    for (int i = tid; i < 64 * CH; i += 256) {
        const int r = i / CH, ch = i - r * CH;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (p0 + r < T) u = *reinterpret_cast<const uint4*>(src + (long)r * HD + ch * 8);
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&u);
#pragma unroll
        for (int q = 0; q < 8; ++q) tile[r][ch * 8 + q] = e[q];
    }
    __syncthreads();
    bf16_t* dst = vt + ((long)sl * cache.kv_heads + kvh) * HD * vt_stride + p0;
    for (int i = tid; i < HD * 8; i += 256) {            // 8 chunks of 8 positions per d row
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
                         const int* slot_of_clip, int n_clips, int max_len, hipStream_t s) {
    if (n_pos <= 0) return;
    const int nh = heads + 2 * kv_heads;
    const int wide = tuning().qknr_wide;      // A/B knob
    if (hd == 128 && nh % 8 == 0 && wide) {
        long waves = (long)n_pos * (nh / 8);
        hipLaunchKernelGGL(qk_norm_rope_wide_kernel<128>, dim3(cdiv(waves, 4)), dim3(256), 0, s, qkv, slot, pos, n_pos, heads,
                           kv_heads, qn_w, kn_w, eps, rope_cos, rope_sin, qr, cache);
        if (vt) hipLaunchKernelGGL(v_transpose_kernel<128>, dim3(cdiv(max_len, 64), kv_heads, n_clips), dim3(256), 0, s,
                                   cache, cu, slot_of_clip, vt, vt_stride);
    } else if (hd == 128 && nh % 2 == 0) {
        long waves = (long)n_pos * (nh / 2);
        hipLaunchKernelGGL(qk_norm_rope_kernel<128>, dim3(cdiv(waves, 4)), dim3(256), 0, s, qkv, slot, pos, n_pos, heads,
                           kv_heads, qn_w, kn_w, eps, rope_cos, rope_sin, qr, cache);
        if (vt) hipLaunchKernelGGL(v_transpose_kernel<128>, dim3(cdiv(max_len, 64), kv_heads, n_clips), dim3(256), 0, s,
                                   cache, cu, slot_of_clip, vt, vt_stride);
    } else if (hd == 32 && nh % 8 == 0) {
        long waves = (long)n_pos * (nh / 8);
        hipLaunchKernelGGL(qk_norm_rope_kernel<32>, dim3(cdiv(waves, 4)), dim3(256), 0, s, qkv, slot, pos, n_pos, heads,
                           kv_heads, qn_w, kn_w, eps, rope_cos, rope_sin, qr, cache);
        if (vt) hipLaunchKernelGGL(v_transpose_kernel<32>, dim3(cdiv(max_len, 64), kv_heads, n_clips), dim3(256), 0, s,
                                   cache, cu, slot_of_clip, vt, vt_stride);
    } else {
        throw std::invalid_argument("qk_norm_rope: unsupported (head_dim, head count)");
    }
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
template <int HD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void prefill_attention2_kernel(const bf16_t* __restrict__ qr, KVLayout cache,
                                                                 const bf16_t* __restrict__ vt, int vt_stride,
                                                                 const int* __restrict__ cu,
                                                                 const int* __restrict__ slot_of_clip, int heads,
                                                                 bf16_t* __restrict__ out, float scale) {
    constexpr int KT = 64, KS = HD / 32, DT = HD / 16, KCH = HD / 8, REP = 2;
    constexpr int TILE_BYTES = KT * HD * 2;                       // K tile and V^T tile have the same size
    constexpr int K_RPI = 64 / KCH, K_IPW = KT / K_RPI / 4;       // rows per wave instruction, instructions per wave
    constexpr int V_IPW = HD / 8 / 4;                             // V^T rows are 128 B: 8 rows per instruction
    static_assert(K_IPW >= 1 && V_IPW >= 1, "tile staging geometry");
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef const __attribute__((address_space(1))) void* glb_ptr_t;
    __shared__ __attribute__((aligned(16))) char smem[2][2][TILE_BYTES];   // [buffer][K | V^T]
    const int clip = blockIdx.z, kvh = blockIdx.y, q0 = blockIdx.x * 64;
    const int row0 = cu[clip], T = cu[clip + 1] - row0;
    if (q0 >= T) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, g = lane >> 4;
    const int sl = slot_of_clip[clip];
    const bf16_t* kbase = cache.k + cache.off(sl, kvh, 0);
    const bf16_t* vbase = vt + ((long)sl * cache.kv_heads + kvh) * HD * vt_stride;
    const int qw = q0 + wave * 16, qpos = qw + fr;

    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < K_IPW; ++i) {
            const int inst = wave * K_IPW + i;
            const int r = inst * K_RPI + lane / KCH, c = lane % KCH;
            int key = k0 + r;
            key = key < cache.max_ctx ? key : cache.max_ctx - 1;          // rows past the prompt are masked, not read as data
            const bf16_t* src = kbase + (long)key * HD + ((c ^ (r & (KCH - 1))) << 3);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&smem[buf][0][inst * 1024], 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < V_IPW; ++i) {
            const int inst = wave * V_IPW + i;
            const int d = inst * 8 + (lane >> 3), c = lane & 7;
            const bf16_t* src = vbase + (long)d * vt_stride + k0 + ((c ^ (d & 7)) << 3);   // V^T is zero past the prompt
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)&smem[buf][1][inst * 1024], 16, 0, 0);
        }
    };

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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // LDS-DMA completion is tracked by vmcnt only (also covers qf)
    __syncthreads();

    for (int kt = 0; kt < n_tiles; ++kt) {
        const int cur = kt & 1, k0 = kt * KT;
        if (kt + 1 < n_tiles) stage(cur ^ 1, k0 + KT);    // streams in under this tile's MFMAs
        if (k0 <= qw + 15 && qw < T) {                    // this wave has unmasked keys in the tile (wave-uniform)
            const char* s_k = smem[cur][0];
            const char* s_v = smem[cur][1];
            f32x4 sc[REP][4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int key = nb * 16 + fr;
                mfma_bf16x8 kf[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    kf[s] = *reinterpret_cast<const mfma_bf16x8*>(s_k + key * (HD * 2) + (((s * 4 + g) ^ (key & (KCH - 1))) << 4));
#pragma unroll
                for (int mi = 0; mi < REP; ++mi) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[s], qf[mi][s], acc, 0, 0, 0);
                    sc[mi][nb] = acc;
                }
            }
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
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
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
                        rs += bf16_lo(pw) + bf16_hi(pw);
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
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                mfma_bf16x8 pb[REP];
#pragma unroll
                for (int mi = 0; mi < REP; ++mi)
                    pb[mi] = __builtin_bit_cast(mfma_bf16x8, make_uint4(pk[mi][p][0], pk[mi][p][1], pk[mi][p][2], pk[mi][p][3]));
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const int dr = d * 16 + fr;
                    // A operand rows = d; k-slots 8g..8g+7 <- keys {32p + 4g + j, 32p + 16 + 4g + j}: two 8-byte reads
                    const char* vrow = s_v + dr * 128 + (g & 1) * 8;
                    const uint2 lo = *reinterpret_cast<const uint2*>(vrow + (((4 * p + (g >> 1)) ^ (dr & 7)) << 4));
                    const uint2 hi = *reinterpret_cast<const uint2*>(vrow + (((4 * p + 2 + (g >> 1)) ^ (dr & 7)) << 4));
                    const mfma_bf16x8 vf = __builtin_bit_cast(mfma_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
                    for (int mi = 0; mi < REP; ++mi) o[mi][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[mi], o[mi][d], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // tile kt+1 has landed (this wave's part) ...
        __syncthreads();                                // ... and every wave's part after the barrier
    }
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
                              hipStream_t s) {
    if (n_clips <= 0 || max_len <= 0) return;
    const int mt = tuning().pa_mt;          // A/B knob: row tiles per wave
    const int form = tuning().pa_form;      // A/B knob: 2 = transposed-score form
    const float scale = 1.0f / sqrtf((float)cache.hd);
    if (form >= 2 && heads == 2 * cache.kv_heads && (cache.hd == 128 || cache.hd == 32)) {
        const dim3 grid(cdiv(max_len, 64), cache.kv_heads, n_clips);
        if (cache.hd == 128)
            hipLaunchKernelGGL((prefill_attention2_kernel<128>), grid, dim3(256), 0, s, qr, cache, vt, vt_stride, cu, slot_of_clip, heads, out, scale);
        else
            hipLaunchKernelGGL((prefill_attention2_kernel<32>), grid, dim3(256), 0, s, qr, cache, vt, vt_stride, cu, slot_of_clip, heads, out, scale);
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

// ------------------------------------------------------------------------------------------------
// Decode-step skinny GEMM:  out[b][n] = sum_k X[b][k] W[n][k],  b < B <= 16*NB.
// Workgroup = 4 waves over 16*NT weight rows; wave w takes k-steps w, w+4, ... of 32 columns.  The
// weight fragment of v_mfma_f32_16x16x32_bf16 (lane: row l&15, 8 consecutive k at 8*(l>>4)) is loaded
// straight from HBM -- every weight byte is fetched once, 64 contiguous bytes per row per step -- the
// activation fragment (lane: batch row l&15) comes from L2.  Partial sums of the 4 waves meet in LDS.
// ------------------------------------------------------------------------------------------------
template <int NT, int NB, int EPI>
__global__ __launch_bounds__(256) void decode_gemv_kernel(DecGemvArgs a) {
    __shared__ __attribute__((aligned(16))) float s_red[3][NT * NB][64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fc = lane >> 4;
    const int n0 = blockIdx.x * 16 * NT;
    const int K = a.K, nsteps = K / 32;
    f32x4 acc[NT][NB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = a.W + (long)(n0 + t * 16 + fr) * K + fc * 8;
    const bf16_t* xp[NB];
    bool xv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        xv[b] = b * 16 + fr < a.B;
        xp[b] = a.X + (long)(xv[b] ? b * 16 + fr : 0) * K + fc * 8;
    }
    constexpr int UNR = 4;
    int s = wave;
    for (; s + 4 * (UNR - 1) < nsteps; s += 4 * UNR) {
        uint4 wf[UNR][NT], xf[UNR][NB];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int k = (s + 4 * u) * 32;
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[u][t] = *reinterpret_cast<const uint4*>(wp[t] + k);
#pragma unroll
            for (int b = 0; b < NB; ++b) xf[u][b] = xv[b] ? *reinterpret_cast<const uint4*>(xp[b] + k) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[t][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, wf[u][t]),
                                                                        __builtin_bit_cast(mfma_bf16x8, xf[u][b]), acc[t][b], 0, 0, 0);
    }
    for (; s < nsteps; s += 4) {
        const int k = s * 32;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            uint4 wf = *reinterpret_cast<const uint4*>(wp[t] + k);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                uint4 xf = xv[b] ? *reinterpret_cast<const uint4*>(xp[b] + k) : make_uint4(0, 0, 0, 0);
                acc[t][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, wf),
                                                                    __builtin_bit_cast(mfma_bf16x8, xf), acc[t][b], 0, 0, 0);
            }
        }
    }
    // cross-wave reduction in a fixed order (wave 0 + 1 + 2 + 3): deterministic
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                *reinterpret_cast<f32x4*>(&s_red[wave - 1][t * NB + b][lane * 4]) = acc[t][b];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                f32x4 r = *reinterpret_cast<const f32x4*>(&s_red[w][t * NB + b][lane * 4]);
                acc[t][b] += r;
            }
    dec_epilogue<NT, NB, EPI>(a, acc, n0, fr, fc);
}

// ------------------------------------------------------------------------------------------------
// Decode-step skinny GEMM, tuned form ("weights stationary in registers, activations in LDS").
//   * every wave first issues ALL of its weight-fragment loads (KSW k-steps x NT row tiles, 16 B per
//     lane each) so the whole weight matrix is in flight across the chip at once -- a 4..12 MB matrix
//     is latency-, not bandwidth-limited, so memory-level parallelism is what matters;
//   * while those are in flight the workgroup stages the activation rows into LDS (optionally
//     applying RMSNorm: y = bf16(w * bf16(x * inv)) -- the separate norm launch disappears);
//   * MFMA B fragments are then ds_read_b128 from the padded LDS image (row stride 2K + 16 bytes:
//     16 batch rows x one 16-byte chunk cover all 64 banks once);
//   * if the LDS image of all batch rows does not fit, rows are processed 16 at a time against the
//     same register-resident weights.
// k-steps are interleaved over the waves (wave w owns steps w, w + WAVES, ...).
// ------------------------------------------------------------------------------------------------
enum DecPro { DEC_PRO_COPY = 0, DEC_PRO_RMSNORM = 1 };

#ifndef QASR_DIAG_STAMPS
#define QASR_DIAG_STAMPS 0     // 1 compiles the in-kernel phase stamps (100 MHz wall clock) into the decode kernels:
#endif                         // diagnostic builds only (make DIAG=1); they cost the product path ~0.4 us per launch
struct DecGemv2Args {
    DecGemvArgs g;
    const bf16_t* norm_w;      // RMSNORM prologue: weight [K]
    float eps;
    unsigned long long* dbg;   // diagnostic phase stamps (see decode_gemv_stamps), null in product launches
    int row_groups;            // > 1: gridDim.y groups of rows_per_group batch rows (see gemv2_nb)
    int rows_per_group;
};

template <int NT, int NB, int WAVES, int KSW, bool ALLROWS, int PRO, int EPI>
__global__ __launch_bounds__(WAVES * 64) void decode_gemv2_kernel(DecGemv2Args a2) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    DecGemvArgs a = a2.g;
    if (gridDim.y > 1) {          // batch rows split over blockIdx.y in groups of 16 * NB (BF16 / RESID epilogues only)
        const int rpg = a2.rows_per_group, r0 = blockIdx.y * rpg;
        a.X += (long)r0 * a.K;
        a.out += (long)r0 * (EPI == DEC_EPI_SWIGLU ? a.N / 2 : a.N);
        a.B = a.B - r0 < rpg ? a.B - r0 : rpg;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fc = lane >> 4;
    constexpr int K = KSW * WAVES * 32;                           // host checks a.K == K
    constexpr int KCH = K / 8;                                    // 16-byte chunks per row
    constexpr int XSTRIDE = 2 * K + 16;                           // bytes
    constexpr int NPH = ALLROWS ? 1 : NB;                         // phases
    constexpr int NBP = ALLROWS ? NB : 1;                         // batch tiles resident per phase
    constexpr int RPP = 16 * NBP;
    constexpr int TPR = WAVES * 64 / RPP;                         // threads sharing one activation row
    constexpr int XI = KCH / TPR;                                 // staged 16-byte chunks per thread per phase
    static_assert(TPR >= 1 && TPR <= 64 && (TPR & (TPR - 1)) == 0 && KCH % TPR == 0, "row staging geometry");
    const int n0 = blockIdx.x * 16 * NT;
    char* s_x = dsm;                                              // [RPP][XSTRIDE]
    float* s_red = reinterpret_cast<float*>(dsm + (size_t)RPP * XSTRIDE);   // [WAVES-1][NT*NB][256]
    // thread -> (row srow, column chunks scol + TPR*i): a row lives on TPR adjacent lanes of one wave, so
    // its sum of squares needs log2(TPR) shuffles and no LDS round trip
    const int srow = tid / TPR, scol = tid % TPR;
#if QASR_DIAG_STAMPS
#define QASR_STAMP(i) do { if (a2.dbg && lane == 0) a2.dbg[((long)blockIdx.x * 16 + wave) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define QASR_STAMP(i) do { } while (0)
#endif
    QASR_STAMP(0);
    uint4 xr[XI];
    auto issue_x = [&](int r0) {
        // unconditional loads (clamped row), zeroed afterwards by a select: no branch, no per-load drain
        const bf16_t* xp = a.X + (long)(r0 + srow < a.B ? r0 + srow : 0) * K + scol * 8;
#pragma unroll
        for (int i = 0; i < XI; ++i) xr[i] = *reinterpret_cast<const uint4*>(xp + i * TPR * 8);
    };
    // Zeroing right after the loads makes the wait for X precede the weight loads on purpose: measured in the real
    // decode step (cold weights from HBM), issuing the weight stream -- or even just the norm weights -- ahead of that
    // wait is SLOWER (decode 149.3 -> 157.6 / 151.1 ms at B=32) although a warm-cache probe of the kernel alone gets
    // faster; the same holds for the copy-prologue kernels alone (o-proj / down: decode 145.9 -> 149.7 ms).  X is on the
    // critical path (staging + barrier) and, after a kernel boundary, is itself a fabric read: with the weight stream
    // queued right behind it, the X reads of later waves wait behind the weight misses of earlier ones.
    auto mask_x = [&](int r0) {
        if (r0 + srow >= a.B) {
#pragma unroll
            for (int i = 0; i < XI; ++i) xr[i] = make_uint4(0, 0, 0, 0);
        }
    };
    // ---- 1. activation loads, then (once they are back, see mask_x) ALL weight fragments ------------------
    issue_x(0);
    mask_x(0);
    uint4 w[NT][KSW];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        // fragment-major packed weights: block (row tile, k-step) = 1 KiB, lane-major -> every wave
        // instruction reads 1 KiB contiguous (see pack_mfma_a_kernel)
        const bf16_t* wp = a.Wp + ((long)(n0 / 16 + t) * (K / 32)) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < KSW; ++i) w[t][i] = *reinterpret_cast<const uint4*>(wp + (long)(wave + WAVES * i) * 512);
    }
    uint2 rsd[NT][NB];
    if constexpr (EPI == DEC_EPI_RESID) {
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int row = b * 16 + fr;
                    rsd[t][b] = *reinterpret_cast<const uint2*>(a.out + (long)(row < a.B ? row : 0) * a.N + n0 + t * 16 + fc * 4);
                }
        }
    }
    f32x4 acc[NT][NB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    QASR_STAMP(1);
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        if (ph > 0) {
            __syncthreads();                                      // previous phase's LDS reads are done
            issue_x(ph * RPP);
            mask_x(ph * RPP);
        }
        // ---- 2. activation rows -> LDS (the weight loads stay in flight) -------------------------------
        char* xrow = s_x + (size_t)srow * XSTRIDE + scol * 16;
        if constexpr (PRO == DEC_PRO_RMSNORM) {
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const bf16_t* e = reinterpret_cast<const bf16_t*>(&xr[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) { float f = bf16_to_f32(e[j]); ss = fmaf(f, f, ss); }
            }
            if constexpr (TPR >= 8) ss = lane_sum<(TPR >= 8 ? TPR : 8)>(ss);
            else {
#pragma unroll
                for (int ofs = 1; ofs < TPR; ofs <<= 1) ss += __shfl_xor(ss, ofs, 64);
            }
            const float inv = rsqrtf(ss / (float)K + a2.eps);
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const uint4 nw = reinterpret_cast<const uint4*>(a2.norm_w)[scol + i * TPR];
                const bf16_t* e = reinterpret_cast<const bf16_t*>(&xr[i]);
                const bf16_t* we = reinterpret_cast<const bf16_t*>(&nw);
                uint4 o;
                bf16_t* oe = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
                for (int j = 0; j < 8; ++j) oe[j] = f32_to_bf16(bf16_to_f32(we[j]) * bf16_round(bf16_to_f32(e[j]) * inv));
                *reinterpret_cast<uint4*>(xrow + i * TPR * 16) = o;
            }
        } else {
#pragma unroll
            for (int i = 0; i < XI; ++i) *reinterpret_cast<uint4*>(xrow + i * TPR * 16) = xr[i];
        }
        if (ph == 0) QASR_STAMP(2);
        __syncthreads();
        if (ph == 0) QASR_STAMP(3);
        // ---- 3. MFMA: register-resident weights x LDS activations --------------------------------------
#pragma unroll
        for (int i = 0; i < KSW; ++i) {
            const int kb = ((wave + WAVES * i) * 32 + fc * 8) * 2;
#pragma unroll
            for (int b = 0; b < NBP; ++b) {
                const uint4 xf = *reinterpret_cast<const uint4*>(s_x + (size_t)(b * 16 + fr) * XSTRIDE + kb);
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t][ALLROWS ? b : ph] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(mfma_bf16x8, w[t][i]), __builtin_bit_cast(mfma_bf16x8, xf),
                        acc[t][ALLROWS ? b : ph], 0, 0, 0);
            }
        }
    }
    QASR_STAMP(4);
    // ---- 4. cross-wave reduction in fixed order, epilogue on wave 0 ------------------------------------
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int b = 0; b < NB; ++b)
                *reinterpret_cast<f32x4*>(&s_red[((size_t)(wave - 1) * NT * NB + t * NB + b) * 256 + lane * 4]) = acc[t][b];
    }
    __syncthreads();
    QASR_STAMP(5);
    if (wave != 0) return;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int wv = 0; wv < WAVES - 1; ++wv)
                acc[t][b] += *reinterpret_cast<const f32x4*>(&s_red[((size_t)wv * NT * NB + t * NB + b) * 256 + lane * 4]);
    if constexpr (EPI == DEC_EPI_RESID) dec_epilogue<NT, NB, EPI>(a, acc, n0, fr, fc, rsd);
    else dec_epilogue<NT, NB, EPI>(a, acc, n0, fr, fc);
    QASR_STAMP(6);
#undef QASR_STAMP
}

// Fragment-major repack of a row-major [N][K] weight for v_mfma_f32_16x16x32_bf16 A operands:
// dst[((tile * K/32 + kstep) * 64 + lane) * 8 + j] = src[tile*16 + (lane & 15)][kstep*32 + (lane >> 4)*8 + j]
__global__ void pack_mfma_a_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N, int K) {
    const long chunk = (long)blockIdx.x * blockDim.x + threadIdx.x;       // one 16-byte chunk per thread
    const long total = (long)N * K / 8;
    if (chunk >= total) return;
    const int lane = (int)(chunk & 63);
    const long blk = chunk >> 6;
    const int ks = (int)(blk % (K / 32));
    const long tile = blk / (K / 32);
    const uint4 v = *reinterpret_cast<const uint4*>(src + (tile * 16 + (lane & 15)) * K + ks * 32 + (lane >> 4) * 8);
    reinterpret_cast<uint4*>(dst)[chunk] = v;
}

void pack_mfma_a_launch(const bf16_t* src, bf16_t* dst, int N, int K, hipStream_t s) {
    if (N % 16 != 0 || K % 32 != 0) throw std::invalid_argument("pack: N must be a multiple of 16 and K of 32");
    const long total = (long)N * K / 8;
    hipLaunchKernelGGL(pack_mfma_a_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, src, dst, N, K);
}

static int dec_nt(DecEpi epi, int N) {
    if (epi == DEC_EPI_LOGITS) return N % 64 == 0 ? 4 : (N % 32 == 0 ? 2 : 1);
    if (epi == DEC_EPI_SWIGLU) return 2;
    return 1;
}

int decode_gemv_blocks(DecEpi epi, int N) { return N / (16 * dec_nt(epi, N)); }

template <int NT, int EPI>
static void dec_launch_nb(const DecGemvArgs& a, int blocks, hipStream_t s) {
    const int nb = (a.B + 15) / 16;
    switch (nb) {
        case 1: hipLaunchKernelGGL((decode_gemv_kernel<NT, 1, EPI>), dim3(blocks), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL((decode_gemv_kernel<NT, 2, EPI>), dim3(blocks), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL((decode_gemv_kernel<NT, 3, EPI>), dim3(blocks), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL((decode_gemv_kernel<NT, 4, EPI>), dim3(blocks), dim3(256), 0, s, a); break;
        default: throw std::length_error("decode batch > 64 rows");
    }
}

static int decode_gemv_generic(DecEpi epi, const DecGemvArgs& a, hipStream_t s) {
    if (a.B <= 0) return 0;
    if (a.K % 32 != 0) throw std::invalid_argument("decode gemv: K must be a multiple of 32");
    const int nt = dec_nt(epi, a.N);
    if (a.N % (16 * nt) != 0) throw std::invalid_argument("decode gemv: N not a multiple of the row tile");
    const int blocks = a.N / (16 * nt);
    switch (epi) {
        case DEC_EPI_BF16: dec_launch_nb<1, DEC_EPI_BF16>(a, blocks, s); break;
        case DEC_EPI_RESID: dec_launch_nb<1, DEC_EPI_RESID>(a, blocks, s); break;
        case DEC_EPI_SWIGLU: dec_launch_nb<2, DEC_EPI_SWIGLU>(a, blocks, s); break;
        case DEC_EPI_LOGITS:
            if (nt == 4) dec_launch_nb<4, DEC_EPI_LOGITS>(a, blocks, s);
            else if (nt == 2) dec_launch_nb<2, DEC_EPI_LOGITS>(a, blocks, s);
            else dec_launch_nb<1, DEC_EPI_LOGITS>(a, blocks, s);
            break;
    }
    return blocks;
}

int decode_gemv_launch(DecEpi epi, const DecGemvArgs& a, hipStream_t s) { return decode_gemv_generic(epi, a, s); }

// ---- tuned dispatch ---------------------------------------------------------------------------------
template <int NT, int NB, int WAVES, int KSW, bool ALLROWS>
constexpr size_t gemv2_lds() {
    return (size_t)(ALLROWS ? 16 * NB : 16) * (2 * (KSW * WAVES * 32) + 16) + (size_t)(WAVES - 1) * NT * NB * 1024;
}

template <int NT, int NB, int WAVES, int KSW, bool ALLROWS, int PRO, int EPI>
static bool gemv2_go(const DecGemv2Args& a2, hipStream_t s) {
    constexpr size_t lds = gemv2_lds<NT, NB, WAVES, KSW, ALLROWS>();
    if constexpr (lds > 156 * 1024) {
        return false;
    } else {
        auto kern = decode_gemv2_kernel<NT, NB, WAVES, KSW, ALLROWS, PRO, EPI>;
        static bool attr_set = false;
        if (!attr_set) {
            QASR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3(a2.g.N / (16 * NT), a2.row_groups > 1 ? a2.row_groups : 1), dim3(WAVES * 64), lds, s, a2);
        return true;
    }
}

template <int NT, int WAVES, int KSW, int PRO, int EPI>
static bool gemv2_nb(const DecGemv2Args& a2, hipStream_t s) {
    const int nb = (a2.g.B + 15) / 16;
    // One workgroup per (row tile, 16 batch rows) -- gridDim.y row groups -- instead of one workgroup walking all batch tiles:
    // twice the workgroups at 32 rows, half the activation bytes (and RMSNorm work) per workgroup, no second staging phase
    // for the K = 3072 matrix.  Measured in the real step, decode at 32 x 30 s: 148.8 ms unsplit, 140.2 ms with only the two
    // residual GEMVs (64 -> 128 workgroups) split, 137.3 ms with all four; 8-row groups (151 ms) and 4-wave workgroups
    // (140 ms) lose.  tuning knob gemv_splitb = 0|1|2 selects none | residual only | all (A/B).
    const int split_b = tuning().gemv_splitb;
    if (nb > 1 && EPI != DEC_EPI_LOGITS && ((split_b == 1 && EPI == DEC_EPI_RESID) || split_b == 2)) {
        DecGemv2Args b2 = a2;
        b2.rows_per_group = 16;
        b2.row_groups = nb;
        return gemv2_go<NT, 1, WAVES, KSW, true, PRO, EPI>(b2, s);
    }
    // all batch rows resident in LDS when they fit (LDS and staging registers), else 16 rows per phase
    constexpr bool fit2 = gemv2_lds<NT, 2, WAVES, KSW, true>() <= 150 * 1024 && NT * KSW * 4 + 2 * KSW * 4 <= 170;
    switch (nb) {
        case 1: return gemv2_go<NT, 1, WAVES, KSW, true, PRO, EPI>(a2, s);
        case 2:
            if constexpr (fit2) return gemv2_go<NT, 2, WAVES, KSW, true, PRO, EPI>(a2, s);
            else return gemv2_go<NT, 2, WAVES, KSW, false, PRO, EPI>(a2, s);
        case 3: return gemv2_go<NT, 3, WAVES, KSW, false, PRO, EPI>(a2, s);
        case 4: return gemv2_go<NT, 4, WAVES, KSW, false, PRO, EPI>(a2, s);
        default: return false;
    }
}

template <int PRO, int EPI, int NT>
static bool gemv2_k(const DecGemv2Args& a2, hipStream_t s) {
    // (K -> waves x k-steps per wave): wide workgroups for the small-N / large-K matrices
    switch (a2.g.K) {
        case 1024: {
            const bool w8 = tuning().gemv_w1024 == 8;   // A/B knob (8 waves x 4 k-steps won)
            if constexpr (EPI == DEC_EPI_LOGITS) return gemv2_nb<NT, 8, 4, PRO, EPI>(a2, s);
            else return w8 ? gemv2_nb<NT, 8, 4, PRO, EPI>(a2, s) : gemv2_nb<NT, 4, 8, PRO, EPI>(a2, s);
        }
        case 2048: return gemv2_nb<NT, 8, 8, PRO, EPI>(a2, s);
        case 3072:            // K = intermediate size: never behind a norm
            if constexpr (PRO == DEC_PRO_COPY) return gemv2_nb<NT, 8, 12, PRO, EPI>(a2, s);
            else return false;
        default: return false;
    }
}

// Fused form used by the decode step: optional RMSNorm prologue (norm_w != null) + epilogue.
// Falls back to [rmsnorm_rows +] the generic kernel for shapes without a tuned instantiation.
static unsigned long long* g_gemv_dbg = nullptr;
void decode_gemv_set_debug(unsigned long long* dbg) { g_gemv_dbg = dbg; }
int decode_gemv_fused_launch(DecEpi epi, const DecGemvArgs& a, const bf16_t* norm_w, float eps, bf16_t* norm_scratch,
                             hipStream_t s) {
    if (a.B <= 0) return 0;
    const int nt = dec_nt(epi, a.N);
    DecGemv2Args a2{a, norm_w, eps, g_gemv_dbg, 1, 16};
    bool ok = false;
    if (a.Wp && a.N % (16 * nt) == 0 && a.B <= 64) {
        if (norm_w) {
            if (epi == DEC_EPI_BF16) ok = gemv2_k<DEC_PRO_RMSNORM, DEC_EPI_BF16, 1>(a2, s);
            else if (epi == DEC_EPI_SWIGLU) ok = gemv2_k<DEC_PRO_RMSNORM, DEC_EPI_SWIGLU, 2>(a2, s);
            else if (epi == DEC_EPI_LOGITS && nt == 4) ok = gemv2_k<DEC_PRO_RMSNORM, DEC_EPI_LOGITS, 4>(a2, s);
        } else {
            if (epi == DEC_EPI_RESID) ok = gemv2_k<DEC_PRO_COPY, DEC_EPI_RESID, 1>(a2, s);
            else if (epi == DEC_EPI_BF16) ok = gemv2_k<DEC_PRO_COPY, DEC_EPI_BF16, 1>(a2, s);
        }
    }
    if (ok) return a.N / (16 * nt);
    DecGemvArgs g = a;
    if (norm_w) {
        rmsnorm_rows_launch(a.X, norm_w, norm_scratch, a.B, a.K, eps, s);
        g.X = norm_scratch;
    }
    return decode_gemv_generic(epi, g, s);
}

// ------------------------------------------------------------------------------------------------
// LM head (tied embedding, N = vocab): persistent form.  The batch rows are normalised (final RMSNorm)
// and staged into LDS ONCE per workgroup; every wave then walks its own 16-row weight tiles over the
// full K with a two-deep register pipeline (16 k-steps = 16 KB per buffer in flight per wave), so
// there is no cross-wave reduction and no barrier after the staging.  Each wave keeps a running
// (max, lowest index) per batch row over bf16-rounded logits; the workgroup publishes one partial.
// ------------------------------------------------------------------------------------------------
constexpr int LMH_WAVES = 8, LMH_CH = 16;     // waves per workgroup, k-steps per register buffer

struct LmHeadArgs {
    const bf16_t* W;        // fragment-major packed [N/16][K/32][64 lanes][8]
    const bf16_t* X;        // [B][K] un-normalised hidden rows
    const bf16_t* norm_w;   // [K]
    float eps;
    int B, N;
    float* logits;          // optional [B][N]
    float* part_val;        // [B][gridDim.x]
    int* part_idx;
    int diag;               // 1: diagnostic build of the loop without LDS reads / MFMA (wrong results, timing only)
};

template <int K, int NB>
__global__ __launch_bounds__(LMH_WAVES * 64) void lm_head_kernel(LmHeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    constexpr int KCH = K / 8, XSTRIDE = 2 * K + 16, NC = K / (32 * LMH_CH);   // chunks per tile
    constexpr int TPR = 32, XI = KCH / TPR;                                      // 16 rows per staging pass
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fc = lane >> 4;
    char* s_x = dsm;
    // ---- weights of the first chunk go in flight before the activation staging -------------------------
    const int total_waves = gridDim.x * LMH_WAVES, gw = blockIdx.x * LMH_WAVES + wave;
    const int ntiles = a.N / 16;
    const int my_tiles = gw < ntiles ? (ntiles - gw + total_waves - 1) / total_waves : 0;
    const int nitems = my_tiles * NC;
    uint4 wa[LMH_CH], wb[LMH_CH];
    auto issue = [&](uint4 (&w)[LMH_CH], int item) {
        const int tile = gw + (item / NC) * total_waves, ch = item % NC;
        const bf16_t* wp = a.W + ((long)tile * (K / 32) + ch * LMH_CH) * 512 + lane * 8;   // packed, see pack_mfma_a_kernel
#pragma unroll
        for (int i = 0; i < LMH_CH; ++i) w[i] = *reinterpret_cast<const uint4*>(wp + i * 512);
    };
    if (nitems > 0) issue(wa, 0);
    // ---- stage + RMSNorm the batch rows, 16 rows per pass (row on 32 adjacent lanes) --------------------
    {
        const int srow = tid / TPR, scol = tid % TPR;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int r = nb * 16 + srow;
            const bool live = r < a.B;
            const bf16_t* xp = a.X + (long)(live ? r : 0) * K + scol * 8;
            uint4 xr[XI];
#pragma unroll
            for (int i = 0; i < XI; ++i) xr[i] = *reinterpret_cast<const uint4*>(xp + i * TPR * 8);
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const bf16_t* e = reinterpret_cast<const bf16_t*>(&xr[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) { float f = bf16_to_f32(e[j]); ss = fmaf(f, f, ss); }
            }
            if constexpr (TPR >= 8) ss = lane_sum<(TPR >= 8 ? TPR : 8)>(ss);
            else {
#pragma unroll
                for (int ofs = 1; ofs < TPR; ofs <<= 1) ss += __shfl_xor(ss, ofs, 64);
            }
            const float inv = rsqrtf(ss / (float)K + a.eps);
            char* xrow = s_x + (size_t)r * XSTRIDE + scol * 16;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const uint4 nw = reinterpret_cast<const uint4*>(a.norm_w)[scol + i * TPR];
                const bf16_t* e = reinterpret_cast<const bf16_t*>(&xr[i]);
                const bf16_t* we = reinterpret_cast<const bf16_t*>(&nw);
                uint4 o;
                bf16_t* oe = reinterpret_cast<bf16_t*>(&o);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    oe[j] = live ? f32_to_bf16(bf16_to_f32(we[j]) * bf16_round(bf16_to_f32(e[j]) * inv)) : (bf16_t)0;
                *reinterpret_cast<uint4*>(xrow + i * TPR * 16) = o;
            }
        }
    }
    __syncthreads();
    // ---- stream the weight tiles ---------------------------------------------------------------------------
    f32x4 acc[NB];
    float best[NB];
    int bidx[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { acc[b] = f32x4{0.f, 0.f, 0.f, 0.f}; best[b] = -INFINITY; bidx[b] = 0x7fffffff; }
    auto consume = [&](const uint4 (&w)[LMH_CH], int item) {
        const int tile = gw + (item / NC) * total_waves, ch = item % NC;
        if (a.diag) {
#pragma unroll
            for (int i = 0; i < LMH_CH; ++i) acc[0][0] += __uint_as_float(w[i].x ^ w[i].y ^ w[i].z ^ w[i].w);
        } else
#pragma unroll
        for (int i = 0; i < LMH_CH; ++i) {
            const int kb = ((ch * LMH_CH + i) * 32 + fc * 8) * 2;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const uint4 xf = *reinterpret_cast<const uint4*>(s_x + (size_t)(b * 16 + fr) * XSTRIDE + kb);
                acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, w[i]),
                                                                 __builtin_bit_cast(mfma_bf16x8, xf), acc[b], 0, 0, 0);
            }
        }
        if (ch == NC - 1) {                                   // tile complete: acc[b][j] = logit[b*16+fr][tile*16+fc*4+j]
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int row = b * 16 + fr;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = tile * 16 + fc * 4 + j;
                    const float v = bf16_round(acc[b][j]);
                    if (a.logits && row < a.B) a.logits[(long)row * a.N + n] = v;
                    if (v > best[b] || (v == best[b] && n < bidx[b])) { best[b] = v; bidx[b] = n; }
                }
                acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    for (int item = 0; item < nitems; item += 2) {
        if (item + 1 < nitems) issue(wb, item + 1);
        consume(wa, item);
        if (item + 1 < nitems) {
            if (item + 2 < nitems) issue(wa, item + 2);
            consume(wb, item + 1);
        }
    }
    // ---- argmax partial of the workgroup ---------------------------------------------------------------------
    float* s_v = reinterpret_cast<float*>(dsm);               // the activation image is dead now
    int* s_i = reinterpret_cast<int*>(dsm + LMH_WAVES * NB * 16 * sizeof(float));
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int ofs = 16; ofs < 64; ofs <<= 1) {
            const float ov = __shfl_xor(best[b], ofs, 64);
            const int oi = __shfl_xor(bidx[b], ofs, 64);
            if (ov > best[b] || (ov == best[b] && oi < bidx[b])) { best[b] = ov; bidx[b] = oi; }
        }
        if (fc == 0) { s_v[(wave * NB + b) * 16 + fr] = best[b]; s_i[(wave * NB + b) * 16 + fr] = bidx[b]; }
    }
    __syncthreads();
    if (tid < NB * 16 && tid < a.B) {
        const int b = tid >> 4, r = tid & 15;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int w = 0; w < LMH_WAVES; ++w) {
            const float ov = s_v[(w * NB + b) * 16 + r];
            const int oi = s_i[(w * NB + b) * 16 + r];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        a.part_val[(long)tid * gridDim.x + blockIdx.x] = bv;
        a.part_idx[(long)tid * gridDim.x + blockIdx.x] = bi;
    }
}

static int lmh_grid() {
    static const int g = tuning().lmh_grid;      // frozen at first use: sizes the argmax partial buffers
    return g;
}

template <int K, int NB>
static void lm_head_go(const LmHeadArgs& a, hipStream_t s) {
    constexpr size_t lds = (size_t)NB * 16 * (2 * K + 16);
    auto kern = lm_head_kernel<K, NB>;
    static bool attr_set = false;
    if (!attr_set) {
        QASR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(lmh_grid()), dim3(LMH_WAVES * 64), lds, s, a);
}

bool lm_head_supported(int N, int K) { return (K == 1024 || K == 2048) && N % 16 == 0 && N / 16 >= 512 * LMH_WAVES; }
int lm_head_parts(int N, int K) { return lm_head_supported(N, K) ? lmh_grid() : decode_gemv_blocks(DEC_EPI_LOGITS, N); }

// final RMSNorm + tied LM head + per-workgroup argmax partials; returns the number of partials per row
int lm_head_launch(const bf16_t* W, const bf16_t* Wp, const bf16_t* X, const bf16_t* norm_w, float eps, int B, int N, int K,
                   float* logits, float* part_val, int* part_idx, bf16_t* norm_scratch, hipStream_t s) {
    if (B <= 0) return 0;
    const int nb = (B + 15) / 16;
    if (Wp && lm_head_supported(N, K) && nb <= (K == 1024 ? 4 : 2)) {
        const int diag = tuning().lmh_diag;
        LmHeadArgs a{Wp, X, norm_w, eps, B, N, logits, part_val, part_idx, diag};
        if (K == 1024) {
            switch (nb) {
                case 1: lm_head_go<1024, 1>(a, s); break;
                case 2: lm_head_go<1024, 2>(a, s); break;
                case 3: lm_head_go<1024, 3>(a, s); break;
                default: lm_head_go<1024, 4>(a, s); break;
            }
        } else {
            if (nb == 1) lm_head_go<2048, 1>(a, s); else lm_head_go<2048, 2>(a, s);
        }
        return lmh_grid();
    }
    if (Wp && lm_head_supported(N, K)) throw std::length_error("LM head: batch rows exceed the LDS image at this hidden size");
    DecGemvArgs g{};
    g.W = W; g.X = X; g.B = B; g.N = N; g.K = K; g.logits = logits; g.part_val = part_val; g.part_idx = part_idx;
    return decode_gemv_fused_launch(DEC_EPI_LOGITS, g, norm_w, eps, norm_scratch, s);
}

// ------------------------------------------------------------------------------------------------
// Decode attention on the matrix cores.  One workgroup per (kv head, batch row); a wave owns 32-key
// chunks of the context (chunk = wave, wave + WAVES, ...), all of whose loads are issued up front:
//   S^T = K Q^T   16x16x32 MFMA, A = 16 cached key rows straight from HBM (natural [key][hd] layout),
//                 B = the two query heads of this kv head in columns 0/1 (other columns zero);
//                 the accumulator puts keys (lane>>4)*4+j of query (lane&15) on a lane, which IS the
//                 A-operand layout of the next MFMA, so P never leaves registers;
//   O  += P V     16x16x32 MFMA over the chunk's 32 keys, B = V in the fragment-major cache image
//                 (KVLayout::vf), one 16-byte load per lane and d tile.
// Every wave norms + ropes the two query rows itself (no workgroup barrier before the sweep); the
// token's own key / value are computed by the last two waves, appended to the caches and folded in
// at the cross-wave merge.  Softmax statistics stay in f32; P is rounded to bf16 like the prompt pass.
// ------------------------------------------------------------------------------------------------
template <int HD>
__device__ __forceinline__ long vfrag_index(int key, int d) {
    // fragment-major V: [key/32][d/16][lane = (d%16) + 16*g][e = half*4 + j],  key%32 = half*16 + g*4 + j
    constexpr int DT = HD / 16;
    const int kb = key >> 5, r = key & 31, half = r >> 4, g = (r & 15) >> 2, j = r & 3;
    return (((long)kb * DT + (d >> 4)) * 64 + (d & 15) + 16 * g) * 8 + half * 4 + j;
}

template <int HD, int WAVES, int UNR, bool SPEC>
__global__ __launch_bounds__(WAVES * 64) void decode_attention_mfma_kernel(
    const bf16_t* __restrict__ qkv, const int* __restrict__ ctx_len, int heads, int kv_heads,
    const bf16_t* __restrict__ qn_w, const bf16_t* __restrict__ kn_w, float eps, const float* __restrict__ rope_cos,
    const float* __restrict__ rope_sin, KVLayout cache, bf16_t* __restrict__ out, float scale,
    unsigned long long* __restrict__ dbg) {
    constexpr int REP = 2, KS = HD / 32, DT = HD / 16, HALF = HD / 2;
#if QASR_DIAG_STAMPS
#define QASR_STAMP(i) do { if (dbg && (threadIdx.x & 63) == 0) dbg[((blockIdx.y * gridDim.x + blockIdx.x) * WAVES + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define QASR_STAMP(i) do { } while (0)
#endif
    QASR_STAMP(0);
    __shared__ __attribute__((aligned(16))) bf16_t s_q[WAVES][REP][HD];      // wave-private query image
    __shared__ float s_m[WAVES][REP], s_l[WAVES][REP];
    __shared__ float s_o[WAVES][REP][HD];
    __shared__ float s_new[REP], s_vn[HD];
    const int kvh = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const bf16_t* kb = cache.k + cache.off(b, kvh, 0) + g * 8;
    const bf16_t* vfb = cache.vf + cache.off(b, kvh, 0) + lane * 8;
    const int max_chunk = cache.max_ctx / 32 - 1;
    uint4 kreg[UNR][2 * KS], vreg[UNR][DT];
    auto issue = [&](int chunk0, int limit) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            int ch = chunk0 + u * WAVES;
            if (ch >= limit) continue;                                   // wave-uniform: no bytes for chunks past the context
            ch = ch < max_chunk ? ch : max_chunk;                        // clamped to the allocation, masked later
            const bf16_t* kr = kb + ((long)ch * 32 + fr) * HD;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    kreg[u][h * KS + ks] = *reinterpret_cast<const uint4*>(kr + (long)h * 16 * HD + ks * 32);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                vreg[u][dt] = *reinterpret_cast<const uint4*>(vfb + ((long)ch * DT + dt) * 512);
        }
    };
    int pos;
    if (SPEC) {
        issue(wave, 0x7fffffff);             // before the position is known: rows past it are masked
        pos = ctx_len[b];
    } else {
        pos = ctx_len[b];
        issue(wave, (pos + 31) >> 5);
    }
    const int nchunks = (pos + 31) >> 5;
    const int nh = heads + 2 * kv_heads;
    const bf16_t* row = qkv + (long)b * nh * HD;
    // ---- the two query heads: norm + rope on every wave, bf16 image in this wave's LDS slice ------------
    const bool act = lane < HALF;
    const float rc = act ? rope_cos[(long)b * HALF + lane] : 0.0f, rs = act ? rope_sin[(long)b * HALF + lane] : 0.0f;
    float qa[REP][2];
    {
        const float w1 = act ? bf16_to_f32(qn_w[lane]) : 0.0f, w2 = act ? bf16_to_f32(qn_w[lane + HALF]) : 0.0f;
        float x1[REP], x2[REP];
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const bf16_t* src = row + (long)(kvh * REP + r) * HD;
            x1[r] = act ? bf16_to_f32(src[lane]) : 0.0f;
            x2[r] = act ? bf16_to_f32(src[lane + HALF]) : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const float inv = rsqrtf(wave_sum(x1[r] * x1[r] + x2[r] * x2[r]) / (float)HD + eps);
            norm_rope_pair(x1[r], x2[r], w1, w2, inv, rc, rs, qa[r][0], qa[r][1]);
            if (act) {
                s_q[wave][r][lane] = f32_to_bf16(qa[r][0]);
                s_q[wave][r][lane + HALF] = f32_to_bf16(qa[r][1]);
            }
        }
    }
    // ---- the token's own key (wave WAVES-1) and value (wave WAVES-2): cache append + merge terms -------
    if (wave == WAVES - 1) {
        const bf16_t* src = row + (long)(heads + kvh) * HD;
        const float x1 = act ? bf16_to_f32(src[lane]) : 0.0f, x2 = act ? bf16_to_f32(src[lane + HALF]) : 0.0f;
        const float inv = rsqrtf(wave_sum(x1 * x1 + x2 * x2) / (float)HD + eps);
        float k1, k2;
        norm_rope_pair(x1, x2, act ? bf16_to_f32(kn_w[lane]) : 0.0f, act ? bf16_to_f32(kn_w[lane + HALF]) : 0.0f, inv, rc, rs, k1, k2);
        if (act) {
            bf16_t* dk = cache.k + cache.off(b, kvh, pos);
            dk[lane] = f32_to_bf16(k1);
            dk[lane + HALF] = f32_to_bf16(k2);
        }
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            const float d = wave_sum(act ? qa[r][0] * k1 + qa[r][1] * k2 : 0.0f);
            if (lane == 0) s_new[r] = d * scale;
        }
    } else if (wave == WAVES - 2) {
        const bf16_t* src = row + (long)(heads + kv_heads + kvh) * HD;
        bf16_t* dvf = cache.vf + cache.off(b, kvh, 0);
        for (int i = lane; i < HD; i += 64) {
            const bf16_t v = src[i];
            s_vn[i] = bf16_to_f32(v);
            dvf[vfrag_index<HD>(pos, i)] = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    mfma_bf16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (fr < REP) u = *reinterpret_cast<const uint4*>(&s_q[wave][fr][ks * 32 + g * 8]);
        qf[ks] = __builtin_bit_cast(mfma_bf16x8, u);
    }
    QASR_STAMP(1);
    // ---- sweep -------------------------------------------------------------------------------------
    f32x4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.0f;
    for (int c0 = wave; c0 < nchunks; c0 += WAVES * UNR) {
        if (c0 != wave) issue(c0, nchunks);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int chunk = c0 + u * WAVES;
            if (chunk < nchunks) {                                       // wave-uniform
                f32x4 sc[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, kreg[u][h * KS + ks]), qf[ks], acc, 0, 0, 0);
                    sc[h] = acc;
                }
                const int key0 = chunk * 32 + g * 4;
                float mx = -INFINITY;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = key0 + h * 16 + j < pos ? sc[h][j] * scale : -INFINITY;   // select: stale rows may be NaN
                        sc[h][j] = v;
                        mx = fmaxf(mx, v);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run, mx);                    // finite: a chunk below nchunks has a valid key
                const float alpha = __expf(m_run - m_new);
                float rsum = 0.0f;
                unsigned pk[4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const unsigned pw = pack_bf16x2(__expf(sc[h][j] - m_new), __expf(sc[h][j + 1] - m_new));
                        rsum += bf16_lo(pw) + bf16_hi(pw);
                        pk[h * 2 + j / 2] = pw;
                    }
                l_run = l_run * alpha + rsum;
                m_run = m_new;
                const float a0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(alpha), 0));
                const float a1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(alpha), 1));
                uint4 vv[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) vv[dt] = vreg[u][dt];
                if (chunk * 32 + 32 > pos) {                             // partial chunk: stale V rows would give 0 * NaN
                    unsigned msk[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const int k_lo = key0 + (w >> 1) * 16 + (w & 1) * 2;
                        msk[w] = (k_lo < pos ? 0x0000ffffu : 0u) | (k_lo + 1 < pos ? 0xffff0000u : 0u);
                    }
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) { vv[dt].x &= msk[0]; vv[dt].y &= msk[1]; vv[dt].z &= msk[2]; vv[dt].w &= msk[3]; }
                }
                const mfma_bf16x8 pa = __builtin_bit_cast(mfma_bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    o[dt][0] *= a0;
                    o[dt][1] *= a1;
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, __builtin_bit_cast(mfma_bf16x8, vv[dt]), o[dt], 0, 0, 0);
                }
            }
        }
    }
    QASR_STAMP(2);
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (lane < REP) { s_m[wave][lane] = m_run; s_l[wave][lane] = l_run; }
    if (g == 0) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            s_o[wave][0][dt * 16 + fr] = o[dt][0];
            s_o[wave][1][dt * 16 + fr] = o[dt][1];
        }
    }
    QASR_STAMP(3);
    QASR_STAMP(4);
    __syncthreads();
    QASR_STAMP(5);
    // merge waves + the token's own key/value: thread t < REP*HD owns one output element
    for (int i = tid; i < REP * HD; i += WAVES * 64) {
        const int r = i / HD, d = i - r * HD;
        float mm = s_new[r];
#pragma unroll
        for (int w = 0; w < WAVES; ++w) mm = fmaxf(mm, s_m[w][r]);
        const float pn = __expf(s_new[r] - mm);
        float num = pn * s_vn[d], den = pn;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const float mw = s_m[w][r];
            if (mw != -INFINITY) {                                       // waves without a chunk left s_o unwritten
                const float a = __expf(mw - mm);
                num += s_o[w][r][d] * a;
                den += s_l[w][r] * a;
            }
        }
        out[(long)b * heads * HD + (long)(kvh * REP + r) * HD + d] = f32_to_bf16(num / den);
    }
    QASR_STAMP(6);
#undef QASR_STAMP
}

void decode_attention_launch(const bf16_t* qkv, const int* ctx_len, int B, int heads, int kv_heads, int hd,
                             const bf16_t* qn_w, const bf16_t* kn_w, float eps, const float* rope_cos,
                             const float* rope_sin, KVLayout cache, bf16_t* out, hipStream_t s, unsigned long long* dbg) {
    if (B <= 0) return;
    if (heads != 2 * kv_heads) throw std::invalid_argument("decode attention: built for 2 query heads per kv head");
    if (!cache.vf) throw std::invalid_argument("decode attention: the fragment-major V image is not allocated");
    if (cache.max_ctx % 32) throw std::invalid_argument("decode attention: cache capacity must be a multiple of 32 keys");
    const float scale = 1.0f / sqrtf((float)hd);
    dim3 grid(kv_heads, B);
    // A/B knobs: waves per workgroup (8 with two chunks in flight per wave | 16 with one) and speculative first loads
    const int nw = tuning().da_waves;
    const int spec = tuning().da_spec;
#define QASR_DAM_GO(HD_, W_, U_, S_)                                                                                         \
    hipLaunchKernelGGL((decode_attention_mfma_kernel<HD_, W_, U_, S_>), grid, dim3(W_ * 64), 0, s, qkv, ctx_len, heads, kv_heads, \
                       qn_w, kn_w, eps, rope_cos, rope_sin, cache, out, scale, dbg)
    if (hd == 128) {
        if (nw == 16 && spec) QASR_DAM_GO(128, 16, 1, true);
        else if (nw == 16) QASR_DAM_GO(128, 16, 1, false);
        else if (spec) QASR_DAM_GO(128, 8, 2, true);
        else QASR_DAM_GO(128, 8, 2, false);
    } else if (hd == 32) {
        QASR_DAM_GO(32, 8, 1, false);
    } else
        throw std::invalid_argument("decode attention: unsupported head_dim");
#undef QASR_DAM_GO
}

// ------------------------------------------------------------------------------------------------
// Row argmax over bf16 logits with MLX argMax's tie rule (lowest index): one workgroup per row.  Used by the forced
// aligner's timestamp head (ForcedAligner.swift:291-296).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_rows_kernel(const bf16_t* __restrict__ x, long ld, int n, int* __restrict__ out) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const bf16_t* row = x + (long)blockIdx.x * ld;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = bf16_to_f32(row[i]);
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }     // NaN never wins
    }
    s_v[threadIdx.x] = bv;
    s_i[threadIdx.x] = bi;
    __syncthreads();
    for (int ofs = 128; ofs > 0; ofs >>= 1) {
        if ((int)threadIdx.x < ofs) {
            const float ov = s_v[threadIdx.x + ofs];
            const int oi = s_i[threadIdx.x + ofs];
            if (oi != 0x7fffffff && (s_i[threadIdx.x] == 0x7fffffff || ov > s_v[threadIdx.x] || (ov == s_v[threadIdx.x] && oi < s_i[threadIdx.x]))) {
                s_v[threadIdx.x] = ov;
                s_i[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = s_i[0] == 0x7fffffff ? 0 : s_i[0];
}

void argmax_rows_launch(const bf16_t* x, long ld, int rows, int n, int* out, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, s, x, ld, n, out);
}

// ------------------------------------------------------------------------------------------------
// greedy bookkeeping + next-token embedding gather: one workgroup per batch row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void greedy_finalize_kernel(const float* __restrict__ part_val,
                                                              const int* __restrict__ part_idx, int n_parts,
                                                              GreedyState st, int advance_ctx,
                                                              const bf16_t* __restrict__ embed, bf16_t* __restrict__ x, int H,
                                                              RopeRows rr, QuantRaw qe) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    // position of the NEXT decode step for this row; its rope row is copied next to the batch row so that
    // the attention kernels of that step need not chain a table lookup behind the ctx_len load
    const int next_pos = st.ctx_len[b] + (advance_ctx ? 1 : 0);
    if (tid < rr.half) {
        rr.cos_rows[(long)b * rr.half + tid] = rr.cos_table[(long)next_pos * rr.half + tid];
        rr.sin_rows[(long)b * rr.half + tid] = rr.sin_table[(long)next_pos * rr.half + tid];
    }
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int i = tid; i < n_parts; i += 256) {
        const float v = part_val[(long)b * n_parts + i];
        const int n = part_idx[(long)b * n_parts + i];
        if (v > best || (v == best && n < bidx)) { best = v; bidx = n; }
    }
    s_v[tid] = best;
    s_i[tid] = bidx;
    __syncthreads();
    for (int ofs = 128; ofs > 0; ofs >>= 1) {
        if (tid < ofs) {
            const float ov = s_v[tid + ofs];
            const int oi = s_i[tid + ofs];
            if (ov > s_v[tid] || (ov == s_v[tid] && oi < s_i[tid])) { s_v[tid] = ov; s_i[tid] = oi; }
        }
        __syncthreads();
    }
    const bool sane = (unsigned)s_i[0] < (unsigned)st.vocab && fabsf(s_v[0]) <= 3.0e38f;   // false for NaN / inf / no winner
    const int tok = (unsigned)s_i[0] < (unsigned)st.vocab ? s_i[0] : 0;
    if (tid == 0) {
        if (!sane && !st.finished[b]) atomicOr(st.err, 1);
        if (advance_ctx) st.ctx_len[b] += 1;
        if (!st.finished[b]) {
            const int n = st.lens[b];
            st.tokens[(long)b * (st.max_new + 1) + n] = tok;
            st.lens[b] = n + 1;
            if ((tok == st.eos && !st.ignore_eos) || n + 1 >= st.max_tokens) {
                st.finished[b] = 1;
                atomicSub(st.n_active, 1);
            }
        }
    }
    uint4* dst = reinterpret_cast<uint4*>(x + (long)b * H);
    if (qe.wq) {                                     // quantised table: dequantized(row) (PreQuantizedEmbedding.swift:35-42)
        for (int i = tid; i < H / 8; i += 256) dst[i] = quant_dequant_chunk(qe, tok, i);
        return;
    }
    const uint4* src = reinterpret_cast<const uint4*>(embed + (long)tok * H);
    for (int i = tid; i < H / 8; i += 256) dst[i] = src[i];
}

void greedy_finalize_launch(const float* part_val, const int* part_idx, int n_parts, GreedyState st, int B,
                            int advance_ctx, const bf16_t* embed, bf16_t* x, int H, RopeRows rr, hipStream_t s,
                            const QuantRaw* qembed) {
    if (B <= 0) return;
    hipLaunchKernelGGL(greedy_finalize_kernel, dim3(B), dim3(256), 0, s, part_val, part_idx, n_parts, st, advance_ctx,
                       embed, x, H, rr, qembed ? *qembed : QuantRaw{});
}

}  // namespace qasr

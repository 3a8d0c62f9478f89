// ctc_kernels.hip -- Omnilingual (wav2vec2-CTC) kernels, see ctc_kernels.h.
#include "ctc_kernels.h"
#include "tuning.h"

namespace qasr {

// ------------------------------------------------------------------------------------------------
// utterance statistics: f32 sums of x and x^2 (the reference sums sequentially in f32; a tree sum differs in the last
// bits of mean / variance, far below the bf16 operand rounding that follows)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void wave_stats_kernel(const float* __restrict__ pcm, const long* __restrict__ pcm_off,
                                                          const int* __restrict__ n_samples, float eps, float* __restrict__ stats) {
    __shared__ float s_a[16], s_b[16];
    const int b = blockIdx.x, n = n_samples[b];
    const float* x = pcm + pcm_off[b];
    float s = 0.0f, ss = 0.0f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const float v = x[i]; s += v; ss = fmaf(v, v, ss); }
    s = wave_sum(s);
    ss = wave_sum(ss);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { s_a[wave] = s; s_b[wave] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0.0f, tss = 0.0f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ts += s_a[w]; tss += s_b[w]; }
        const float mean = n > 0 ? ts / (float)n : 0.0f;
        const float var = n > 0 ? fmaxf(0.0f, tss / (float)n - mean * mean) : 0.0f;
        stats[2 * b] = mean;
        stats[2 * b + 1] = 1.0f / sqrtf(var + eps);
    }
}

void wave_stats_launch(const float* pcm, const long* pcm_off, const int* n_samples, int B, float eps, float* stats, hipStream_t s) {
    if (B <= 0) return;
    hipLaunchKernelGGL(wave_stats_kernel, dim3(B), dim3(1024), 0, s, pcm, pcm_off, n_samples, eps, stats);
}

// ------------------------------------------------------------------------------------------------
// conv layer 0 (C_in = 1): thread = output channel, a workgroup walks FPB consecutive frames of one clip.
// LayerNorm over the C channels of a frame through a block reduction, exact GELU, bf16 store.
// ------------------------------------------------------------------------------------------------
constexpr int CONV0_FPB = 16;

__global__ void w2v_conv0_kernel(const float* __restrict__ pcm, const long* __restrict__ pcm_off, const float* __restrict__ stats,
                                 const int* __restrict__ frame_off, const int* __restrict__ n_out, const float* __restrict__ w,
                                 const float* __restrict__ bias, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                 float eps, bf16_t* __restrict__ out, int C) {
    extern __shared__ float sm[];                   // [FPB * 5 + 5] normalised samples | [2][waves] reduction
    const int b = blockIdx.y, f0 = blockIdx.x * CONV0_FPB, nf = n_out[b];
    if (f0 >= nf) return;
    const int c = threadIdx.x, nwaves = (blockDim.x + 63) >> 6;
    const int fcount = nf - f0 < CONV0_FPB ? nf - f0 : CONV0_FPB;
    const int ns = fcount * 5 + 5;
    float* s_x = sm;
    float* s_r = sm + CONV0_FPB * 5 + 8;
    const float mean = stats[2 * b], inv = stats[2 * b + 1];
    const float* x = pcm + pcm_off[b] + (long)f0 * 5;
    for (int i = threadIdx.x; i < ns; i += blockDim.x) s_x[i] = (x[i] - mean) * inv;
    float wk[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) wk[k] = c < C ? w[c * 10 + k] : 0.0f;
    const float bc = c < C ? bias[c] : 0.0f, g = c < C ? ln_g[c] : 0.0f, be = c < C ? ln_b[c] : 0.0f;
    __syncthreads();
    for (int f = 0; f < fcount; ++f) {
        float v = bc;
#pragma unroll
        for (int k = 0; k < 10; ++k) v = fmaf(wk[k], s_x[f * 5 + k], v);
        if (c >= C) v = 0.0f;
        float s = wave_sum(v), ss = wave_sum(v * v);
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        __syncthreads();
        if (lane == 0) { s_r[wave] = s; s_r[16 + wave] = ss; }
        __syncthreads();
        float ts = 0.0f, tss = 0.0f;
        for (int q = 0; q < nwaves; ++q) { ts += s_r[q]; tss += s_r[16 + q]; }
        const float mu = ts / (float)C;
        const float var = fmaxf(tss / (float)C - mu * mu, 0.0f);
        const float y = (v - mu) * rsqrtf(var + eps) * g + be;
        if (c < C) out[((long)frame_off[b] + f0 + f) * C + c] = f32_to_bf16(gelu_erf(y));
    }
}

void w2v_conv0_launch(const float* pcm, const long* pcm_off, const float* stats, const int* frame_off, const int* n_out, int B,
                      int max_out, const float* w, const float* bias, const float* ln_g, const float* ln_b, float eps, bf16_t* out,
                      int C, hipStream_t s) {
    if (B <= 0 || max_out <= 0) return;
    if (C > 1024) throw std::invalid_argument("conv0: at most 1024 channels");
    const int threads = ((C + 63) / 64) * 64;
    const size_t lds = (size_t)(CONV0_FPB * 5 + 8 + 32) * sizeof(float);
    hipLaunchKernelGGL(w2v_conv0_kernel, dim3(cdiv(max_out, CONV0_FPB), B), dim3(threads), lds, s, pcm, pcm_off, stats, frame_off,
                       n_out, w, bias, ln_g, ln_b, eps, out, C);
}

__global__ void w2v_conv_rows_kernel(const int* __restrict__ in_off, const int* __restrict__ out_off, const int* __restrict__ n_out,
                                     int B, int total_out, int stride, int C, long* __restrict__ row_off) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= total_out) return;
    int b = 0;
    while (b + 1 < B && m >= out_off[b + 1]) ++b;        // clips are few (<= 64): linear search
    const int t = m - out_off[b];
    row_off[m] = t < n_out[b] ? ((long)in_off[b] + (long)stride * t) * C : 0;
}

void w2v_conv_rows_launch(const int* in_off, const int* out_off, const int* n_out, int B, int total_out, int stride, int C,
                          long* row_off, hipStream_t s) {
    if (total_out <= 0) return;
    hipLaunchKernelGGL(w2v_conv_rows_kernel, dim3(cdiv(total_out, 256)), dim3(256), 0, s, in_off, out_off, n_out, B, total_out,
                       stride, C, row_off);
}

__global__ void w2v_frame_info_kernel(const int* __restrict__ frame_off, const int* __restrict__ n_frames, int B, int total,
                                      int2* __restrict__ info) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= total) return;
    int b = 0;
    while (b + 1 < B && m >= frame_off[b + 1]) ++b;
    info[m] = make_int2(m - frame_off[b], n_frames[b]);
}

void w2v_frame_info_launch(const int* frame_off, const int* n_frames, int B, int total, int2* info, hipStream_t s) {
    if (total <= 0) return;
    hipLaunchKernelGGL(w2v_frame_info_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, frame_off, n_frames, B, total, info);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over rows of f32 with f32 affine parameters -> bf16 (optionally through exact GELU): one wave per row
// ------------------------------------------------------------------------------------------------
constexpr int LNF_MAXV = 8;      // float4 per lane: D <= 2048

template <int ACT, typename OUT>
__global__ __launch_bounds__(256) void layernorm_f32p_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, OUT* __restrict__ y, int rows,
                                                             int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int nv = D / 4;
    const float4* xr = reinterpret_cast<const float4*>(x + (long)row * D);
    float4 v[LNF_MAXV];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < LNF_MAXV; ++i) {
        const int idx = lane + 64 * i;
        v[i] = idx < nv ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < LNF_MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            ss += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < LNF_MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[idx], b = reinterpret_cast<const float4*>(beta)[idx];
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x;
            o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z;
            o.w = (v[i].w - mean) * rstd * g.w + b.w;
            if (ACT == 1) { o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w); }
            if constexpr (sizeof(OUT) == 4) *reinterpret_cast<float4*>(y + (long)row * D + idx * 4) = o;
            else *reinterpret_cast<uint2*>(y + (long)row * D + idx * 4) = pack_bf16x4(o);
        }
    }
}

void layernorm_gelu_f32_launch(const float* x, const float* gamma, const float* beta, float* y, int rows, int D, float eps,
                               hipStream_t s) {
    if (rows <= 0) return;
    if (D % 4 != 0 || D > 256 * LNF_MAXV) throw std::invalid_argument("layernorm: unsupported width");
    hipLaunchKernelGGL((layernorm_f32p_kernel<1, float>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, gamma, beta, y, rows, D, eps);
}

void layernorm_f32p_launch(const float* x, const float* gamma, const float* beta, bf16_t* y, int rows, int D, float eps, int act,
                           hipStream_t s) {
    if (rows <= 0) return;
    if (D % 4 != 0 || D > 256 * LNF_MAXV) throw std::invalid_argument("layernorm: unsupported width");
    if (act) hipLaunchKernelGGL((layernorm_f32p_kernel<1, bf16_t>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, gamma, beta, y, rows, D, eps);
    else hipLaunchKernelGGL((layernorm_f32p_kernel<0, bf16_t>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, gamma, beta, y, rows, D, eps);
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long n4) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) reinterpret_cast<uint2*>(y)[i] = pack_bf16x4(reinterpret_cast<const float4*>(x)[i]);
}

void cast_f32_bf16_launch(const float* x, bf16_t* y, long n, hipStream_t s) {
    if (n <= 0) return;
    if (n % 4) throw std::invalid_argument("cast: element count must be a multiple of 4");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, s, x, y, n / 4);
}

// ------------------------------------------------------------------------------------------------
// Full multi-head attention, flash style.  Workgroup = 64 query rows of one (clip, head): 4 waves x 16 rows.
// Per 64-key tile: K rows and a transposed V image go to LDS once for the four waves; S = Q K^T as 16x16x32 MFMAs with the
// query fragments held in registers; online softmax on the accumulator layout (a lane holds query rows 4 (lane >> 4) + j,
// key column lane & 15 of each 16-key tile: row statistics = four DPP / shuffle steps over the 16 lanes); P (bf16) through a
// wave-private LDS image to become the A operand of O += P V.
// ------------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void mha_attention_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ cu, int D,
                                                            bf16_t* __restrict__ out, float scale) {
    constexpr int KT = 64, KS = HD / 32, DT = HD / 16, LDK = HD + 8, LDV = KT + 8, LDP = KT + 8;
    __shared__ __attribute__((aligned(16))) bf16_t s_k[KT][LDK];
    __shared__ __attribute__((aligned(16))) bf16_t s_vt[HD][LDV];
    __shared__ __attribute__((aligned(16))) bf16_t s_p[4][16][LDP];
    const int clip = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
    const int r0 = cu[clip], L = cu[clip + 1] - r0;
    if (q0 >= L) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fc = lane >> 4;
    const long ld = 3L * D;
    const bf16_t* qb = qkv + (long)r0 * ld + h * HD;
    const bf16_t* kb = qb + D;
    const bf16_t* vb = qb + 2 * D;
    // query fragments of this wave's 16 rows (A operand: row fr, k = 8 fc + e of each 32-wide k-step)
    mfma_bf16x8 qf[KS];
    const int qrow = q0 + wave * 16 + fr;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (qrow < L) u = *reinterpret_cast<const uint4*>(qb + (long)qrow * ld + s * 32 + fc * 8);
        qf[s] = __builtin_bit_cast(mfma_bf16x8, u);
    }
    f32x4 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, l_run[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int CH = HD / 8;
    for (int k0 = 0; k0 < L; k0 += KT) {
        __syncthreads();                                    // previous tile fully consumed
        for (int i = tid; i < KT * CH; i += 256) {
            const int key = i / CH, ch = i - key * CH;
            uint4 uk = make_uint4(0, 0, 0, 0), uv = make_uint4(0, 0, 0, 0);
            if (k0 + key < L) {
                uk = *reinterpret_cast<const uint4*>(kb + (long)(k0 + key) * ld + ch * 8);
                uv = *reinterpret_cast<const uint4*>(vb + (long)(k0 + key) * ld + ch * 8);
            }
            *reinterpret_cast<uint4*>(&s_k[key][ch * 8]) = uk;
            const bf16_t* e = reinterpret_cast<const bf16_t*>(&uv);
#pragma unroll
            for (int j = 0; j < 8; ++j) s_vt[ch * 8 + j][key] = e[j];
        }
        __syncthreads();
        // S tile: 4 key sub-tiles of 16
        f32x4 sc[KT / 16];
#pragma unroll
        for (int kt = 0; kt < KT / 16; ++kt) {
            sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const mfma_bf16x8 kf = *reinterpret_cast<const mfma_bf16x8*>(&s_k[kt * 16 + fr][s * 32 + fc * 8]);
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[s], kf, sc[kt], 0, 0, 0);
            }
        }
        // online softmax: lane holds rows fc*4 + j, key column fr of each sub-tile
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kt = 0; kt < KT / 16; ++kt) {
            const bool valid = k0 + kt * 16 + fr < L;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = valid ? sc[kt][j] * scale : -INFINITY;
                sc[kt][j] = v;
                mx[j] = fmaxf(mx[j], v);
            }
        }
        float alpha[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mx[j] = lane_max16(mx[j]);
            const float mn = fmaxf(m_run[j], mx[j]);          // finite: every tile below L holds a valid key
            alpha[j] = __expf(m_run[j] - mn);
            m_run[j] = mn;
        }
        float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT / 16; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16_t pb = f32_to_bf16(__expf(sc[kt][j] - m_run[j]));
                rs[j] += bf16_to_f32(pb);
                s_p[wave][fc * 4 + j][kt * 16 + fr] = pb;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rs[j] = lane_sum<16>(rs[j]);
            l_run[j] = l_run[j] * alpha[j] + rs[j];
        }
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) o[d][j] *= alpha[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                      // the wave's own P image is complete
        // O += P V: A = P rows (this wave's 16 queries), B = V^T image
#pragma unroll
        for (int ks = 0; ks < KT / 32; ++ks) {
            const mfma_bf16x8 pf = *reinterpret_cast<const mfma_bf16x8*>(&s_p[wave][fr][ks * 32 + fc * 8]);
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                const mfma_bf16x8 vf = *reinterpret_cast<const mfma_bf16x8*>(&s_vt[d * 16 + fr][ks * 32 + fc * 8]);
                o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o[d], 0, 0, 0);
            }
        }
    }
    // o[d][j] = O[row fc*4 + j][d*16 + fr]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = q0 + wave * 16 + fc * 4 + j;
        if (row < L) {
            const float inv = 1.0f / l_run[j];
#pragma unroll
            for (int d = 0; d < DT; ++d) out[((long)r0 + row) * D + h * HD + d * 16 + fr] = f32_to_bf16(o[d][j] * inv);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// head_dim 64 (every published Omnilingual variant): the same online softmax on 32x32x16 MFMAs with the score tile
// transposed, so that nothing but K and V tiles ever goes through LDS.
//   workgroup = 128 queries of one (clip, head): 4 waves x 32 queries, the query fragments in registers as the B operand;
//   S^T = K Q^T  per 32-key block (A = K rows from an XOR-swizzled LDS image): a lane holds, for ITS query (lane & 31),
//                the scores of keys (i & 3) + 8 (i >> 2) + 4 (lane >> 5) in accumulator register i -- the row statistics are
//                in-lane maxima / sums plus one v_permlane32_swap between the two lane halves;
//   O^T += V^T P^T: registers 8 s .. 8 s + 7 of the exponentiated block, packed to bf16, ARE the B fragment of k-step s
//                (cdna_hip_programming.md section 3, "an accumulator tile as the next MFMA's operand"); the matching A fragment
//                = V^T rows in the same permuted key order comes from the row-major V tile by ds_read_b64_tr_b16;
//   K / V tiles of 64 keys are double-buffered in LDS (loads of tile t + 1 issued before the MFMAs of tile t, written
//   after them: one barrier per tile); O is rescaled only in tiles where some row maximum grew (exact, wave-uniform).
// Row sums add the unrounded f32 probabilities (the 16x16 kernel above adds the bf16-rounded ones): |delta| ~ 1e-4 relative.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short tr_b16x4;

__device__ __forceinline__ uint2 lds_read_tr16(const bf16_t* p) {
    const tr_b16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_b16x4*)(p));
    return __builtin_bit_cast(uint2, r);
}

__device__ __forceinline__ float half_swap_max(float x) {       // max over lanes l and l ^ 32
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_swap_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__global__ __launch_bounds__(256) void mha64_attention_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ cu, int D,
                                                              bf16_t* __restrict__ out, float c_exp /* scale * log2(e) */) {
    constexpr int HD = 64, KT = 64, QW = 32, NW = 4;
    __shared__ __attribute__((aligned(16))) bf16_t s_k[2][KT * HD];     // row = key (128 B), 16-byte chunk ch at ch ^ ((key >> 1) & 7)
    __shared__ __attribute__((aligned(16))) bf16_t s_v[2][KT * HD];     // row = key, chunk ch at ch ^ (((key >> 1) & 1) << 2)
    const int clip = blockIdx.z, hq = blockIdx.y, q0 = blockIdx.x * (QW * NW);
    const int r0 = cu[clip], L = cu[clip + 1] - r0;
    if (q0 >= L) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const long ld = 3L * D;
    const bf16_t* qb = qkv + (long)r0 * ld + hq * HD;
    const bf16_t* kb = qb + D;
    const bf16_t* vb = qb + 2 * D;
    // staging map: 512 16-byte chunks per tile and operand, two per thread
    const int skey = tid >> 3, sch = tid & 7;                            // keys skey and skey + 32
    uint4 kst[2], vst[2];
    auto stage_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = k0 + skey + 32 * i;
            kst[i] = make_uint4(0, 0, 0, 0);
            vst[i] = make_uint4(0, 0, 0, 0);
            if (key < L) {
                kst[i] = *reinterpret_cast<const uint4*>(kb + (long)key * ld + sch * 8);
                vst[i] = *reinterpret_cast<const uint4*>(vb + (long)key * ld + sch * 8);
            }
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int key = skey + 32 * i;
            *reinterpret_cast<uint4*>(&s_k[buf][key * HD + ((sch ^ ((key >> 1) & 7)) << 3)]) = kst[i];
            *reinterpret_cast<uint4*>(&s_v[buf][key * HD + ((sch ^ (((key >> 1) & 1) << 2)) << 3)]) = vst[i];
        }
    };
    stage_load(0);
    // query fragments (B operand): column = query r, k = 16 s + 8 h + e
    mfma_bf16x8 qf[4];
    const int qrow = q0 + wave * QW + r;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        uint4 u = make_uint4(0, 0, 0, 0);
        if (qrow < L) u = *reinterpret_cast<const uint4*>(qb + (long)qrow * ld + s * 16 + h * 8);
        qf[s] = __builtin_bit_cast(mfma_bf16x8, u);
    }
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                               // m in raw score units; l = this lane's share of the row sum
    stage_write(0);
    __syncthreads();
    // lane-constant LDS offsets (elements)
    const int g = lane >> 4, li = lane & 15;
    const int v_row = 4 * (g >> 1) + (li >> 2);                          // + 32 kb + 16 s2 (+ 8): key row this lane addresses
    const int v_ch = 2 * (g & 1) + ((li & 3) >> 1), v_in = (li & 1) * 4;  // + 4 dt: chunk, element inside the chunk
    const int ntiles = (L + KT - 1) / KT;
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1, k0 = t * KT;
        if (t + 1 < ntiles) stage_load(k0 + KT);
        const bf16_t* sk = s_k[buf];
        const bf16_t* sv = s_v[buf];
        f32x16 sc[2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sc[kbk][i] = 0.f;
            const int key = kbk * 32 + r;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const mfma_bf16x8 kf =
                    *reinterpret_cast<const mfma_bf16x8*>(&sk[key * HD + (((2 * s + h) ^ ((key >> 1) & 7)) << 3)]);
                sc[kbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], sc[kbk], 0, 0, 0);
            }
        }
        if (k0 + KT > L) {                                               // last tile: keys past the clip (wave-uniform branch)
#pragma unroll
            for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (k0 + kbk * 32 + (i & 3) + 8 * (i >> 2) + 4 * h >= L) sc[kbk][i] = -INFINITY;
        }
        float mx = fmaxf(sc[0][0], sc[1][0]);
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, fmaxf(sc[0][i], sc[1][i]));
        mx = half_swap_max(mx);                                          // finite: key k0 < L is in every tile
        if (__any(mx > m_run)) {
            const float mn = fmaxf(m_run, mx);
            const float alpha = exp2f((m_run - mn) * c_exp);
            m_run = mn;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
        }
        const float mc = m_run * c_exp;
        mfma_bf16x8 pf[2][2];
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 u;
                unsigned* w = reinterpret_cast<unsigned*>(&u);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p0 = exp2f(fmaf(sc[kbk][8 * s2 + 2 * e], c_exp, -mc));
                    const float p1 = exp2f(fmaf(sc[kbk][8 * s2 + 2 * e + 1], c_exp, -mc));
                    l_run += p0 + p1;
                    w[e] = (unsigned)f32_to_bf16(p0) | ((unsigned)f32_to_bf16(p1) << 16);
                }
                pf[kbk][s2] = __builtin_bit_cast(mfma_bf16x8, u);
            }
#pragma unroll
        for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int ka = kbk * 32 + s2 * 16 + v_row, kc = ka + 8;
                    const uint2 lo = lds_read_tr16(&sv[ka * HD + (((4 * dt + v_ch) ^ (((ka >> 1) & 1) << 2)) << 3) + v_in]);
                    const uint2 hi = lds_read_tr16(&sv[kc * HD + (((4 * dt + v_ch) ^ (((kc >> 1) & 1) << 2)) << 3) + v_in]);
                    const mfma_bf16x8 vf = __builtin_bit_cast(mfma_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kbk][s2], o[dt], 0, 0, 0);
                }
        if (t + 1 < ntiles) stage_write(buf ^ 1);
        __syncthreads();
    }
    // o[dt][i] = O[query r][dt * 32 + (i & 3) + 8 (i >> 2) + 4 h]
    const float inv = 1.0f / half_swap_sum(l_run);
    if (qrow < L) {
        bf16_t* dst = out + ((long)r0 + qrow) * D + hq * HD + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 v = make_float4(o[dt][4 * q4] * inv, o[dt][4 * q4 + 1] * inv, o[dt][4 * q4 + 2] * inv, o[dt][4 * q4 + 3] * inv);
                *reinterpret_cast<uint2*>(dst + dt * 32 + q4 * 8) = pack_bf16x4(v);
            }
    }
}

void mha_attention_launch(const bf16_t* qkv, const int* cu, int n_clips, int max_len, int heads, int head_dim, bf16_t* out,
                          hipStream_t s) {
    if (n_clips <= 0 || max_len <= 0) return;
    const int D = heads * head_dim;
    const float scale = 1.0f / sqrtf((float)head_dim);
    if (head_dim == 64 && tuning().mha_form == 1) {
        hipLaunchKernelGGL(mha64_attention_kernel, dim3(cdiv(max_len, 128), heads, n_clips), dim3(256), 0, s, qkv, cu, D, out,
                           scale * 1.4426950408889634f);
        return;
    }
    dim3 grid(cdiv(max_len, 64), heads, n_clips);
    if (head_dim == 64) hipLaunchKernelGGL(mha_attention_kernel<64>, grid, dim3(256), 0, s, qkv, cu, D, out, scale);
    else if (head_dim == 32) hipLaunchKernelGGL(mha_attention_kernel<32>, grid, dim3(256), 0, s, qkv, cu, D, out, scale);
    else throw std::invalid_argument("attention: head_dim must be 32 or 64");
}

__global__ __launch_bounds__(256) void argmax_f32_kernel(const float* __restrict__ x, long ld, int n, int* __restrict__ ids) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const float* row = x + (long)blockIdx.x * ld;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = row[i];
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    s_v[threadIdx.x] = bv;
    s_i[threadIdx.x] = bi;
    __syncthreads();
    for (int ofs = 128; ofs > 0; ofs >>= 1) {
        if ((int)threadIdx.x < ofs) {
            const float ov = s_v[threadIdx.x + ofs];
            const int oi = s_i[threadIdx.x + ofs];
            if (ov > s_v[threadIdx.x] || (ov == s_v[threadIdx.x] && oi < s_i[threadIdx.x])) { s_v[threadIdx.x] = ov; s_i[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) ids[blockIdx.x] = s_i[0] == 0x7fffffff ? 0 : s_i[0];      // all-NaN row: index 0, like the reference's loop
}

void argmax_f32_launch(const float* x, long ld, int rows, int n, int* ids, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(argmax_f32_kernel, dim3(rows), dim3(256), 0, s, x, ld, n, ids);
}

}  // namespace qasr

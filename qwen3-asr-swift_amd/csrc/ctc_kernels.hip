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

// The published geometry (C = 512): a WAVE owns whole frames, a lane 8 consecutive channels (80 weights in registers), so
// the LayerNorm statistics are two DPP wave sums and nothing synchronises across waves; the samples of a frame are
// wave-uniform (scalar loads), one 16-byte store per lane and frame.  Vector-ALU-bound by the GELU (16 instructions per
// element, 1.6 G elements at 32 x 30 s).
constexpr int CONV0W_FPW = 16, CONV0W_WAVES = 4;

__global__ __launch_bounds__(CONV0W_WAVES * 64) void w2v_conv0_wave_kernel(
    const float* __restrict__ pcm, const long* __restrict__ pcm_off, const float* __restrict__ stats, const int* __restrict__ frame_off,
    const int* __restrict__ n_out, const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ ln_g,
    const float* __restrict__ ln_b, float eps, bf16_t* __restrict__ out) {
    constexpr int C = 512, CPL = 8;
    const int b = blockIdx.y, nf = n_out[b];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int f0 = (blockIdx.x * CONV0W_WAVES + wave) * CONV0W_FPW;
    if (f0 >= nf) return;
    const int f1 = f0 + CONV0W_FPW < nf ? f0 + CONV0W_FPW : nf;
    float wk[CPL][10], bc[CPL], g[CPL], be[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int c = lane * CPL + j;
#pragma unroll
        for (int k = 0; k < 10; ++k) wk[j][k] = w[c * 10 + k];
        bc[j] = bias[c]; g[j] = ln_g[c]; be[j] = ln_b[c];
    }
    const float mean = stats[2 * b], inv = stats[2 * b + 1];
    const float* x = pcm + pcm_off[b];
    bf16_t* dst = out + ((long)frame_off[b] + f0) * C + lane * CPL;
    float xs[10];
#pragma unroll
    for (int k = 0; k < 5; ++k) xs[5 + k] = (x[(long)f0 * 5 + k] - mean) * inv;
    for (int f = f0; f < f1; ++f, dst += C) {
#pragma unroll
        for (int k = 0; k < 5; ++k) { xs[k] = xs[5 + k]; xs[5 + k] = (x[(long)f * 5 + 5 + k] - mean) * inv; }
        float v[CPL], s = 0.f, ss = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            float a = bc[j];
#pragma unroll
            for (int k = 0; k < 10; ++k) a = fmaf(wk[j][k], xs[k], a);
            v[j] = a;
            s += a;
            ss = fmaf(a, a, ss);
        }
        s = lane_sum<64>(s);
        ss = lane_sum<64>(ss);
        const float mu = s * (1.0f / C);
        const float rstd = rsqrtf(fmaxf(ss * (1.0f / C) - mu * mu, 0.0f) + eps);
        float y[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) y[j] = gelu_erf((v[j] - mu) * rstd * g[j] + be[j]);
        uint4 o;
        const uint2 lo = pack_bf16x4(make_float4(y[0], y[1], y[2], y[3])), hi = pack_bf16x4(make_float4(y[4], y[5], y[6], y[7]));
        o.x = lo.x; o.y = lo.y; o.z = hi.x; o.w = hi.y;
        *reinterpret_cast<uint4*>(dst) = o;
    }
}

void w2v_conv0_launch(const float* pcm, const long* pcm_off, const float* stats, const int* frame_off, const int* n_out, int B,
                      int max_out, const float* w, const float* bias, const float* ln_g, const float* ln_b, float eps, bf16_t* out,
                      int C, hipStream_t s) {
    if (B <= 0 || max_out <= 0) return;
    if (C == 512) {
        hipLaunchKernelGGL(w2v_conv0_wave_kernel, dim3(cdiv(max_out, CONV0W_FPW * CONV0W_WAVES), B), dim3(CONV0W_WAVES * 64), 0, s, pcm,
                           pcm_off, stats, frame_off, n_out, w, bias, ln_g, ln_b, eps, out);
        return;
    }
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
// LayerNorm over rows of f32 with f32 affine parameters -> bf16 / f32 (optionally through GELU).  A wave owns RPW consecutive
// rows and requests all of them before it reduces the first (VPR float4 per lane and row, RPW * VPR = 8 in registers: 4 rows at
// D <= 512, 2 at <= 1024, 1 at <= 2048); the row statistics are DPP / permlane wave sums, not ds_bpermute shuffles.  Two-pass
// variance (sum of squared deviations), f32 throughout, like the reference's MLXNN.LayerNorm on f32.
// Generic fallback (one row per wave, guards per float4): D not a multiple of 256.
// ------------------------------------------------------------------------------------------------
constexpr int LNF_MAXV = 8;      // float4 per lane: D <= 2048

template <int ACT, typename OUT>
__device__ __forceinline__ void lnf_store(OUT* dst, float4 o) {
    if (ACT == 1) { o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w); }
    if constexpr (sizeof(OUT) == 4) *reinterpret_cast<float4*>(dst) = o;
    else *reinterpret_cast<uint2*>(dst) = pack_bf16x4(o);
}

template <int ACT, typename OUT, int VPR>
__global__ __launch_bounds__(256) void layernorm_f32p_rows_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, OUT* __restrict__ y, int rows,
                                                                  float eps) {
    constexpr int RPW = LNF_MAXV / VPR, D = 256 * VPR;
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    float4 v[RPW][VPR];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r < rows ? row0 + r : rows - 1;             // a clamped duplicate, never stored
        const float4* xr = reinterpret_cast<const float4*>(x + (long)row * D);
#pragma unroll
        for (int i = 0; i < VPR; ++i) v[r][i] = xr[lane + 64 * i];
    }
    float4 g[VPR], b[VPR];
#pragma unroll
    for (int i = 0; i < VPR; ++i) {
        g[i] = reinterpret_cast<const float4*>(gamma)[lane + 64 * i];
        b[i] = reinterpret_cast<const float4*>(beta)[lane + 64 * i];
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < VPR; ++i) s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
        const float mean = lane_sum<64>(s) * (1.0f / D);
        float ss = 0.0f;
#pragma unroll
        for (int i = 0; i < VPR; ++i) {
            const float a = v[r][i].x - mean, c = v[r][i].y - mean, d = v[r][i].z - mean, e = v[r][i].w - mean;
            ss += (a * a + c * c) + (d * d + e * e);
        }
        const float rstd = rsqrtf(lane_sum<64>(ss) * (1.0f / D) + eps);
        if (row0 + r < rows) {
#pragma unroll
            for (int i = 0; i < VPR; ++i) {
                float4 o;
                o.x = (v[r][i].x - mean) * rstd * g[i].x + b[i].x;
                o.y = (v[r][i].y - mean) * rstd * g[i].y + b[i].y;
                o.z = (v[r][i].z - mean) * rstd * g[i].z + b[i].z;
                o.w = (v[r][i].w - mean) * rstd * g[i].w + b[i].w;
                lnf_store<ACT, OUT>(y + (long)(row0 + r) * D + (lane + 64 * i) * 4, o);
            }
        }
    }
}

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
    const float mean = lane_sum<64>(s) / (float)D;
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < LNF_MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            ss += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = rsqrtf(lane_sum<64>(ss) / (float)D + eps);
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
            lnf_store<ACT, OUT>(y + (long)row * D + idx * 4, o);
        }
    }
}

template <int ACT, typename OUT>
static void layernorm_f32p_go(const float* x, const float* gamma, const float* beta, OUT* y, int rows, int D, float eps, hipStream_t s) {
    if (D % 4 != 0 || D > 256 * LNF_MAXV) throw std::invalid_argument("layernorm: unsupported width");
#define QASR_LNF(VPR_)                                                                                                         \
    hipLaunchKernelGGL((layernorm_f32p_rows_kernel<ACT, OUT, VPR_>), dim3(cdiv(rows, 4 * (LNF_MAXV / VPR_))), dim3(256), 0, s, x, gamma, \
                       beta, y, rows, eps)
    if (D == 512) QASR_LNF(2);
    else if (D == 1024) QASR_LNF(4);
    else if (D == 2048) QASR_LNF(8);
    else if (D == 256) QASR_LNF(1);
    else hipLaunchKernelGGL((layernorm_f32p_kernel<ACT, OUT>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, gamma, beta, y, rows, D, eps);
#undef QASR_LNF
}

void layernorm_gelu_f32_launch(const float* x, const float* gamma, const float* beta, float* y, int rows, int D, float eps,
                               hipStream_t s) {
    if (rows <= 0) return;
    layernorm_f32p_go<1, float>(x, gamma, beta, y, rows, D, eps, s);
}

void layernorm_f32p_launch(const float* x, const float* gamma, const float* beta, bf16_t* y, int rows, int D, float eps, int act,
                           hipStream_t s) {
    if (rows <= 0) return;
    if (act) layernorm_f32p_go<1, bf16_t>(x, gamma, beta, y, rows, D, eps, s);
    else layernorm_f32p_go<0, bf16_t>(x, gamma, beta, y, rows, D, eps, s);
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

__global__ __launch_bounds__(256) void argmax_f32_kernel(const float* __restrict__ x, long ld, int n, int* __restrict__ ids,
                                                         int* __restrict__ err) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const float* row = x + (long)blockIdx.x * ld;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    bool bad = false;
    auto take = [&](float v, int i) {
        bad |= !(fabsf(v) <= 3.0e38f);                     // NaN or infinity: a broken checkpoint, not a transcript
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    };
    if ((n & 3) == 0 && (ld & 3) == 0) {                    // 16-byte loads (every published vocabulary: 10288)
        for (int i = threadIdx.x * 4; i < n; i += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(row + i);
            take(v.x, i); take(v.y, i + 1); take(v.z, i + 2); take(v.w, i + 3);
        }
    } else {
        for (int i = threadIdx.x; i < n; i += 256) take(row[i], i);
    }
    if (bad && err) atomicOr(err, 1);
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

void argmax_f32_launch(const float* x, long ld, int rows, int n, int* ids, int* err, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(argmax_f32_kernel, dim3(rows), dim3(256), 0, s, x, ld, n, ids, err);
}

}  // namespace qasr

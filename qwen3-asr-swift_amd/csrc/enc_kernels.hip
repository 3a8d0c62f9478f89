// enc_kernels.hip -- conv2d1, LayerNorm, window attention (see enc_kernels.h).
#include "enc_kernels.h"

namespace qasr {

// ------------------------------------------------------------------------------------------------
// conv2d1: one workgroup per (output row oh, image).  C_in = 1 so K = 9: pure VALU f32; the cost is
// the NHWC bf16 store (C*2 bytes contiguous per pixel, 16 bytes per lane).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ mel, int mel_stride, int n_mels,
                                                    const ChunkMeta* __restrict__ chunks, const bf16_t* __restrict__ w,
                                                    const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                    int H1, int W1, int C) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* s_in = sm;                 // [3][W_IN + 2], column 0 = iw -1
    const int W_IN = 2 * W1;          // 100
    float* s_w = sm + 3 * (W_IN + 2); // [9][C]
    float* s_b = s_w + 9 * C;         // [C]
    const int oh = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const ChunkMeta cm = chunks[img];
    const float* src = mel + (long)cm.clip * n_mels * mel_stride + cm.t0;
    for (int i = tid; i < 3 * (W_IN + 2); i += 256) {
        int kh = i / (W_IN + 2), col = i - kh * (W_IN + 2);
        int ih = 2 * oh - 1 + kh, iw = col - 1;
        float v = 0.0f;
        if (ih >= 0 && ih < n_mels && iw >= 0 && iw < cm.clen) v = src[(long)ih * mel_stride + iw];
        s_in[i] = v;
    }
    for (int i = tid; i < 9 * C; i += 256) {
        int t = i / C, c = i - t * C;
        s_w[i] = bf16_to_f32(w[c * 9 + t]);
    }
    for (int i = tid; i < C; i += 256) s_b[i] = bias[i];
    __syncthreads();
    const int P = C / 8, G = 256 / P;
    const int g = tid / P, c0 = (tid - g * P) * 8;
    if (g >= G) return;
    bf16_t* orow = out + (((long)img * H1 + oh) * W1) * C;
    // the thread's 8 channels x 9 taps stay in registers for all of its pixels (they were re-read from LDS per pixel: 18
    // ds_read_b128 beside 72 FMAs)
    float wr[9][8], br[8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 8; ++c) wr[t][c] = s_w[t * C + c0 + c];
#pragma unroll
    for (int c = 0; c < 8; ++c) br[c] = s_b[c0 + c];
    for (int ow = g; ow < W1; ow += G) {
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = br[c];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const float x = s_in[kh * (W_IN + 2) + 2 * ow + kw];
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = fmaf(x, wr[kh * 3 + kw][c], acc[c]);
            }
        uint4 o;
        unsigned pk[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = ow < cm.w1 ? gelu_erf(acc[2 * c]) : 0.0f;
            float b = ow < cm.w1 ? gelu_erf(acc[2 * c + 1]) : 0.0f;
            pk[c] = pack_bf16x2(a, b);
        }
        o.x = pk[0]; o.y = pk[1]; o.z = pk[2]; o.w = pk[3];
        *reinterpret_cast<uint4*>(orow + (long)ow * C + c0) = o;
    }
}

void conv1_launch(const float* mel, int mel_stride, int n_mels, const ChunkMeta* chunks, int n_img, const bf16_t* w,
                  const float* bias, bf16_t* out, int H1, int W1, int C, hipStream_t s) {
    if (n_img <= 0) return;
    size_t sh = (3 * (2 * W1 + 2) + 10 * C) * sizeof(float);
    hipLaunchKernelGGL(conv1_kernel, dim3(H1, n_img), dim3(256), sh, s, mel, mel_stride, n_mels, chunks, w, bias, out,
                       H1, W1, C);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wavefront per row, two-pass in registers (D <= 64 * 4 * LN_MAXV).
// ------------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 5;   // float4 per lane: D <= 1280

__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const bf16_t* __restrict__ gamma,
                                                        const bf16_t* __restrict__ beta, bf16_t* __restrict__ y, int T,
                                                        int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= T) return;
    const int nv = D / 4;
    const float4* xr = reinterpret_cast<const float4*>(x + (long)row * D);
    float4 v[LN_MAXV];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        int idx = lane + 64 * i;
        v[i] = idx < nv ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        int idx = lane + 64 * i;
        if (idx < nv) {
            float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            sq += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        int idx = lane + 64 * i;
        if (idx < nv) {
            float4 g = load_bf16x4(gamma + idx * 4), b = load_bf16x4(beta + idx * 4);
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x;
            o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z;
            o.w = (v[i].w - mean) * rstd * g.w + b.w;
            *reinterpret_cast<uint2*>(y + (long)row * D + idx * 4) = pack_bf16x4(o);
        }
    }
}

void layernorm_launch(const float* x, const bf16_t* gamma, const bf16_t* beta, bf16_t* y, int T, int D, float eps,
                      hipStream_t s) {
    if (T <= 0) return;
    if (D % 4 != 0 || D > 256 * LN_MAXV) throw std::invalid_argument("layernorm: unsupported width");
    hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(T, 4)), dim3(256), 0, s, x, gamma, beta, y, T, D, eps);
}

// ------------------------------------------------------------------------------------------------
// Window attention.  One workgroup per (window, head); window length L <= 128.
// S = Q K^T on MFMA with Q/K fragments read straight from the packed qkv rows (each 16-byte
// fragment chunk is 8 consecutive head dims of one token), softmax in registers (rows live on 16
// lanes), P (bf16) goes through a wave-private LDS image to become the A operand of P V, whose
// B operand is a transposed V image (keys contiguous) built once per workgroup.
// ------------------------------------------------------------------------------------------------
constexpr int WA_MAXL = 128, WA_LD = WA_MAXL + 8;

template <int HD>
__global__ __launch_bounds__(256) void window_attention_kernel(const bf16_t* __restrict__ qkv,
                                                               const int* __restrict__ cu, int D,
                                                               bf16_t* __restrict__ out, float scale) {
    __shared__ __attribute__((aligned(16))) bf16_t s_vt[HD][WA_LD];
    __shared__ __attribute__((aligned(16))) bf16_t s_p[4][16][WA_LD];
    const int w = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s0 = cu[w], L = cu[w + 1] - s0;
    const long ld = 3L * D;
    const bf16_t* qb = qkv + (long)s0 * ld + h * HD;
    const bf16_t* kb = qb + D;
    const bf16_t* vb = qb + 2 * D;
    // V^T image (zero beyond L)
    constexpr int CH = HD / 8;
    for (int i = tid; i < WA_MAXL * CH; i += 256) {
        int key = i / CH, ch = i - key * CH;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (key < L) u = *reinterpret_cast<const uint4*>(vb + (long)key * ld + ch * 8);
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&u);
#pragma unroll
        for (int j = 0; j < 8; ++j) s_vt[ch * 8 + j][key] = e[j];
    }
    __syncthreads();
    const int fr = lane & 15, fc = lane >> 4;
    constexpr int KS = HD / 32;        // k-steps of QK^T
    constexpr int DT = HD / 16;        // d tiles of the output
    const int nkt = (L + 15) / 16;
    for (int qt = wave; qt * 16 < L; qt += 4) {
        mfma_bf16x8 a[KS];
        const int qrow = qt * 16 + fr;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            uint4 u = make_uint4(0, 0, 0, 0);
            if (qrow < L) u = *reinterpret_cast<const uint4*>(qb + (long)qrow * ld + s * 32 + fc * 8);
            a[s] = __builtin_bit_cast(mfma_bf16x8, u);
        }
        f32x4 sc[WA_MAXL / 16];
#pragma unroll
        for (int kt = 0; kt < WA_MAXL / 16; ++kt) {
            sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kt < nkt) {
                const int key = kt * 16 + fr;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    uint4 u = make_uint4(0, 0, 0, 0);
                    if (key < L) u = *reinterpret_cast<const uint4*>(kb + (long)key * ld + s * 32 + fc * 8);
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s], __builtin_bit_cast(mfma_bf16x8, u), sc[kt], 0, 0, 0);
                }
            }
        }
        // softmax over keys: this lane holds rows fc*4 + j, column fr of every key tile
        float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int kt = 0; kt < WA_MAXL / 16; ++kt) {
            const bool valid = kt * 16 + fr < L;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = valid ? sc[kt][j] * scale : -INFINITY;
                sc[kt][j] = v;
                mx[j] = fmaxf(mx[j], v);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], o, 64));
        float sm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < WA_MAXL / 16; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float e = expf(sc[kt][j] - mx[j]);      // exp(-inf) = 0 for masked keys
                sc[kt][j] = e;
                sm[j] += e;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sm[j] += __shfl_xor(sm[j], o, 64);
            sm[j] = 1.0f / sm[j];
        }
#pragma unroll
        for (int kt = 0; kt < WA_MAXL / 16; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) s_p[wave][fc * 4 + j][kt * 16 + fr] = f32_to_bf16(sc[kt][j] * sm[j]);
        // wave-private image: LDS ops of one wave complete in order, the fence stops compiler reordering
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4 o[DT];
#pragma unroll
        for (int d = 0; d < DT; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < WA_MAXL / 32; ++ks) {
            if (ks * 32 < L) {
                mfma_bf16x8 pa = *reinterpret_cast<const mfma_bf16x8*>(&s_p[wave][fr][ks * 32 + fc * 8]);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    mfma_bf16x8 vbf = *reinterpret_cast<const mfma_bf16x8*>(&s_vt[d * 16 + fr][ks * 32 + fc * 8]);
                    o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, vbf, o[d], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int row = qt * 16 + fc * 4 + j;
                if (row < L) out[(long)(s0 + row) * D + h * HD + d * 16 + fr] = f32_to_bf16(o[d][j]);
            }
    }
}

void window_attention_launch(const bf16_t* qkv, const int* cu_seqlens, int n_windows, int heads, int head_dim,
                             bf16_t* out, hipStream_t s) {
    if (n_windows <= 0) return;
    const int D = heads * head_dim;
    const float scale = 1.0f / sqrtf((float)head_dim);
    dim3 grid(n_windows, heads);
    if (head_dim == 64)
        hipLaunchKernelGGL(window_attention_kernel<64>, grid, dim3(256), 0, s, qkv, cu_seqlens, D, out, scale);
    else if (head_dim == 32)
        hipLaunchKernelGGL(window_attention_kernel<32>, grid, dim3(256), 0, s, qkv, cu_seqlens, D, out, scale);
    else
        throw std::invalid_argument("window attention: head_dim must be 32 or 64");
}

}  // namespace qasr

// dec_quant.hip -- decode-step kernels on MLX affine-quantised weights (see dec_quant.h).
#include "dec_quant.h"
#include "dec_epilogue.h"
#include "gemm.h"      // lds_ptr_t / glb_ptr_t
#include "tuning.h"
#include "dec_quant_dev.h"
#include <cstdio>

namespace qasr {

// element g of a run of scale (or bias) values held in 16-byte registers v[0 ..]
template <bool F32, int N>
__device__ __forceinline__ float sb_reg(const uint4 (&v)[N], int base, int g) {
    if constexpr (F32) return __uint_as_float(u4_word(v[base + g / 4], g % 4));
    else {
        const unsigned w = u4_word(v[base + g / 8], (g % 8) / 2);
        return __uint_as_float((g & 1) ? (w & 0xffff0000u) : (w << 16));
    }
}

// ------------------------------------------------------------------------------------------------
// image builders
// ------------------------------------------------------------------------------------------------
size_t quant_q_bytes(int N, int K, int bits) { return (size_t)N * K * bits / 8; }
size_t quant_sb_bytes(int N, int K, int sb_f32) { return (size_t)N * (K / 64) * 2 * (sb_f32 ? 4 : 2); }

__global__ void quant_pack_q_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int N, int K, int bits) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;           // one destination uint4 per thread
    const int BLK = bits == 4 ? 128 : 64;
    const long total = (long)N * K / BLK * 4;                                // (N/16) * (K/BLK) * 64
    if (idx >= total) return;
    const int lane = (int)(idx & 63), nblk = K / BLK;
    const long blk = idx >> 6;
    const int kb = (int)(blk % nblk);
    const long tile = blk / nblk;
    const long row = tile * 16 + (lane & 15);
    const int fc = lane >> 4, wpr = K * bits / 32;
    uint4 o;
    unsigned* ow = reinterpret_cast<unsigned*>(&o);
    if (bits == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned w = src[row * wpr + (kb * 4 + i) * 4 + fc];          // checkpoint order: element j in bits [4j, 4j + 4)
            unsigned r = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                       // image order: pair (2k, 2k+1) in nibbles k, k + 4
                r |= ((w >> (8 * k)) & 0xFu) << (4 * k);
                r |= ((w >> (8 * k + 4)) & 0xFu) << (4 * k + 16);
            }
            ow[i] = r;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) ow[2 * i + h] = src[row * wpr + (kb * 2 + i) * 8 + fc * 2 + h];
    }
    reinterpret_cast<uint4*>(dst)[idx] = o;
}

// sb image [tile][row 16][g][which = scale, bias]: element size esz (dec_quant_dev.h sb_load)
__global__ void quant_pack_sb_kernel(const char* __restrict__ scales, const char* __restrict__ biases, char* __restrict__ dst,
                                     int N, int G, int esz) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;           // one element of scales AND of biases
    if (idx >= (long)N * G) return;
    const long row = idx / G, tile = row >> 4;
    const int g = (int)(idx - row * G), r = (int)(row & 15);
    const long d0 = ((tile * 16 + r) * G + g) * 2, d1 = d0 + 1;
    for (int b = 0; b < esz; ++b) { dst[d0 * esz + b] = scales[idx * esz + b]; dst[d1 * esz + b] = biases[idx * esz + b]; }
}

void quant_pack_launch(const QuantRaw& src, uint32_t* qp, void* sb, hipStream_t s) {
    if (src.N % 16 != 0 || src.K % 128 != 0 || (src.bits != 4 && src.bits != 8))
        throw std::invalid_argument("quant pack: N must be a multiple of 16, K of 128, bits 4 or 8");
    const int BLK = src.bits == 4 ? 128 : 64;
    const long total = (long)src.N * src.K / BLK * 4;
    hipLaunchKernelGGL(quant_pack_q_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, src.wq, qp, src.N, src.K, src.bits);
    const int G = src.K / 64;
    hipLaunchKernelGGL(quant_pack_sb_kernel, dim3(cdiv((long)src.N * G, 256)), dim3(256), 0, s,
                       reinterpret_cast<const char*>(src.scales), reinterpret_cast<const char*>(src.biases),
                       reinterpret_cast<char*>(sb), src.N, G, src.sb_f32 ? 4 : 2);
}

// src row r0 + i -> out row (i / block) * block_stride + block_off + i % block (block = nrows, stride = 0: rows back to back)
__global__ void quant_dequant_rows_kernel(QuantRaw q, int r0, bf16_t* __restrict__ out, int block, int block_stride, int block_off) {
    const long row = blockIdx.x;
    const long orow = (row / block) * block_stride + block_off + row % block;
    uint4* dst = reinterpret_cast<uint4*>(out + orow * q.K);
    for (int c = threadIdx.x; c < q.K / 8; c += blockDim.x) dst[c] = quant_dequant_chunk(q, r0 + row, c);
}

void quant_dequant_rows_launch(const QuantRaw& src, int r0, int nrows, bf16_t* out, hipStream_t s, int block, int block_stride, int block_off) {
    if (nrows <= 0) return;
    if (src.K % 64 != 0) throw std::invalid_argument("quantised matrix: K must be a multiple of the group size 64");
    if (block <= 0) { block = nrows; block_stride = 0; block_off = 0; }
    hipLaunchKernelGGL(quant_dequant_rows_kernel, dim3(nrows), dim3(128), 0, s, src, r0, out, block, block_stride, block_off);
}

// Several matrices in ONE launch (a decoder layer's seven for the prompt pass: seven launches of row-sized workgroups cost 46 us per layer,
// four times the time the 40 MB they move need).  A workgroup works inside one matrix (the job index is wave-uniform: its descriptor
// comes through scalar loads from the kernel arguments), 256 threads x 4 chunks of 8 elements.
__global__ __launch_bounds__(256) void quant_dequant_multi_kernel(DequantJobs J) {
    int k = 0;
#pragma unroll
    for (int i = 1; i < DequantJobs::MAX; ++i) k += (i < J.n && (int)blockIdx.x >= J.first_block[i]) ? 1 : 0;
    k = __builtin_amdgcn_readfirstlane(k);
    const DequantJob jb = J.job[k];
    const int cpr = jb.q.K / 8;
    const long total = (long)jb.nrows * cpr;
    const long base = ((long)blockIdx.x - J.first_block[k]) * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long c = base + u * 256;
        if (c < total) {
            const long row = c / cpr;
            const int cc = (int)(c - row * cpr);
            const long orow = (row / jb.block) * jb.block_stride + jb.block_off + row % jb.block;
            reinterpret_cast<uint4*>(jb.out + orow * jb.q.K)[cc] = quant_dequant_chunk(jb.q, jb.r0 + row, cc);
        }
    }
}

void quant_dequant_multi_launch(const DequantJob* jobs, int n, hipStream_t s) {
    if (n <= 0) return;
    if (n > DequantJobs::MAX) throw std::invalid_argument("quant_dequant_multi_launch: too many matrices for one launch");
    DequantJobs J{};
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        DequantJob jb = jobs[i];
        if (jb.q.K % 64 != 0) throw std::invalid_argument("quantised matrix: K must be a multiple of the group size 64");
        if (jb.nrows < 0) throw std::invalid_argument("quant_dequant_multi_launch: negative row count");
        if (jb.block <= 0) { jb.block = jb.nrows > 0 ? jb.nrows : 1; jb.block_stride = 0; jb.block_off = 0; }
        J.job[i] = jb;
        J.first_block[i] = blocks;
        blocks += (int)cdiv((long)jb.nrows * (jb.q.K / 8), 1024);
    }
    J.n = n;
    if (blocks == 0) return;
    hipLaunchKernelGGL(quant_dequant_multi_kernel, dim3(blocks), dim3(256), 0, s, J);
}

__global__ void embed_splice_q_kernel(const int* __restrict__ ids, const int* __restrict__ audio_src, QuantRaw q,
                                      const bf16_t* __restrict__ audio, bf16_t* __restrict__ x, int H) {
    const int p = blockIdx.x, a = audio_src ? audio_src[p] : -1;
    uint4* dst = reinterpret_cast<uint4*>(x + (long)p * H);
    if (a >= 0) {
        const uint4* src = reinterpret_cast<const uint4*>(audio + (long)a * H);
        for (int i = threadIdx.x; i < H / 8; i += blockDim.x) dst[i] = src[i];
    } else {
        const long row = ids[p];
        for (int i = threadIdx.x; i < H / 8; i += blockDim.x) dst[i] = quant_dequant_chunk(q, row, i);
    }
}

void embed_splice_q_launch(const int* ids, const int* audio_src, const QuantRaw& embed, const bf16_t* audio, bf16_t* x,
                           int n_pos, int H, hipStream_t s) {
    if (n_pos <= 0) return;
    hipLaunchKernelGGL(embed_splice_q_kernel, dim3(n_pos), dim3(128), 0, s, ids, audio_src, embed, audio, x, H);
}

void gather_rows_q_launch(const QuantRaw& embed, const int* row_idx, bf16_t* dst, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(embed_splice_q_kernel, dim3(n), dim3(128), 0, s, row_idx, (const int*)nullptr, embed, (const bf16_t*)nullptr,
                       dst, embed.K);
}

// ------------------------------------------------------------------------------------------------
// Tuned decode-step kernel.  Same skeleton as decode_gemv2_kernel (dec_gemv.hip): a workgroup = one 16-row weight tile
// group (NT tiles) x 16 batch rows; activation loads first, then ALL packed weights of the wave (a whole matrix is
// 1 - 4 MB at 4 bit: in flight at once), activations staged through LDS with the RMSNorm applied on the way, k-blocks
// interleaved over the waves, fixed-order cross-wave reduction.  Differences:
//   * MFMA operands are swapped (A = activations, B = weights) so a lane owns ONE weight row (n = lane & 15) and four
//     batch rows: one scale / bias per group per lane instead of four;
//   * per 64-column group: acc = sum q x (two 16x16x32 MFMAs from a zero accumulator), then
//     tot += scale * acc + bias * xsum[batch row], xsum = f32 sum of the staged (bf16) activations of that group, built
//     by the staging threads with three shuffles per chunk.
// ------------------------------------------------------------------------------------------------
enum { QPRO_COPY = 0, QPRO_RMSNORM = 1 };

struct DecGemvQArgs {
    DecGemvArgs g;
    const uint32_t* qp;
    const void* sb;
    const bf16_t* norm_w;
    float eps;
};

// KPH > 1 (K = 6144, the 1.7B preset's down-projection: 16 rows x 12 KiB do not fit the LDS): the activation rows pass through the
// same LDS image in KPH column phases of K / KPH columns; all packed weights and all rows are requested up front as before.
// XBAR: as in decode_gemv2_kernel -- the weight requests follow the row requests behind a bare s_barrier instead of behind the rows' return
template <int BITS, bool SBF32, int NT, int WAVES, int KBW, int PRO, int EPI, int KPH = 1, bool XBAR = false>
__global__ __launch_bounds__(WAVES * 64) void decode_gemvq_kernel(DecGemvQArgs a2) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    constexpr int BLK = BITS == 4 ? 128 : 64, GPB = BLK / 64;
    constexpr int KH = KBW * WAVES * BLK;                                      // columns per phase
    constexpr int K = KH * KPH, G = K / 64, GH = KH / 64, KCH = KH / 8, XSTRIDE = 2 * KH + 16;
    constexpr int TPR = WAVES * 64 / 16, XI = KCH / TPR;
    static_assert(TPR % 8 == 0 && TPR <= 64 && KCH % TPR == 0, "row staging geometry");
    static_assert(KPH == 1 || PRO == QPRO_COPY, "column phases: no RMSNorm prologue (the row statistic needs the whole row)");
    DecGemvArgs a = a2.g;
    {
        const int r0 = blockIdx.y * 16;
        a.X += (long)r0 * K;
        a.out += (long)r0 * (EPI == DEC_EPI_SWIGLU ? a.N / 2 : a.N);
        a.B = a.B - r0 < 16 ? a.B - r0 : 16;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fc = lane >> 4;
    const int n0 = blockIdx.x * 16 * NT;
    char* s_x = dsm;                                                          // [16][XSTRIDE] bf16
    float* s_xs = reinterpret_cast<float*>(dsm + 16 * XSTRIDE);               // [GH][16] group sums (of the phase in LDS)
    float* s_red = s_xs + GH * 16;                                            // [WAVES][NT][16 batch][16 n]
    const int srow = tid / TPR, scol = tid % TPR;
    // ---- 1. activation loads (clamped row, zeroed by a select) --------------------------------------------
    uint4 xr[KPH][XI];
    const bf16_t* xp = a.X + (long)(srow < a.B ? srow : 0) * K + scol * 8;
    {
#pragma unroll
        for (int i = 0; i < XI; ++i) xr[0][i] = *reinterpret_cast<const uint4*>(xp + i * TPR * 8);
        if constexpr (XBAR) __builtin_amdgcn_s_barrier();
        else if (srow >= a.B) {
#pragma unroll
            for (int i = 0; i < XI; ++i) xr[0][i] = make_uint4(0, 0, 0, 0);
        }
    }
    // norm weights: requested here, all at once (left inside the staging loop the compiler issues them one by one, each
    // behind a full wait -- XI dependent round trips)
    uint4 nwr[PRO == QPRO_RMSNORM ? XI : 1];
    if constexpr (PRO == QPRO_RMSNORM) {
#pragma unroll
        for (int i = 0; i < XI; ++i) nwr[i] = reinterpret_cast<const uint4*>(a2.norm_w)[scol + i * TPR];
    }
    // ---- 2. every packed weight block of this wave + the scales / biases of its groups --------------------
    // block index of (phase kp, i): kp * (KH / BLK) + wave + WAVES * i, stored at [kp * KBW + i]
    uint4 wq[NT][KBW * KPH];
    float sc[NT][KBW * KPH][GPB], bi[NT][KBW * KPH][GPB];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const long tile = n0 / 16 + t;
        const uint32_t* qp = a2.qp + (tile * (K / BLK) * 64 + lane) * 4;
#pragma unroll
        for (int kp = 0; kp < KPH; ++kp)
#pragma unroll
            for (int i = 0; i < KBW; ++i)
                wq[t][kp * KBW + i] = *reinterpret_cast<const uint4*>(qp + (long)(kp * (KH / BLK) + wave + WAVES * i) * 256);
#pragma unroll
        for (int kp = 0; kp < KPH; ++kp)
#pragma unroll
            for (int i = 0; i < KBW; ++i)
            {
                const int g = (kp * (KH / BLK) + wave + WAVES * i) * GPB;
                sb_load<SBF32, GPB>(a2.sb, ((tile * 16 + fr) * G + g) * 2, sc[t][kp * KBW + i], bi[t][kp * KBW + i]);
#pragma unroll
                for (int h = 0; h < GPB; ++h) bi[t][kp * KBW + i][h] = eff_bias<BITS>(sc[t][kp * KBW + i][h], bi[t][kp * KBW + i][h]);
            }
    }
    // the later phases' activation rows: behind the weight requests, in registers long before their phase
#pragma unroll
    for (int kp = 1; kp < KPH; ++kp)
#pragma unroll
        for (int i = 0; i < XI; ++i) xr[kp][i] = *reinterpret_cast<const uint4*>(xp + kp * KH + i * TPR * 8);
    if constexpr (XBAR) {
        if (srow >= a.B) {
#pragma unroll
            for (int i = 0; i < XI; ++i) xr[0][i] = make_uint4(0, 0, 0, 0);
        }
    }
    // the requests above stay above: left to itself the scheduler sinks them below the RMSNorm reduction (five dependent
    // cross-lane steps), i.e. the weights are asked for a microsecond late
    __builtin_amdgcn_sched_barrier(0);
    // epilogue lane map: batch row = lane >> 2, four consecutive n at (lane & 3) * 4
    const int erow = lane >> 2, eq = lane & 3;
    uint2 rsd[NT][1];
    if constexpr (EPI == DEC_EPI_RESID) {
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                rsd[t][0] = *reinterpret_cast<const uint2*>(a.out + (long)(erow < a.B ? erow : 0) * a.N + n0 + t * 16 + eq * 4);
        }
    }
    // ---- 3. + 4. per column phase: activation rows -> LDS (+ RMSNorm), group sums of the staged values; then per group two MFMAs from
    // zero and scale / bias on the vector unit ---------------------------------------------------------------------------------------
    f32x4 tot[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) tot[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < KPH; ++kp) {
        if (kp > 0) {
            __syncthreads();                                 // the previous phase's LDS reads are done
            if (srow >= a.B) {
#pragma unroll
                for (int i = 0; i < XI; ++i) xr[kp][i] = make_uint4(0, 0, 0, 0);
            }
        }
        {
            char* xrow = s_x + (size_t)srow * XSTRIDE + scol * 16;
            float inv = 0.0f;
            if constexpr (PRO == QPRO_RMSNORM) {
                float ss = 0.0f;
#pragma unroll
                for (int i = 0; i < XI; ++i) {
                    const bf16_t* e = reinterpret_cast<const bf16_t*>(&xr[kp][i]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float f = bf16_to_f32(e[j]); ss = fmaf(f, f, ss); }
                }
                ss = lane_sum<TPR>(ss);
                inv = rsqrtf(ss / (float)K + a2.eps);
            }
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                uint4 o = xr[kp][i];
                if constexpr (PRO == QPRO_RMSNORM) {
                    const uint4 nw = nwr[i];
                    o = make_uint4(rmsnorm_pair_bf16(o.x, nw.x, inv), rmsnorm_pair_bf16(o.y, nw.y, inv),
                                   rmsnorm_pair_bf16(o.z, nw.z, inv), rmsnorm_pair_bf16(o.w, nw.w, inv));
                }
                *reinterpret_cast<uint4*>(xrow + i * TPR * 16) = o;
                const bf16_t* oe = reinterpret_cast<const bf16_t*>(&o);
                float p = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) p += bf16_to_f32(oe[j]);
                // chunk c = scol + TPR * i belongs to group c / 8: the 8 chunks of a group sit on 8 adjacent lanes
                p = lane_sum8(p);
                if ((scol & 7) == 0) s_xs[((scol + TPR * i) >> 3) * 16 + srow] = p;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KBW; ++i) {
            const int blk = wave + WAVES * i;                // block within the phase
#pragma unroll
            for (int h = 0; h < GPB; ++h) {
                const int g = blk * GPB + h;                 // group within the phase
                const f32x4 xs = *reinterpret_cast<const f32x4*>(s_xs + g * 16 + fc * 4);
                const uint4 x0 = *reinterpret_cast<const uint4*>(s_x + (size_t)fr * XSTRIDE + ((g * 2 + 0) * 32 + fc * 8) * 2);
                const uint4 x1 = *reinterpret_cast<const uint4*>(s_x + (size_t)fr * XSTRIDE + ((g * 2 + 1) * 32 + fc * 8) * 2);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, x0), frag_of<BITS>(wq[t][kp * KBW + i], 2 * h + 0), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, x1), frag_of<BITS>(wq[t][kp * KBW + i], 2 * h + 1), acc, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) tot[t][j] += sc[t][kp * KBW + i][h] * acc[j] + bi[t][kp * KBW + i][h] * xs[j];
                }
            }
        }
    }
    // ---- 5. cross-wave reduction in fixed order through a [batch][n] image, epilogue on wave 0 ---------------
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) s_red[((size_t)(wave * NT + t) * 16 + fc * 4 + j) * 16 + fr] = tot[t][j];
    __syncthreads();
    if (wave != 0) return;
    f32x4 acc[NT][1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int wv = 0; wv < WAVES; ++wv)
            acc[t][0] += *reinterpret_cast<const f32x4*>(s_red + ((size_t)(wv * NT + t) * 16 + erow) * 16 + eq * 4);
    }
    if constexpr (EPI == DEC_EPI_RESID) dec_epilogue<NT, 1, EPI>(a, acc, n0, erow, eq, rsd);
    else dec_epilogue<NT, 1, EPI>(a, acc, n0, erow, eq);
}

template <int BITS, int WAVES, int KBW, int NT>
constexpr size_t gemvq_lds() {
    constexpr int K = KBW * WAVES * (BITS == 4 ? 128 : 64);                  // columns per phase
    return (size_t)16 * (2 * K + 16) + (size_t)(K / 64) * 16 * 4 + (size_t)WAVES * NT * 1024;
}

template <int BITS, bool SBF32, int NT, int WAVES, int KBW, int PRO, int EPI, int KPH = 1>
static bool gemvq_go(const DecGemvQArgs& a2, hipStream_t s) {
    constexpr size_t lds = gemvq_lds<BITS, WAVES, KBW, NT>();
    static_assert(lds <= 156 * 1024, "LDS image too large");
    const dim3 grid(a2.g.N / (16 * NT), (a2.g.B + 15) / 16);
    if constexpr (KPH == 1) {
        // the rule of decode_gemv2_kernel (knob gemv_xbar): every GEMV up to 16 rows, the residual ones above; not below 8 rows (4-bit decode, xbar | wait:
        // 122.49 | 122.69 ms at 32 clips, 101.04 | 101.95 at 8, 94.2 | 93.0 at 1: with a row or two there is nothing to order)
        const int xb = tuning().gemv_xbar == 4 ? (a2.g.B < 8 ? 0 : a2.g.B <= 16 ? 3 : 1) : tuning().gemv_xbar;
        if ((xb & 1 && EPI == DEC_EPI_RESID) || (xb & 2 && EPI != DEC_EPI_RESID)) {
            auto kx = decode_gemvq_kernel<BITS, SBF32, NT, WAVES, KBW, PRO, EPI, KPH, true>;
            ensure_dynamic_lds(reinterpret_cast<const void*>(kx), (int)lds);
            hipLaunchKernelGGL(kx, grid, dim3(WAVES * 64), lds, s, a2);
            return true;
        }
    }
    auto kern = decode_gemvq_kernel<BITS, SBF32, NT, WAVES, KBW, PRO, EPI, KPH>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, s, a2);
    return true;
}

// (K, bits) -> k-blocks per wave at 8 waves; false: no tuned instantiation
template <int BITS, bool SBF32, int NT, int PRO, int EPI>
static bool gemvq_k(const DecGemvQArgs& a2, hipStream_t s) {
    constexpr int M = BITS == 4 ? 1 : 2;           // 8 bit: half the columns per 16-byte block
    switch (a2.g.K) {
        case 1024: return gemvq_go<BITS, SBF32, NT, 8, 1 * M, PRO, EPI>(a2, s);
        case 2048: return gemvq_go<BITS, SBF32, NT, 8, 2 * M, PRO, EPI>(a2, s);
        case 3072:
            if constexpr (PRO == QPRO_COPY) return gemvq_go<BITS, SBF32, NT, 8, 3 * M, PRO, EPI>(a2, s);
            else return false;
        case 6144:            // 1.7B down-projection: two phases of 3072 columns
            if constexpr (PRO == QPRO_COPY) return tuning().gemv_wide ? gemvq_go<BITS, SBF32, NT, 8, 3 * M, PRO, EPI, 2>(a2, s) : false;
            else return false;
        default: return false;
    }
}

template <int BITS, bool SBF32>
static bool gemvq_epi(DecEpi epi, bool norm, const DecGemvQArgs& a2, hipStream_t s) {
    // RMSNorm staging (~500 vector instructions per thread) is replicated in every workgroup and sits on the critical path:
    // at two workgroups per CU it takes twice as long, so the norm kernels take as many 16-row tiles per workgroup as
    // leaves about one workgroup per CU at 32 batch rows (4096 / 32 x 2 = 256, 6144 / 64 x 2 = 192); the packed weights of
    // the extra tiles are a few registers
    if (norm && epi == DEC_EPI_BF16) {
        if (a2.g.N % 32 == 0 && a2.g.N >= 4096) return gemvq_k<BITS, SBF32, 2, QPRO_RMSNORM, DEC_EPI_BF16>(a2, s);
        return gemvq_k<BITS, SBF32, 1, QPRO_RMSNORM, DEC_EPI_BF16>(a2, s);
    }
    if (norm && epi == DEC_EPI_SWIGLU) {
        if (a2.g.N % 64 == 0 && a2.g.N >= 6144) return gemvq_k<BITS, SBF32, 4, QPRO_RMSNORM, DEC_EPI_SWIGLU>(a2, s);
        return gemvq_k<BITS, SBF32, 2, QPRO_RMSNORM, DEC_EPI_SWIGLU>(a2, s);
    }
    if (!norm && epi == DEC_EPI_RESID) return gemvq_k<BITS, SBF32, 1, QPRO_COPY, DEC_EPI_RESID>(a2, s);
    return false;
}

// ------------------------------------------------------------------------------------------------
// Generic kernel for shapes without a tuned instantiation (test geometries; K = 6144 only with gemv_wide = 0): one workgroup per output column,
// thread b = batch row, the factored sum on the vector unit straight from the row-major triplet.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float qdot_row(const QuantRaw& q, long row, const bf16_t* __restrict__ x) {
    const int G = q.K / 64;
    float tot = 0.0f;
    for (int g = 0; g < G; ++g) {
        float acc = 0.0f, xs = 0.0f;
        for (int c = 0; c < 8; ++c) {
            const int ch = g * 8 + c;
            unsigned e[8];
            if (q.bits == 4) {
                const uint32_t w = q.wq[row * (q.K / 8) + ch];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = (w >> (4 * j)) & 0xFu;
            } else {
                const uint32_t w0 = q.wq[row * (q.K / 4) + 2 * ch], w1 = q.wq[row * (q.K / 4) + 2 * ch + 1];
#pragma unroll
                for (int j = 0; j < 4; ++j) { e[j] = (w0 >> (8 * j)) & 0xFFu; e[4 + j] = (w1 >> (8 * j)) & 0xFFu; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = bf16_to_f32(x[ch * 8 + j]);
                acc = fmaf((float)e[j], xv, acc);
                xs += xv;
            }
        }
        const float s = q.sb_f32 ? reinterpret_cast<const float*>(q.scales)[row * G + g] : bf16_to_f32(reinterpret_cast<const bf16_t*>(q.scales)[row * G + g]);
        const float b = q.sb_f32 ? reinterpret_cast<const float*>(q.biases)[row * G + g] : bf16_to_f32(reinterpret_cast<const bf16_t*>(q.biases)[row * G + g]);
        tot += s * acc + b * xs;
    }
    return tot;
}

template <int EPI>
__global__ __launch_bounds__(64) void gemvq_generic_kernel(DecGemvArgs a, QuantRaw q) {
    const int col = blockIdx.x, b = threadIdx.x;
    if (b >= a.B) return;
    const bf16_t* x = a.X + (long)b * a.K;
    if constexpr (EPI == DEC_EPI_SWIGLU) {
        // rows in blocks of 32: 16 gate rows then the 16 matching up rows (engine layout of the fused gate|up matrix)
        const long rg = (long)(col >> 4) * 32 + (col & 15);
        a.out[(long)b * (a.N / 2) + col] = f32_to_bf16(swiglu_bf16(qdot_row(q, rg, x), qdot_row(q, rg + 16, x)));
    } else {
        const float v = qdot_row(q, col, x);
        bf16_t* p = a.out + (long)b * a.N + col;
        if constexpr (EPI == DEC_EPI_RESID) *p = f32_to_bf16(bf16_to_f32(*p) + bf16_round(v));
        else if constexpr (EPI == DEC_EPI_LOGITS) a.logits[(long)b * a.N + col] = bf16_round(v);
        else *p = f32_to_bf16(v);
    }
}

static void gemvq_generic(DecEpi epi, const DecGemvArgs& a, const QuantRaw& q, hipStream_t s) {
    if (a.B > 64) throw std::length_error("decode batch > 64 rows");
    if (q.K % 64 != 0 || q.K != a.K || q.N != a.N) throw std::invalid_argument("quantised gemv: shape mismatch");
    switch (epi) {
        case DEC_EPI_BF16: hipLaunchKernelGGL(gemvq_generic_kernel<DEC_EPI_BF16>, dim3(a.N), dim3(64), 0, s, a, q); break;
        case DEC_EPI_RESID: hipLaunchKernelGGL(gemvq_generic_kernel<DEC_EPI_RESID>, dim3(a.N), dim3(64), 0, s, a, q); break;
        case DEC_EPI_SWIGLU: hipLaunchKernelGGL(gemvq_generic_kernel<DEC_EPI_SWIGLU>, dim3(a.N / 2), dim3(64), 0, s, a, q); break;
        case DEC_EPI_LOGITS: hipLaunchKernelGGL(gemvq_generic_kernel<DEC_EPI_LOGITS>, dim3(a.N), dim3(64), 0, s, a, q); break;
    }
}

void decode_gemv_q_launch(DecEpi epi, const DecGemvArgs& a, const QuantImg& w, const bf16_t* norm_w, float eps,
                          bf16_t* norm_scratch, hipStream_t s) {
    if (a.B <= 0) return;
    const int nt = epi == DEC_EPI_SWIGLU ? 2 : 1;
    bool ok = false;
    if (w.qp && a.B <= 64 && a.N % (16 * nt) == 0 && epi != DEC_EPI_LOGITS) {
        DecGemvQArgs a2{a, w.qp, w.sb, norm_w, eps};
        if (w.bits == 4) ok = w.sb_f32 ? gemvq_epi<4, true>(epi, norm_w != nullptr, a2, s) : gemvq_epi<4, false>(epi, norm_w != nullptr, a2, s);
        else if (w.bits == 8) ok = w.sb_f32 ? gemvq_epi<8, true>(epi, norm_w != nullptr, a2, s) : gemvq_epi<8, false>(epi, norm_w != nullptr, a2, s);
    }
    if (ok) return;
    DecGemvArgs g = a;
    if (norm_w) {
        rmsnorm_rows_launch(a.X, norm_w, norm_scratch, a.B, a.K, eps, s);
        g.X = norm_scratch;
    }
    gemvq_generic(epi, g, w.raw, s);
}

// ------------------------------------------------------------------------------------------------
// LM head on the quantised tied embedding, persistent form (cf. lm_head_kernel): batch rows normalised and staged once
// per workgroup together with their group sums; every wave walks its own 16-row tiles over the full K, 512 columns
// (8 groups) per step, no cross-wave reduction.  A lane owns weight row n = tile * 16 + (lane & 15) and batch rows
// 4 (lane >> 4) .. + 3 of each batch tile; running argmax per batch row over bf16-rounded logits, lowest index on ties.
// Weight pipeline: a ring of four 16-byte q blocks per lane, each slot refilled as soon as it has been consumed (4 KiB per
// wave, 32 KiB per CU in flight: 1.9 TB/s, latency-bound; eight slots compile to 256 registers + scratch).  A two-deep pipeline of whole tiles (two register buffers) compiled to 256 registers
// plus kilobytes of scratch per lane and ran 20x slower; hipcc also wants ~58 registers per batch tile for this loop
// (166 / 224 at 16 / 32 rows), so the kernel runs one workgroup per CU at two waves per SIMD.
// ------------------------------------------------------------------------------------------------
constexpr int LMQ_WAVES = 8;
typedef __attribute__((ext_vector_type(4))) unsigned lds_u32x4;

struct LmHeadQArgs {
    const uint32_t* qp;
    const void* sb;
    const bf16_t* X;
    const bf16_t* norm_w;
    float eps;
    int B, N;
    float* logits;
    float* part_val;
    int* part_idx;
};

// LR > 0: the weight stream goes through a wave-private LDS ring of LR one-KiB slots instead (direct-to-LDS loads, each lane
// reads back the 16 bytes it requested): LDS as an extension of the register file for bytes in flight.  Per 16-row tile the
// wave's block sequence is [scale / bias block(s), q block 0 .. NBLK-1]; block c is consumed after `s_waitcnt vmcnt(LR - 1)`
// (LR blocks are always outstanding: block c + LR is requested as soon as block c has been read, dummy blocks past the end),
// scale / bias blocks are copied to a per-wave "current tile" area so that their slot is free at once.  8 waves x LR KiB in
// flight per CU instead of 32 KiB.
template <int BITS, bool SBF32, int K, int NB, int LR>
__global__ __launch_bounds__(LMQ_WAVES * 64) void lm_head_q_kernel(LmHeadQArgs a, const char* __restrict__ qsrc,
                                                                   const char* __restrict__ sbsrc) {
    constexpr int BLK = BITS == 4 ? 128 : 64, GPB = BLK / 64, NBLK = K / BLK, G = K / 64;
    constexpr int KCH = K / 8, XSTRIDE = 2 * K + 16, TPR = 32, XI = KCH / TPR;
    constexpr int RING = 4;                                       // 16-byte q blocks in flight per lane
    static_assert(NBLK % RING == 0, "K must be a multiple of 512");
    constexpr int SBB = 2 * 16 * G * (SBF32 ? 4 : 2) / 1024;      // scale / bias KiB per tile
    constexpr int PT = SBB + NBLK;                                // blocks per tile in the LDS-ring form
    constexpr size_t XBYTES = ((size_t)NB * 16 * XSTRIDE + (size_t)G * NB * 16 * 4 + 1023) / 1024 * 1024;
    // ONE statically sized LDS object: with a dynamic array hipcc cannot tell the ring from the activation image and drains
    // vmcnt to 0 in front of every LDS read while a direct-to-LDS load is in flight
    __shared__ __attribute__((aligned(1024))) char dsm[XBYTES + (LR > 0 ? (size_t)LMQ_WAVES * (LR + SBB) * 1024 : 0)];
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fc = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* s_x = dsm;                                                            // [NB*16][XSTRIDE]
    float* s_xs = reinterpret_cast<float*>(dsm + (size_t)NB * 16 * XSTRIDE);     // [G][NB*16]
    char* ring = dsm + XBYTES + (size_t)wave * (LR + SBB) * 1024;                // LR slots, then the current tile's scale / bias image
    const int total_waves = gridDim.x * LMQ_WAVES, gw = wave * gridDim.x + blockIdx.x;     // workgroup fastest: see lm_head_kernel
    const int ntiles = a.N / 16;
    const int my_tiles = gw < ntiles ? (ntiles - gw + total_waves - 1) / total_waves : 0;
    const int nblocks = my_tiles * NBLK;
    // ring slot r holds block (it * RING + r) of this wave's block sequence: the packed q words and the scale / bias of its
    // GPB groups.  A slot is refilled right after it has been consumed, so RING blocks (4 KiB per wave) stay in flight
    // without a second register buffer.
    uint4 qr[RING];
    float scr[RING][GPB], bir[RING][GPB];
    auto load_blk = [&](uint4& q, float (&sc)[GPB], float (&bi)[GPB], int bi_seq) {
        const long tile = gw + (long)(bi_seq / NBLK) * total_waves;
        const int kb = bi_seq % NBLK;
        q = *reinterpret_cast<const uint4*>(a.qp + ((tile * NBLK + kb) * 64 + lane) * 4);
        sb_load<SBF32, GPB>(a.sb, ((tile * 16 + fr) * G + kb * GPB) * 2, sc, bi);
#pragma unroll
        for (int h = 0; h < GPB; ++h) bi[h] = eff_bias<BITS>(sc[h], bi[h]);
    };
    // LDS-ring form: request block `seq` of this wave's sequence into slot seq % LR (a dummy block past the end)
    const int nseq = my_tiles * PT;
    auto issue = [&](int seq) {
        const char* src = qsrc;
        if (seq < nseq) {
            const int ti = seq / PT, j = seq - ti * PT;
            const long tile = gw + (long)ti * total_waves;
            src = j < SBB ? sbsrc + (tile * SBB + j) * 1024 : qsrc + (tile * NBLK + (j - SBB)) * 1024;
        }
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + lane * 16), (lds_ptr_t)(ring + (seq % (LR > 0 ? LR : 1)) * 1024), 16, 0, 0);
    };
    // Request order as in lm_head_kernel (dec_lmhead.hip): the norm weights and the first pass's rows go out IN FRONT of the weight prefetch, whose LR
    // requests are unrolled so that the rows' counted wait stays exact -- behind the prefetch, the rows could not be normalised before the ring's first
    // fill had landed (ISA: a full vmcnt(0) in front of the norm weights, then a second round trip for the rows).
    const int srow = tid / TPR, scol = tid % TPR;
    uint4 nwr[XI], xr0[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) nwr[i] = reinterpret_cast<const uint4*>(a.norm_w)[scol + i * TPR];
    {
        const bf16_t* xp0 = a.X + (long)(srow < a.B ? srow : 0) * K + scol * 8;
#pragma unroll
        for (int i = 0; i < XI; ++i) xr0[i] = *reinterpret_cast<const uint4*>(xp0 + i * TPR * 8);
    }
    if constexpr (LR > 0) {
        // hipcc waits with vmcnt(0) for anything that precedes direct-to-LDS copies still in flight, so the rows are waited for HERE, before the ring's
        // first fill is requested (a cache round trip, ~1 us): their normalisation then runs under the fill instead of behind it
        reg_use(xr0);
        reg_use(nwr);
#pragma unroll
        for (int r = 0; r < LR; ++r) issue(r);
    } else if (nblocks > 0) {
#pragma unroll
        for (int r = 0; r < RING; ++r) load_blk(qr[r], scr[r], bir[r], r);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage + RMSNorm the batch rows, 16 rows per pass; group sums of the staged values ------------------
    {
#pragma unroll 1
        for (int nb = 0; nb < NB; ++nb) {
            const int r = nb * 16 + srow;
            const bool live = r < a.B;
            const bf16_t* xp = a.X + (long)(live ? r : 0) * K + scol * 8;
            uint4 xr[XI];
            if (nb == 0) {
#pragma unroll
                for (int i = 0; i < XI; ++i) xr[i] = xr0[i];
            } else {
#pragma unroll
                for (int i = 0; i < XI; ++i) xr[i] = *reinterpret_cast<const uint4*>(xp + i * TPR * 8);
            }
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const bf16_t* e = reinterpret_cast<const bf16_t*>(&xr[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float f = bf16_to_f32(e[j]); ss = fmaf(f, f, ss); }
            }
            ss = lane_sum<TPR>(ss);
            const float inv = rsqrtf(ss / (float)K + a.eps);
            char* xrow = s_x + (size_t)r * XSTRIDE + scol * 16;
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const uint4 nw = nwr[i];
                uint4 o = make_uint4(0, 0, 0, 0);
                if (live)
                    o = make_uint4(rmsnorm_pair_bf16(xr[i].x, nw.x, inv), rmsnorm_pair_bf16(xr[i].y, nw.y, inv),
                                   rmsnorm_pair_bf16(xr[i].z, nw.z, inv), rmsnorm_pair_bf16(xr[i].w, nw.w, inv));
                // the group sum in the order of the per-element loop this replaces: ((((((e0 + e1) + e2) + e3) + ...
                float p = 0.0f;
                p += bf16_lo(o.x); p += bf16_hi(o.x); p += bf16_lo(o.y); p += bf16_hi(o.y);
                p += bf16_lo(o.z); p += bf16_hi(o.z); p += bf16_lo(o.w); p += bf16_hi(o.w);
                *reinterpret_cast<uint4*>(xrow + i * TPR * 16) = o;
                p = lane_sum8(p);
                if ((scol & 7) == 0) s_xs[((scol + TPR * i) >> 3) * (NB * 16) + r] = p;
            }
        }
    }
    __syncthreads();
    float best[NB][4];
    int bidx[NB][4];
    f32x4 tot[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        tot[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) { best[b][j] = -INFINITY; bidx[b][j] = 0x7fffffff; }
    }
    if constexpr (LR > 0) {
        const char* sbcur = ring + LR * 1024;
#pragma unroll 1
        for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll 1
            for (int j = 0; j < PT; ++j) {
                const int seq = ti * PT + j;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LR - 1) : "memory");       // block seq has landed
                // native vector type on purpose: hipcc's wait insertion leaves an LDS read alone while direct-to-LDS loads are in
                // flight only if the read carries type-based alias info (a HIP_vector_type / char access gets `vmcnt(0)` in
                // front of it, which would serialise the ring); the counted wait above is the real dependency
                const lds_u32x4 raw = *reinterpret_cast<const lds_u32x4*>(ring + (seq % LR) * 1024 + lane * 16);
                const uint4 blk = make_uint4(raw[0], raw[1], raw[2], raw[3]);
                if (j < SBB) {                                  // wave-uniform
                    *reinterpret_cast<lds_u32x4*>(const_cast<char*>(sbcur) + j * 1024 + lane * 16) = raw;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the slot has been read before it is requested again
                    issue(seq + LR);
                    continue;
                }
                const int kb = j - SBB;
                mfma_bf16x8 wf[2 * GPB];
#pragma unroll
                for (int h = 0; h < GPB; ++h) {
                    wf[2 * h] = frag_of<BITS>(blk, 2 * h + 0);
                    wf[2 * h + 1] = frag_of<BITS>(blk, 2 * h + 1);
                }
                float sg[GPB], bg[GPB];
                sb_load<SBF32, GPB>(sbcur, ((long)fr * G + kb * GPB) * 2, sg, bg);
#pragma unroll
                for (int h = 0; h < GPB; ++h) bg[h] = eff_bias<BITS>(sg[h], bg[h]);
                __builtin_amdgcn_sched_barrier(0);
                issue(seq + LR);                                // blk is in registers (the fragment conversion consumed it)
                const char* xw = s_x + (size_t)fr * XSTRIDE + fc * 16 + (size_t)kb * GPB * 128;
                const float* xsw = s_xs + (size_t)kb * GPB * (NB * 16) + fc * 4;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
#pragma unroll
                    for (int h = 0; h < GPB; ++h) {
                        const mfma_bf16x8 x0 = *reinterpret_cast<const mfma_bf16x8*>(xw + (size_t)b * 16 * XSTRIDE + (h * 2 + 0) * 64);
                        const mfma_bf16x8 x1 = *reinterpret_cast<const mfma_bf16x8*>(xw + (size_t)b * 16 * XSTRIDE + (h * 2 + 1) * 64);
                        const f32x4 xs = *reinterpret_cast<const f32x4*>(xsw + h * (NB * 16) + b * 16);
                        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x0, wf[2 * h], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x1, wf[2 * h + 1], acc, 0, 0, 0);
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) tot[b][jj] += sg[h] * acc[jj] + bg[h] * xs[jj];
                    }
                }
            }
            const int n = (gw + ti * total_waves) * 16 + fr;    // tile complete: tot[b][j] = logit[batch b*16 + fc*4 + j][n]
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int row = b * 16 + fc * 4 + jj;
                    const float v = bf16_round(tot[b][jj]);
                    if (a.logits && row < a.B) a.logits[(long)row * a.N + n] = v;
                    if (v > best[b][jj] || (v == best[b][jj] && n < bidx[b][jj])) { best[b][jj] = v; bidx[b][jj] = n; }
                    tot[b][jj] = 0.0f;
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the dummy blocks still in flight write LDS
    } else {
#pragma unroll 1
    for (int b0 = 0; b0 < nblocks; b0 += RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            const int kb = (b0 + r) % NBLK;                       // block within the tile (RING divides NBLK: same tile for all r)
            // fragments converted once, then used for every batch tile
            mfma_bf16x8 wf[2 * GPB];
#pragma unroll
            for (int h = 0; h < GPB; ++h) {
                wf[2 * h] = frag_of<BITS>(qr[r], 2 * h + 0);
                wf[2 * h + 1] = frag_of<BITS>(qr[r], 2 * h + 1);
            }
            const float sg0 = scr[r][0], bg0 = bir[r][0], sg1 = scr[r][GPB - 1], bg1 = bir[r][GPB - 1];
            if (b0 + RING < nblocks) load_blk(qr[r], scr[r], bir[r], b0 + RING + r);     // refill the slot at once
            const char* xw = s_x + (size_t)fr * XSTRIDE + fc * 16 + (size_t)kb * GPB * 128;
            const float* xsw = s_xs + (size_t)kb * GPB * (NB * 16) + fc * 4;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int h = 0; h < GPB; ++h) {
                    const uint4 x0 = *reinterpret_cast<const uint4*>(xw + (size_t)b * 16 * XSTRIDE + (h * 2 + 0) * 64);
                    const uint4 x1 = *reinterpret_cast<const uint4*>(xw + (size_t)b * 16 * XSTRIDE + (h * 2 + 1) * 64);
                    const f32x4 xs = *reinterpret_cast<const f32x4*>(xsw + h * (NB * 16) + b * 16);
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, x0), wf[2 * h], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, x1), wf[2 * h + 1], acc, 0, 0, 0);
                    const float sg = h == 0 ? sg0 : sg1, bg = h == 0 ? bg0 : bg1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) tot[b][j] += sg * acc[j] + bg * xs[j];
                }
            }
            // one block at a time: hoisting the LDS reads of later blocks costs more registers than the ring leaves
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((b0 + RING) % NBLK == 0) {                         // tile complete: tot[b][j] = logit[batch b*16 + fc*4 + j][n]
            const int n = (gw + (b0 / NBLK) * total_waves) * 16 + fr;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = b * 16 + fc * 4 + j;
                    const float v = bf16_round(tot[b][j]);
                    if (a.logits && row < a.B) a.logits[(long)row * a.N + n] = v;
                    if (v > best[b][j] || (v == best[b][j] && n < bidx[b][j])) { best[b][j] = v; bidx[b][j] = n; }
                    tot[b][j] = 0.0f;
                }
        }
    }
    }
    // ---- argmax partial of the workgroup: over the 16 weight rows of a lane group, then over the waves ----------
    float* s_v = reinterpret_cast<float*>(dsm);                                   // the activation image is dead now
    int* s_i = reinterpret_cast<int*>(dsm + LMQ_WAVES * NB * 16 * sizeof(float));
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int ofs = 1; ofs < 16; ofs <<= 1) {
                const float ov = __shfl_xor(best[b][j], ofs, 64);
                const int oi = __shfl_xor(bidx[b][j], ofs, 64);
                if (ov > best[b][j] || (ov == best[b][j] && oi < bidx[b][j])) { best[b][j] = ov; bidx[b][j] = oi; }
            }
            if (fr == 0) {
                s_v[(wave * NB + b) * 16 + fc * 4 + j] = best[b][j];
                s_i[(wave * NB + b) * 16 + fc * 4 + j] = bidx[b][j];
            }
        }
    __syncthreads();
    if (tid < NB * 16 && tid < a.B) {
        const int b = tid >> 4, r = tid & 15;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int w = 0; w < LMQ_WAVES; ++w) {
            const float ov = s_v[(w * NB + b) * 16 + r];
            const int oi = s_i[(w * NB + b) * 16 + r];
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        a.part_val[(long)tid * gridDim.x + blockIdx.x] = bv;
        a.part_idx[(long)tid * gridDim.x + blockIdx.x] = bi;
    }
}

constexpr int LMQ_GRID = 256;

static bool lm_head_q_supported(int N, int K, int bits) {
    return (K == 1024 || K == 2048) && (bits == 4 || bits == 8) && N % 16 == 0 && N / 16 >= LMQ_GRID * LMQ_WAVES;
}
int lm_head_q_parts(int N, int K, int bits) { return lm_head_q_supported(N, K, bits) ? LMQ_GRID : 1; }

template <int BITS, bool SBF32, int K, int NB, int LR>
static void lm_head_q_go1(const LmHeadQArgs& a, hipStream_t s) {
    constexpr size_t xbytes = ((size_t)NB * 16 * (2 * K + 16) + (size_t)(K / 64) * NB * 16 * 4 + 1023) / 1024 * 1024;
    constexpr int sbb = 2 * 16 * (K / 64) * (SBF32 ? 4 : 2) / 1024;
    constexpr size_t lds = xbytes + (LR > 0 ? (size_t)LMQ_WAVES * (LR + sbb) * 1024 : 0);
    static_assert(lds <= 160 * 1024, "LM head: LDS image + weight ring exceed the CU");
    auto kern = lm_head_q_kernel<BITS, SBF32, K, NB, LR>;
    hipLaunchKernelGGL(kern, dim3(LMQ_GRID), dim3(LMQ_WAVES * 64), 0, s, a, reinterpret_cast<const char*>(a.qp),
                       reinterpret_cast<const char*>(a.sb));
}

// ring depth by what the activation image leaves of the 160 KiB (K = 1024: 33 KiB per batch tile): 13 slots per wave at
// <= 16 rows, 9 at <= 32 (f32 scales: one slot less); larger batches / hidden sizes keep the register ring
template <int BITS, bool SBF32, int K, int NB>
static void lm_head_q_go(const LmHeadQArgs& a, hipStream_t s) {
    if constexpr (K == 1024 && NB <= 2) {
        if (tuning().lmh_q_ring != 0) {
            lm_head_q_go1<BITS, SBF32, K, NB, (NB == 1 ? 13 : 9) - (SBF32 ? 1 : 0)>(a, s);
            return;
        }
    }
    lm_head_q_go1<BITS, SBF32, K, NB, 0>(a, s);
}

template <int BITS, bool SBF32, int K>
static void lm_head_q_nb(const LmHeadQArgs& a, hipStream_t s) {
    switch ((a.B + 15) / 16) {
        case 1: lm_head_q_go<BITS, SBF32, K, 1>(a, s); break;
        case 2: lm_head_q_go<BITS, SBF32, K, 2>(a, s); break;
        case 3: if constexpr (K == 1024) { lm_head_q_go<BITS, SBF32, K, 3>(a, s); break; }
        case 4: if constexpr (K == 1024) { lm_head_q_go<BITS, SBF32, K, 4>(a, s); break; }
        default: throw std::length_error("LM head: batch rows exceed the LDS image at this hidden size");
    }
}

// generic path: logits of every row to `logits_scratch`, then one argmax partial per row
__global__ __launch_bounds__(256) void argmax_f32_rows_kernel(const float* __restrict__ x, int n, float* __restrict__ part_val,
                                                              int* __restrict__ part_idx) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const float* row = x + (long)blockIdx.x * n;
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
    if (threadIdx.x == 0) { part_val[blockIdx.x] = s_v[0]; part_idx[blockIdx.x] = s_i[0]; }
}

int lm_head_q_launch(const QuantImg& w, const bf16_t* X, const bf16_t* norm_w, float eps, int B, int N, int K, float* logits,
                     float* part_val, int* part_idx, bf16_t* norm_scratch, hipStream_t s) {
    if (B <= 0) return 0;
    if (w.qp && lm_head_q_supported(N, K, w.bits)) {
        LmHeadQArgs a{w.qp, w.sb, X, norm_w, eps, B, N, logits, part_val, part_idx};
#define QASR_LMQ(BITS_, F32_)                                                              \
        do { if (K == 1024) lm_head_q_nb<BITS_, F32_, 1024>(a, s); else lm_head_q_nb<BITS_, F32_, 2048>(a, s); } while (0)
        if (w.bits == 4) { if (w.sb_f32) QASR_LMQ(4, true); else QASR_LMQ(4, false); }
        else { if (w.sb_f32) QASR_LMQ(8, true); else QASR_LMQ(8, false); }
#undef QASR_LMQ
        return LMQ_GRID;
    }
    // generic: needs a logits buffer (the engine always passes one on this path)
    if (!logits) throw std::invalid_argument("quantised LM head (generic path) needs the logits buffer");
    DecGemvArgs g{};
    g.X = X; g.B = B; g.N = N; g.K = K; g.logits = logits;
    rmsnorm_rows_launch(X, norm_w, norm_scratch, B, K, eps, s);
    g.X = norm_scratch;
    gemvq_generic(DEC_EPI_LOGITS, g, w.raw, s);
    hipLaunchKernelGGL(argmax_f32_rows_kernel, dim3(B), dim3(256), 0, s, logits, N, part_val, part_idx);
    return 1;
}

}  // namespace qasr

// dec_gemv.hip -- decode-step skinny GEMMs: generic and tuned (weights register-resident, activations in LDS, fused RMSNorm / epilogues), weight repack (declarations: dec_kernels.h).
#include "dec_kernels.h"
#include "dec_epilogue.h"
#include <cstdlib>
#include <cstdio>

namespace qasr {

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

// PARTIAL (host: fewer than 16 batch rows, norm prologue): batch rows past the end get a zero image without the ~250 vector instructions of
// the normalisation, so the waves that own live rows have the SIMDs to themselves (1 / 8 clips: decode -1.5 %); as a run-time branch in
// the one kernel it cost the full-tile launches 3 % (profiles/r02_ab_gemv_skip_dead_rows.txt), hence a separate instantiation
// EARLYW (host: at most 8 batch rows): the weight stream is requested WITHOUT waiting for the activation rows.  With a full batch that
// order loses in the real step (see mask_x below); with a few rows X is a couple of KB, nothing queues behind it, and the launch is the
// serial chain  X back (1.4 us) -> weights requested -> weights back (1.9 us)  that this overlaps (in-kernel stamps at 1 clip, round 3).
// NTW: weight fragments by non-temporal loads (a template parameter: as a run-time branch hipcc merged the two load blocks and dropped the
// hint, round 2); A/B in profiles/r04_ab_gemv_nt.txt
typedef __attribute__((ext_vector_type(4))) unsigned gemv_u32x4;
// XBAR: the weight requests follow the row requests WITHOUT waiting for the rows to come back -- a bare s_barrier between the two makes every wave's row
// requests enter the CU's request queue before any wave's weight requests, which is what the wait for the data was used for (see mask_x below).
template <int NT, int NB, int WAVES, int KSW, bool ALLROWS, int PRO, int EPI, bool PARTIAL = false, bool EARLYW = false, bool NTW = false, bool XBAR = false>
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
    // norm weights: requested with the rows (2 KB every workgroup shares: cache hits).  Left inside the staging loop they were a dependent round
    // trip behind the rows' wait, and hipcc sank the weight requests below them AND below the norm reduction (ISA of the gate|up instantiation:
    // weights issued ~1 us after the rows were back)
    // Only in the small-batch instantiations (EARLYW): at 32 rows the late order is the faster one (in-box A/B of two builds: decode 123.35 against
    // 124.3 ms; +0.3 % / +0.7 % the other way at 1 / 8 rows) -- the same finding as for the weight stream itself, see the comment below.
    constexpr bool HOISTNW = PRO == DEC_PRO_RMSNORM && (EARLYW || XBAR);
    uint4 nwr[HOISTNW ? XI : 1];
    if constexpr (HOISTNW) {
#pragma unroll
        for (int i = 0; i < XI; ++i) nwr[i] = reinterpret_cast<const uint4*>(a2.norm_w)[scol + i * TPR];
    }
    if constexpr (XBAR) __builtin_amdgcn_s_barrier();
    if (!EARLYW && !XBAR) mask_x(0);
    uint4 w[NT][KSW];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        // fragment-major packed weights: block (row tile, k-step) = 1 KiB, lane-major -> every wave
        // instruction reads 1 KiB contiguous (see pack_mfma_a_kernel)
        const bf16_t* wp = a.Wp + ((long)(n0 / 16 + t) * (K / 32)) * 512 + lane * 8;
#pragma unroll
        for (int i = 0; i < KSW; ++i) {
            if constexpr (NTW) {
                const gemv_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const gemv_u32x4*>(wp + (long)(wave + WAVES * i) * 512));
                w[t][i] = make_uint4(v.x, v.y, v.z, v.w);
            } else {
                w[t][i] = *reinterpret_cast<const uint4*>(wp + (long)(wave + WAVES * i) * 512);
            }
        }
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
    if (EARLYW || XBAR) mask_x(0);
    // the requests above stay above: the norm arithmetic below (a reduction with five dependent cross-lane steps) must not be scheduled in front of them
    if constexpr (HOISTNW) __builtin_amdgcn_sched_barrier(0);
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
        if (PARTIAL && PRO == DEC_PRO_RMSNORM && ph * RPP + srow >= a.B) {
#pragma unroll
            for (int i = 0; i < XI; ++i) *reinterpret_cast<uint4*>(xrow + i * TPR * 16) = make_uint4(0, 0, 0, 0);
        } else if constexpr (PRO == DEC_PRO_RMSNORM) {
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
                uint4 nw;
                if constexpr (HOISTNW) nw = nwr[i];
                else nw = reinterpret_cast<const uint4*>(a2.norm_w)[scol + i * TPR];
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

template <int NT, int NB, int WAVES, int KSW, bool ALLROWS, int PRO, int EPI, bool PARTIAL = false, bool EARLYW = false>
static bool gemv2_go(const DecGemv2Args& a2, hipStream_t s) {
    constexpr size_t lds = gemv2_lds<NT, NB, WAVES, KSW, ALLROWS>();
    if constexpr (lds > 156 * 1024) {
        return false;
    } else {
        auto go = [&](auto kern) {
            ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds);
            hipLaunchKernelGGL(kern, dim3(a2.g.N / (16 * NT), a2.row_groups > 1 ? a2.row_groups : 1), dim3(WAVES * 64), lds, s, a2);
        };
        // non-temporal weight loads: only the plain full-tile instantiations carry the variant (EPI LOGITS is the generic head, untouched)
        if constexpr (!PARTIAL && EPI != DEC_EPI_LOGITS) {
            if (tuning().gemv_nt) { go(decode_gemv2_kernel<NT, NB, WAVES, KSW, ALLROWS, PRO, EPI, PARTIAL, EARLYW, true>); return true; }
        }
        if constexpr (!PARTIAL && !EARLYW && EPI != DEC_EPI_LOGITS && NB == 1 && ALLROWS) {
            // gemv_xbar: 1 residual GEMVs | 2 the norm GEMVs | 3 both | 4 (default) by batch: both up to 16 rows (-1.7 % decode at 16), the residual ones
            // above (-0.2 % at 32; the norm GEMV, 384 workgroups there, loses 1.1 %) | 0 none
            const int xb = tuning().gemv_xbar == 4 ? (a2.g.B <= 16 ? 3 : 1) : tuning().gemv_xbar;
            if ((xb & 1 && EPI == DEC_EPI_RESID) || (xb & 2 && EPI != DEC_EPI_RESID)) {
                go(decode_gemv2_kernel<NT, NB, WAVES, KSW, ALLROWS, PRO, EPI, PARTIAL, EARLYW, false, true>);
                return true;
            }
        }
        go(decode_gemv2_kernel<NT, NB, WAVES, KSW, ALLROWS, PRO, EPI, PARTIAL, EARLYW, false>);
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
        const int ew = tuning().gemv_earlyw;          // 2: every GEMV | 3: the residual GEMVs only (A/B at full batches; see EARLYW)
        if (ew == 2 || (ew == 3 && EPI == DEC_EPI_RESID)) return gemv2_go<NT, 1, WAVES, KSW, true, PRO, EPI, false, true>(b2, s);
        return gemv2_go<NT, 1, WAVES, KSW, true, PRO, EPI>(b2, s);
    }
    // all batch rows resident in LDS when they fit (LDS and staging registers), else 16 rows per phase
    constexpr bool fit2 = gemv2_lds<NT, 2, WAVES, KSW, true>() <= 150 * 1024 && NT * KSW * 4 + 2 * KSW * 4 <= 170;
    switch (nb) {
        case 1:
            if constexpr (EPI != DEC_EPI_LOGITS) {
                const bool earlyw = (a2.g.B <= 8 && tuning().gemv_earlyw == 1) || tuning().gemv_earlyw == 2;
                if constexpr (PRO == DEC_PRO_RMSNORM) {
                    if (a2.g.B < 16 && tuning().gemv_partial)
                        return earlyw ? gemv2_go<NT, 1, WAVES, KSW, true, PRO, EPI, true, true>(a2, s) : gemv2_go<NT, 1, WAVES, KSW, true, PRO, EPI, true>(a2, s);
                }
                if (earlyw) return gemv2_go<NT, 1, WAVES, KSW, true, PRO, EPI, false, true>(a2, s);
            }
            return gemv2_go<NT, 1, WAVES, KSW, true, PRO, EPI>(a2, s);
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
static thread_local unsigned long long* g_gemv_dbg = nullptr;      // per host thread: engines of several devices step concurrently (qasr_dp_*)
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

}  // namespace qasr

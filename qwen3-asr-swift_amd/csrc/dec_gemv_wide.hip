// dec_gemv_wide.hip -- decode-step skinny GEMM for K too wide for one LDS image of the 16 activation rows (the 1.7B preset's
// down-projection: K = 6144, 16 rows x 12 KiB = 192 KiB against 160 KiB of LDS).
//
// Same form as decode_gemv2_kernel (dec_gemv.hip): one workgroup of 8 waves per (16-column weight tile, 16 batch rows), every
// wave first puts ALL its weight fragments in flight (fragment-major image, 1 KiB per wave instruction), MFMA operands A = weights
// from registers, B = activation rows from a padded LDS image, fixed-order cross-wave reduction, epilogue on wave 0.  The difference:
// the activation rows are staged in KPH column phases of K / KPH columns through the same image (barrier, MFMAs of that phase's
// k-steps, barrier, next phase); the later phases' rows are already in registers when the first phase is consumed.
// Accumulation order per output: wave w adds its k-steps in ascending order (phase 0 first), then wave 0 + 1 + ... + 7.
#include "dec_gemv_wide.h"
#include "dec_epilogue.h"

namespace qasr {

template <int KSW, int KPH, int EPI>
__global__ __launch_bounds__(512) void decode_gemv_wide_kernel(DecGemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    constexpr int WAVES = 8;
    constexpr int KH = KSW * WAVES * 32;                    // columns per phase
    constexpr int K = KH * KPH;                             // host checks a.K == K
    constexpr int XSTRIDE = 2 * KH + 16;                    // bytes: 16 rows x one 16-byte chunk cover all 64 banks once
    constexpr int TPR = WAVES * 64 / 16;                    // threads sharing one activation row
    constexpr int XI = KH / 8 / TPR;                        // staged 16-byte chunks per thread per phase
    static_assert(KH / 8 % TPR == 0, "row staging geometry");
    {                                                       // batch rows in groups of 16 on gridDim.y
        const int r0 = blockIdx.y * 16;
        a.X += (long)r0 * K;
        a.out += (long)r0 * a.N;
        a.B = a.B - r0 < 16 ? a.B - r0 : 16;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fc = lane >> 4;
    const int n0 = blockIdx.x * 16;
    char* s_x = dsm;                                                                 // [16][XSTRIDE]
    float* s_red = reinterpret_cast<float*>(dsm + (size_t)16 * XSTRIDE);             // [WAVES - 1][256]
    const int srow = tid / TPR, scol = tid % TPR;
    const bf16_t* xp = a.X + (long)(srow < a.B ? srow : 0) * K + scol * 8;
    uint4 xr[KPH][XI];
    // activation rows of phase 0 first and waited for (they are a fabric read behind the kernel boundary; the weight stream queued
    // ahead of them delays them: dec_gemv.hip, decode_gemv2_kernel), then every weight fragment, then the later phases' rows
#pragma unroll
    for (int i = 0; i < XI; ++i) xr[0][i] = *reinterpret_cast<const uint4*>(xp + i * TPR * 8);
    if (srow >= a.B) {
#pragma unroll
        for (int i = 0; i < XI; ++i) xr[0][i] = make_uint4(0, 0, 0, 0);
    }
    uint4 w[KPH][KSW];
    {
        const bf16_t* wp = a.Wp + ((long)(n0 / 16) * (K / 32)) * 512 + lane * 8;
#pragma unroll
        for (int kp = 0; kp < KPH; ++kp)
#pragma unroll
            for (int i = 0; i < KSW; ++i) w[kp][i] = *reinterpret_cast<const uint4*>(wp + (long)(kp * (KH / 32) + wave + WAVES * i) * 512);
    }
#pragma unroll
    for (int kp = 1; kp < KPH; ++kp)
#pragma unroll
        for (int i = 0; i < XI; ++i) xr[kp][i] = *reinterpret_cast<const uint4*>(xp + kp * KH + i * TPR * 8);
    uint2 rsd[1][1];
    if constexpr (EPI == DEC_EPI_RESID) {
        if (wave == 0) rsd[0][0] = *reinterpret_cast<const uint2*>(a.out + (long)(fr < a.B ? fr : 0) * a.N + n0 + fc * 4);
    }
    f32x4 acc[1][1];
    acc[0][0] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < KPH; ++kp) {
        if (kp > 0) {
            __syncthreads();                                // the previous phase's LDS reads are done
            if (srow >= a.B) {
#pragma unroll
                for (int i = 0; i < XI; ++i) xr[kp][i] = make_uint4(0, 0, 0, 0);
            }
        }
        char* xrow = s_x + (size_t)srow * XSTRIDE + scol * 16;
#pragma unroll
        for (int i = 0; i < XI; ++i) *reinterpret_cast<uint4*>(xrow + i * TPR * 16) = xr[kp][i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < KSW; ++i) {
            const int kb = ((wave + WAVES * i) * 32 + fc * 8) * 2;
            const uint4 xf = *reinterpret_cast<const uint4*>(s_x + (size_t)fr * XSTRIDE + kb);
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, w[kp][i]),
                                                                __builtin_bit_cast(mfma_bf16x8, xf), acc[0][0], 0, 0, 0);
        }
    }
    if (wave > 0) *reinterpret_cast<f32x4*>(&s_red[(size_t)(wave - 1) * 256 + lane * 4]) = acc[0][0];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int wv = 0; wv < WAVES - 1; ++wv) acc[0][0] += *reinterpret_cast<const f32x4*>(&s_red[(size_t)wv * 256 + lane * 4]);
    if constexpr (EPI == DEC_EPI_RESID) dec_epilogue<1, 1, EPI>(a, acc, n0, fr, fc, rsd);
    else dec_epilogue<1, 1, EPI>(a, acc, n0, fr, fc);
}

template <int KSW, int KPH, int EPI>
static void gemv_wide_go(const DecGemvArgs& a, hipStream_t s) {
    constexpr size_t lds = (size_t)16 * (2 * (KSW * 8 * 32) + 16) + (size_t)7 * 1024;
    static_assert(lds <= 156 * 1024, "activation image + reduction scratch exceed the CU");
    auto kern = decode_gemv_wide_kernel<KSW, KPH, EPI>;
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds);
    hipLaunchKernelGGL(kern, dim3(a.N / 16, (a.B + 15) / 16), dim3(512), lds, s, a);
}

bool decode_gemv_wide_supported(DecEpi epi, const DecGemvArgs& a) {
    return a.Wp != nullptr && a.K == 6144 && a.N % 16 == 0 && a.B > 0 && a.B <= 64 && (epi == DEC_EPI_RESID || epi == DEC_EPI_BF16);
}

int decode_gemv_wide_launch(DecEpi epi, const DecGemvArgs& a, hipStream_t s) {
    if (!decode_gemv_wide_supported(epi, a)) throw std::invalid_argument("decode_gemv_wide: unsupported shape");
    if (epi == DEC_EPI_RESID) gemv_wide_go<12, 2, DEC_EPI_RESID>(a, s);     // 2 phases x 3072 columns
    else gemv_wide_go<12, 2, DEC_EPI_BF16>(a, s);
    return a.N / 16;
}

}  // namespace qasr

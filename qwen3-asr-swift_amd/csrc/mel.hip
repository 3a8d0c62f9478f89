// mel.hip -- batched 128-bin log-mel front-end for gfx950.
//
// Replaces WhisperFeatureExtractor.extractFeatures (reference:
// Sources/Qwen3ASR/AudioPreprocessing.swift:169-317; tables :39-53,:61-164):
//   reflect pad 200 | frames of 400 @ hop 160 x periodic Hann | zero-pad to 512 | real FFT |
//   power (x fft_scale^2: vDSP_fft_zrip returns 2x the DFT) | slaney mel (128 x 257) |
//   max(.,1e-10) | log10 | per-clip max over ALL frames | max(., gmax-8) * 0.25 + 1 |
//   drop last frame | [128, T] row-major.
//
// Kernel 1 (mel_frames): one wavefront per frame.  The 512-point real FFT is a 256-point
// complex Stockham radix-4 FFT (4 passes, ping-pong in LDS, 4 points per lane) followed by the
// even/odd split; the filterbank is applied in its sparse form (every FFT bin feeds <= 2
// triangles, 504 non-zeros instead of 32896) with the power spectrum staged in LDS.  HBM traffic:
// each PCM sample is read 2.5x (frame overlap, served by L2) and 128 floats are written per frame.
// Kernel 2 (mel_finalize): clamp/scale with the per-clip max and transpose [frame][mel] ->
// [mel][frame] through an LDS tile so both sides stay coalesced.
#include "mel.h"
#include "mel_core.h"

namespace qasr {

void MelTables::build(float fft_scale) {
    std::vector<float> t(T_TOTAL, 0.0f);
    for (int i = 0; i < MEL_NFFT; ++i)    // :41-44
        t[T_HANN + i] = 0.5f * (1.0f - cosf(2.0f * (float)M_PI * (float)i / 400.0f));
    melc_fill_tables(t, fft_scale * fft_scale);     // twiddles, slaney filterbank on the 512-point grid (:88-153), power scale
    bytes = T_TOTAL * sizeof(float);
    QASR_HIP(hipMalloc(&dev, bytes));
    QASR_HIP(hipMemcpy(dev, t.data(), bytes, hipMemcpyHostToDevice));
}

void MelTables::release() {
    if (dev) (void)hipFree(dev);
    dev = nullptr;
}

// order-preserving float <-> uint so atomicMax works on signed floats
__device__ __forceinline__ unsigned f32_ordered(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_f32(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

constexpr int MEL_WAVES = 4;              // frames in flight per workgroup
constexpr int MEL_FPW = 8;                // frames per wave (sequential)

__global__ __launch_bounds__(MEL_WAVES * 64) void mel_frames_kernel(
    const float* __restrict__ tab, const float* __restrict__ pcm, const long* __restrict__ pcm_off,
    const int* __restrict__ n_samples, const int* __restrict__ frame_off, float* __restrict__ raw,
    unsigned* __restrict__ gmax) {
    __shared__ float s_tab[T_TOTAL];
    __shared__ float2 s_buf[MEL_WAVES][2][256];
    __shared__ float s_pow[MEL_WAVES][260];
    __shared__ float s_max[MEL_WAVES];

    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < T_TOTAL; i += MEL_WAVES * 64) s_tab[i] = tab[i];
    __syncthreads();

    const int n = n_samples[b];
    const int nf = n / MEL_HOP + 1;                       // frames incl. the dropped last one
    const float* x = pcm + pcm_off[b];
    float* out = raw + (long)frame_off[b] * MEL_NMELS;
    const float2* tw256 = reinterpret_cast<const float2*>(&s_tab[T_TW256]);
    const float2* tw512 = reinterpret_cast<const float2*>(&s_tab[T_TW512]);
    const float scale2 = s_tab[T_SCALE2];
    float2* bufA = s_buf[wave][0];
    float2* bufB = s_buf[wave][1];
    float* pw = s_pow[wave];
    float vmax = -INFINITY;

    const int frame0 = blockIdx.x * (MEL_WAVES * MEL_FPW) + wave * MEL_FPW;
    for (int fi = 0; fi < MEL_FPW; ++fi) {
        const int frame = frame0 + fi;
        const bool live = frame < nf;                     // barriers below stay uniform
        // ---- load 400 windowed samples as 200 packed complex points, 4 per lane -------------
        cplx v[4];
        const long start = (long)frame * MEL_HOP - MEL_NFFT / 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = lane + 64 * r;                  // packed index: samples 2p, 2p+1
            float a0 = 0.0f, a1 = 0.0f;
            if (live && p < MEL_NFFT / 2) {
                long i0 = start + 2 * p, i1 = i0 + 1;
                // reflect pad with the reference's clamps (AudioPreprocessing.swift:178-192)
                long j0 = i0 < 0 ? -i0 : (i0 >= n ? 2L * n - 2 - i0 : i0);
                long j1 = i1 < 0 ? -i1 : (i1 >= n ? 2L * n - 2 - i1 : i1);
                j0 = j0 < 0 ? 0 : (j0 > n - 1 ? n - 1 : j0);
                j1 = j1 < 0 ? 0 : (j1 > n - 1 ? n - 1 : j1);
                a0 = x[j0] * s_tab[T_HANN + 2 * p];
                a1 = x[j1] * s_tab[T_HANN + 2 * p + 1];
            }
            v[r] = {a0, a1};
        }
        // ---- 512-point real FFT of the frame -> power spectrum in pw (mel_core.h) --------------
        melc_frame_power(v, lane, bufA, bufB, pw, tw256, tw512, scale2);
        // ---- sparse slaney filterbank + log10 ------------------------------------------------
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = lane + 64 * h;
            const float acc = melc_filter(s_tab, pw, m);
            float lg = log10f(fmaxf(acc, 1e-10f));
            if (live) {
                out[(long)frame * MEL_NMELS + m] = lg;
                vmax = fmaxf(vmax, lg);
            }
        }
        __syncthreads();
    }
    vmax = wave_max(vmax);
    if (lane == 0) s_max[wave] = vmax;
    __syncthreads();
    if (tid == 0) {
        float m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        if (m > -INFINITY) atomicMax(&gmax[b], f32_ordered(m));
    }
}

// raw [frames_all][128] -> out [128][stride] (first T frames), clamp + scale.
__global__ __launch_bounds__(256) void mel_finalize_kernel(
    const float* __restrict__ raw, const unsigned* __restrict__ gmax, const int* __restrict__ n_samples,
    const int* __restrict__ frame_off, float* __restrict__ out, int out_stride) {
    __shared__ float tile[64][MEL_NMELS + 1];
    const int b = blockIdx.y;
    const int n = n_samples[b];
    int T = n / MEL_HOP;                                  // frames_all - 1 (drop last, :296)
    if (T > MEL_MAX_FRAMES) T = MEL_MAX_FRAMES;           // cap (:304)
    const int t0 = blockIdx.x * 64;
    if (t0 >= T) return;
    const float floor_v = ordered_f32(gmax[b]) - 8.0f;
    const float* src = raw + (long)frame_off[b] * MEL_NMELS;
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * MEL_NMELS; i += 256) {
        int tt = i >> 7, m = i & 127;
        float v = (t0 + tt < T) ? src[(long)(t0 + tt) * MEL_NMELS + m] : 0.0f;
        tile[tt][m] = fmaxf(v, floor_v) * 0.25f + 1.0f;
    }
    __syncthreads();
    float* dst = out + (long)b * MEL_NMELS * out_stride;
    for (int i = tid; i < 64 * MEL_NMELS; i += 256) {
        int m = i >> 6, tt = i & 63;
        if (t0 + tt < T) dst[(long)m * out_stride + t0 + tt] = tile[tt][m];
    }
}

void mel_launch(const MelTables& t, const MelBatch& b, hipStream_t s) {
    if (b.B <= 0) return;
    QASR_HIP(hipMemsetAsync(b.gmax, 0, sizeof(unsigned) * b.B, s));
    dim3 g1(cdiv(b.max_frames_all, MEL_WAVES * MEL_FPW), b.B);
    hipLaunchKernelGGL(mel_frames_kernel, g1, dim3(MEL_WAVES * 64), 0, s, t.dev, b.pcm, b.pcm_off,
                       b.n_samples, b.frame_off, b.raw, b.gmax);
    dim3 g2(cdiv(b.max_frames_all, 64), b.B);
    hipLaunchKernelGGL(mel_finalize_kernel, g2, dim3(256), 0, s, b.raw, b.gmax, b.n_samples,
                       b.frame_off, b.out, b.out_stride);
    QASR_HIP(hipGetLastError());
}

}  // namespace qasr

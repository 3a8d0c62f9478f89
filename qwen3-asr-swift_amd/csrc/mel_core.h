// mel_core.h -- the pieces the two log-mel front-ends share (mel.hip: WhisperFeatureExtractor of the Qwen3-ASR path; nemo_mel.hip:
// the NeMo-style front-ends of Parakeet / Nemotron): constant-table layout, the 512-point real FFT of one frame per wavefront
// (256-point complex Stockham radix-4 in LDS + even/odd split), the sparse slaney filterbank.  Both reference front-ends build the
// same filterbank (AudioPreprocessing.swift:61-164 and ParakeetASR/MelPreprocessor.swift:207-266 are the same arithmetic, checked
// bit for bit on the oracle side) on the same 512-point grid; they differ in window, padding, pre-emphasis, log and normalisation.
#pragma once
#include "common.h"
#include <math.h>
#include <vector>

namespace qasr {

// ---- table layout (float words) -----------------------------------------------------------
constexpr int T_HANN = 0;                 // [512] window over the FFT frame (zero where the frame is zero-padded)
constexpr int T_TW256 = 512;              // [256][2] cos,sin(-2 pi k/256)
constexpr int T_TW512 = T_TW256 + 512;    // [257][2] cos,sin(-2 pi k/512) (padded to 520)
constexpr int T_FBSTART = T_TW512 + 520;  // [128] int: first bin of mel m
constexpr int T_FBLEN = T_FBSTART + 128;  // [128] int: number of bins
constexpr int T_FBWOFF = T_FBLEN + 128;   // [128] int: offset into packed weights
constexpr int T_FBW = T_FBWOFF + 128;     // [FBW_CAP] packed weights
constexpr int FBW_CAP = 640;
constexpr int T_SCALE2 = T_FBW + FBW_CAP; // [1] power scale (fft_scale^2, x 0.25 where the reference divides by 4)
constexpr int T_TOTAL = T_SCALE2 + 8;
constexpr int MELC_NMELS = 128, MELC_NBINS = 257;

inline float melc_hz_to_mel(float hz) {   // AudioPreprocessing.swift:72-78 = MelPreprocessor.swift:218-220 (Float32)
    if (hz < 1000.0f) return 3.0f * hz / 200.0f;
    return 15.0f + logf(hz / 1000.0f) * (27.0f / logf(6.4f));
}
inline float melc_mel_to_hz(float mel) {  // :80-86 = :222-224
    if (mel < 15.0f) return 200.0f * mel / 3.0f;
    return 1000.0f * expf((mel - 15.0f) * (logf(6.4f) / 27.0f));
}

// twiddles + sparse slaney filterbank on the 512-point grid + the power scale; the window slots [0, 512) are the caller's
inline void melc_fill_tables(std::vector<float>& t, float power_scale) {
    for (int k = 0; k < 256; ++k) {
        double a = -2.0 * M_PI * k / 256.0;
        t[T_TW256 + 2 * k] = (float)cos(a);
        t[T_TW256 + 2 * k + 1] = (float)sin(a);
    }
    for (int k = 0; k <= 256; ++k) {
        double a = -2.0 * M_PI * k / 512.0;
        t[T_TW512 + 2 * k] = (float)cos(a);
        t[T_TW512 + 2 * k + 1] = (float)sin(a);
    }
    const int npts = MELC_NMELS + 2;
    float mel_min = melc_hz_to_mel(0.0f), mel_max = melc_hz_to_mel(16000.0f / 2.0f);
    std::vector<float> filt(npts), diff(npts - 1);
    for (int i = 0; i < npts; ++i) filt[i] = melc_mel_to_hz(mel_min + (float)i * (mel_max - mel_min) / (float)(npts - 1));
    for (int i = 0; i < npts - 1; ++i) diff[i] = filt[i + 1] - filt[i];
    int* fb_start = reinterpret_cast<int*>(&t[T_FBSTART]);
    int* fb_len = reinterpret_cast<int*>(&t[T_FBLEN]);
    int* fb_woff = reinterpret_cast<int*>(&t[T_FBWOFF]);
    int w = 0;
    for (int m = 0; m < MELC_NMELS; ++m) {
        float enorm = 2.0f / (filt[m + 2] - filt[m]);
        int first = -1, last = -1;
        std::vector<float> row(MELC_NBINS);
        for (int k = 0; k < MELC_NBINS; ++k) {
            float f = (float)k * 16000.0f / 512.0f;
            float down = (f - filt[m]) / diff[m];
            float up = (filt[m + 2] - f) / diff[m + 1];
            float v = fmaxf(0.0f, fminf(down, up)) * enorm;
            row[k] = v;
            if (v != 0.0f) { if (first < 0) first = k; last = k; }
        }
        fb_start[m] = first < 0 ? 0 : first;
        fb_len[m] = first < 0 ? 0 : last - first + 1;
        fb_woff[m] = w;
        for (int k = fb_start[m]; k < fb_start[m] + fb_len[m]; ++k) {
            if (w >= FBW_CAP) throw std::runtime_error("mel filterbank exceeds FBW_CAP");
            t[T_FBW + w++] = row[k];
        }
    }
    t[T_SCALE2] = power_scale;
}

struct cplx { float re, im; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// One wavefront, one frame: v[r] = packed complex point lane + 64 r of the windowed 512-sample frame (samples 2p, 2p + 1).
// 256-point complex FFT (Stockham radix-4, Ns = 1, 4, 16, 64, ping-pong between the wave's two LDS buffers), split into the 512-point
// real spectrum, power x scale2 -> pw[0..256].  Contains workgroup barriers: every wave of the workgroup must call it together.
__device__ __forceinline__ void melc_frame_power(cplx v[4], int lane, float2* bufA, float2* bufB, float* pw, const float2* tw256,
                                                 const float2* tw512, float scale2) {
    float2* src = bufA;
    float2* dst = bufB;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int Ns = 1 << (2 * pass);
        if (pass > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { float2 t = src[lane + 64 * r]; v[r] = {t.x, t.y}; }
        }
        const int k = lane & (Ns - 1);
        const int tstep = k * (64 / Ns);              // twiddle w256^(tstep * r)
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            float2 t = tw256[(tstep * r) & 255];
            v[r] = cmul(v[r], {t.x, t.y});
        }
        cplx t0 = {v[0].re + v[2].re, v[0].im + v[2].im};
        cplx t1 = {v[0].re - v[2].re, v[0].im - v[2].im};
        cplx t2 = {v[1].re + v[3].re, v[1].im + v[3].im};
        cplx t3 = {v[1].im - v[3].im, -(v[1].re - v[3].re)};     // (v1 - v3) * (-i)
        const int base = (lane / Ns) * Ns * 4 + k;
        dst[base] = make_float2(t0.re + t2.re, t0.im + t2.im);
        dst[base + Ns] = make_float2(t1.re + t3.re, t1.im + t3.im);
        dst[base + 2 * Ns] = make_float2(t0.re - t2.re, t0.im - t2.im);
        dst[base + 3 * Ns] = make_float2(t1.re - t3.re, t1.im - t3.im);
        __syncthreads();
        float2* tmp = src; src = dst; dst = tmp;
    }
    // `src` now holds Z[0..255]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = lane + 64 * r;
        float2 zk = src[k];
        float2 zn = src[(256 - k) & 255];
        float er = 0.5f * (zk.x + zn.x), ei = 0.5f * (zk.y - zn.y);        // E = (Z[k] + conj Z[N-k]) / 2
        float orr = 0.5f * (zk.y + zn.y), oi = -0.5f * (zk.x - zn.x);      // O = (Z[k] - conj Z[N-k]) / (2i)
        float2 w = tw512[k];
        float xr = er + (orr * w.x - oi * w.y);
        float xi = ei + (orr * w.y + oi * w.x);
        pw[k] = (xr * xr + xi * xi) * scale2;
        if (k == 0) {                                                      // Nyquist: E[0] - O[0]
            float nr = er - orr;
            pw[256] = nr * nr * scale2;
        }
    }
    __syncthreads();
}

// mel bin m of the frame whose power spectrum is in pw: sparse slaney triangle (every FFT bin feeds <= 2 triangles)
__device__ __forceinline__ float melc_filter(const float* s_tab, const float* pw, int m) {
    const int* fb_start = reinterpret_cast<const int*>(&s_tab[T_FBSTART]);
    const int* fb_len = reinterpret_cast<const int*>(&s_tab[T_FBLEN]);
    const int* fb_woff = reinterpret_cast<const int*>(&s_tab[T_FBWOFF]);
    const int s0 = fb_start[m], len = fb_len[m], wo = fb_woff[m];
    float acc = 0.0f;
    for (int i = 0; i < len; ++i) acc += pw[s0 + i] * s_tab[T_FBW + wo + i];
    return acc;
}

}  // namespace qasr

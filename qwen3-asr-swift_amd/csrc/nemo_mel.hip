// nemo_mel.hip -- batched NeMo-style log-mel front-ends for gfx950 (Parakeet-TDT, Parakeet-EOU, Nemotron streaming; nemo_mel.h).
//
// Reference pipeline (ParakeetASR/MelPreprocessor.swift:52-202 and its streaming copies): pre-emphasis 0.97 | centre padding 256
// (reflect or zeros) | frames of 512 @ hop 160, Hann[400] at offset 0 or 56 | real FFT | power (x4 from vDSP, or x1 where the Swift code
// divides by 4) | slaney mel 128 | ln(x + 2^-24) | optional per-feature normalisation over the first n / 160 frames | [128, frames].
//
// Kernel 1 (nemo_frames): one wavefront per frame, the FFT core of mel_core.h.  Pre-emphasis and padding are index arithmetic on the
// sample reads (each PCM sample is read ~3.2 x 2 times through L2: 512-sample frames at hop 160, two taps); the window table covers the
// whole 512-sample frame, so the left-aligned and the centred placement are the same code.  HBM-bound in principle (n x 4 bytes in,
// frames x 512 bytes out); at the streaming shape (64 streams x 18 frames) the launch is latency-bound and the win is batching:
// one launch for every stream's chunk instead of 64 CPU calls.
// Kernel 2 (nemo_stats): per (row, mel bin) mean and 1 / (std + 1e-5) over the valid frames in the reference's two-pass form, or from
// the stream's running sums (EOU_STREAMING).  Kernel 3 (nemo_finalize): normalise, zero past melLength, transpose [frame][mel] ->
// [mel][frame] through an LDS tile, fit to the requested frame count.
#include "nemo_mel.h"
#include "mel_core.h"
#include <cstring>

namespace qasr {

constexpr int NM_WAVES = 4;
constexpr float NM_GUARD = 5.960464477539063e-08f;     // 2^-24 (MelPreprocessor.swift:18)
constexpr float NM_PREEMPH = 0.97f;

struct NemoMeta {              // device pointers into one meta block
    const long* pcm_off;       // [B]
    const int* n_samples;      // [B]
    const int* frame_off;      // [B] row offset into raw
    const int* stream_id;      // [B]
};

// pre-emphasised sample j of a clip of n samples (MelPreprocessor.swift:54-65: y[0] = x[0], y[j] = x[j] + (-0.97) x[j-1])
__device__ __forceinline__ float nm_pre(const float* __restrict__ x, long j) {
    const float a = x[j];
    return j == 0 ? a : x[j - 1] * (-NM_PREEMPH) + a;
}

template <bool REFLECT>
__global__ __launch_bounds__(NM_WAVES * 64) void nemo_frames_kernel(const float* __restrict__ tab, const float* __restrict__ pcm, NemoMeta meta,
                                                                    float* __restrict__ raw, int fpw) {
    __shared__ float s_tab[T_TOTAL];
    __shared__ float2 s_buf[NM_WAVES][2][256];
    __shared__ float s_pow[NM_WAVES][260];
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < T_TOTAL; i += NM_WAVES * 64) s_tab[i] = tab[i];
    __syncthreads();
    const long n = meta.n_samples[b];
    const int nf = (int)(n / NEMO_HOP) + 1;
    const float* x = pcm + meta.pcm_off[b];
    float* out = raw + (long)meta.frame_off[b] * NEMO_NMELS;
    const float2* tw256 = reinterpret_cast<const float2*>(&s_tab[T_TW256]);
    const float2* tw512 = reinterpret_cast<const float2*>(&s_tab[T_TW512]);
    const float scale2 = s_tab[T_SCALE2];
    const int frame0 = (blockIdx.x * NM_WAVES + wave) * fpw;
    for (int fi = 0; fi < fpw; ++fi) {
        const int frame = frame0 + fi;
        const bool live = frame < nf;                      // barriers inside melc_frame_power stay uniform
        cplx v[4];
        const long start = (long)frame * NEMO_HOP - NEMO_PAD;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = lane + 64 * r;
            const float w0 = s_tab[T_HANN + 2 * p], w1 = s_tab[T_HANN + 2 * p + 1];
            float a0 = 0.0f, a1 = 0.0f;
            if (live && (w0 != 0.0f || w1 != 0.0f)) {
                long i0 = start + 2 * p, i1 = i0 + 1;
                if (REFLECT) {
                    // padded[k] = pre[256 - k], padded[256 + n + k] = pre[max(0, n - 2 - k)] (MelPreprocessor.swift:68-83); the left
                    // mirror needs n > 256 (the Swift code indexes out of bounds below that; the host refuses such clips)
                    long j0 = i0 < 0 ? -i0 : (i0 >= n ? 2 * n - 2 - i0 : i0);
                    long j1 = i1 < 0 ? -i1 : (i1 >= n ? 2 * n - 2 - i1 : i1);
                    j0 = j0 < 0 ? 0 : j0;
                    j1 = j1 < 0 ? 0 : j1;
                    a0 = nm_pre(x, j0) * w0;
                    a1 = nm_pre(x, j1) * w1;
                } else {                                   // zeros outside the clip (pad_mode "constant")
                    if (i0 >= 0 && i0 < n) a0 = nm_pre(x, i0) * w0;
                    if (i1 >= 0 && i1 < n) a1 = nm_pre(x, i1) * w1;
                }
            }
            v[r] = {a0, a1};
        }
        melc_frame_power(v, lane, s_buf[wave][0], s_buf[wave][1], s_pow[wave], tw256, tw512, scale2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = lane + 64 * h;
            const float acc = melc_filter(s_tab, s_pow[wave], m);
            if (live) out[(long)frame * NEMO_NMELS + m] = logf(acc + NM_GUARD);
        }
        __syncthreads();
    }
}

// stats [B][2][128]: mean, 1 / (std + 1e-5).  One workgroup per row, thread = (mel bin, one of 8 frame phases).
// mode 0: TDT (denominator L - 1), 1: EOU (max(L - 1, 1)), 2: running sums of the row's stream (run [streams][257]: sum[128], sumsq[128], count).
__global__ __launch_bounds__(1024) void nemo_stats_kernel(const float* __restrict__ raw, NemoMeta meta, float* __restrict__ stats,
                                                          float* __restrict__ run, int mode) {
    __shared__ float red[8][NEMO_NMELS];
    __shared__ float red2[8][NEMO_NMELS];
    const int b = blockIdx.x, m = threadIdx.x & 127, ph = threadIdx.x >> 7;
    const long n = meta.n_samples[b];
    const int nf = (int)(n / NEMO_HOP) + 1;
    const int L = (int)(n / NEMO_HOP);
    const int valid = L < nf ? L : nf;
    const float* src = raw + (long)meta.frame_off[b] * NEMO_NMELS + m;
    float s = 0.0f, s2 = 0.0f;
    for (int t = ph; t < valid; t += 8) {
        const float v = src[(long)t * NEMO_NMELS];
        s += v;
        if (mode == 2) s2 += v * v;
    }
    red[ph][m] = s;
    red2[ph][m] = s2;
    __syncthreads();
    float mean = 0.0f, inv = 0.0f;
    if (mode == 2) {
        float* r = run + (long)meta.stream_id[b] * 257;
        if (ph == 0) {
            float cs = 0.0f, cs2 = 0.0f;
            for (int i = 0; i < 8; ++i) { cs += red[i][m]; cs2 += red2[i][m]; }
            const float rs = r[m] + cs, rs2 = r[128 + m] + cs2;                 // runningSum / runningSumSq (:352-358)
            const float cnt = r[256] + (float)valid;                             // read before thread 0 updates it below
            r[m] = rs;
            r[128 + m] = rs2;
            const float nn = fmaxf(cnt, 1.0f);
            mean = rs / nn;
            const float var = fmaxf(rs2 / nn - mean * mean, 0.0f);
            const float sd = sqrtf(var * nn / fmaxf(nn - 1.0f, 1.0f));
            inv = 1.0f / (sd + 1e-5f);
            stats[((long)b * 2) * NEMO_NMELS + m] = mean;
            stats[((long)b * 2 + 1) * NEMO_NMELS + m] = inv;
        }
        __syncthreads();
        if (threadIdx.x == 0) r[256] += (float)valid;                            // runningCount (exact in f32 below 2^24 frames = 46 hours)
        return;
    }
    float tot = 0.0f;
    for (int i = 0; i < 8; ++i) tot += red[i][m];
    mean = L > 0 ? tot / (float)L : 0.0f;                                        // vDSP_meanv over melLength
    __syncthreads();
    float q = 0.0f;
    for (int t = ph; t < valid; t += 8) {
        const float c = src[(long)t * NEMO_NMELS] - mean;
        q += c * c;
    }
    red[ph][m] = q;
    __syncthreads();
    if (ph == 0) {
        float qq = 0.0f;
        for (int i = 0; i < 8; ++i) qq += red[i][m];
        const float meansq = L > 0 ? qq / (float)L : 0.0f;                       // vDSP_measqv of the centred values
        const float den = mode == 0 ? (float)(L - 1) : (float)(L - 1 > 1 ? L - 1 : 1);
        const float sd = sqrtf((float)L * meansq / den);
        inv = 1.0f / (sd + 1e-5f);
        stats[((long)b * 2) * NEMO_NMELS + m] = mean;
        stats[((long)b * 2 + 1) * NEMO_NMELS + m] = inv;
    }
}

// raw [frames][128] -> out [b][128][stride]: frames t < lim written, t in [lim, fit) zero.  norm: (v - mean) * inv for t < melLength, 0 after.
__global__ __launch_bounds__(256) void nemo_finalize_kernel(const float* __restrict__ raw, NemoMeta meta, const float* __restrict__ stats,
                                                            float* __restrict__ out, int stride, int fit, int norm, int f16) {
    __shared__ float tile[64][NEMO_NMELS + 1];
    const int b = blockIdx.y;
    const long n = meta.n_samples[b];
    const int nf = (int)(n / NEMO_HOP) + 1, L = (int)(n / NEMO_HOP);
    const int t0 = blockIdx.x * 64;
    if (t0 >= fit) return;
    const float* src = raw + (long)meta.frame_off[b] * NEMO_NMELS;
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * NEMO_NMELS; i += 256) {
        const int tt = i >> 7, m = i & 127, t = t0 + tt;
        float v = 0.0f;
        if (t < nf && t < fit) {
            v = src[(long)t * NEMO_NMELS + m];
            if (norm) v = t < L ? (v - stats[((long)b * 2) * NEMO_NMELS + m]) * stats[((long)b * 2 + 1) * NEMO_NMELS + m] : 0.0f;
            if (f16) v = (float)(_Float16)v;
        }
        tile[tt][m] = v;
    }
    __syncthreads();
    float* dst = out + (long)b * NEMO_NMELS * stride;
    for (int i = tid; i < 64 * NEMO_NMELS; i += 256) {
        const int m = i >> 6, tt = i & 63;
        if (t0 + tt < fit) dst[(long)m * stride + t0 + tt] = tile[tt][m];
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
static void build_table(DevBuf& d, bool periodic, int offset, float power_scale) {
    std::vector<float> t(T_TOTAL, 0.0f);
    const float den = periodic ? 400.0f : 399.0f;            // MelPreprocessor.swift:27-31 | StreamingMelPreprocessor.swift:31-38
    for (int i = 0; i < 400; ++i) t[T_HANN + offset + i] = 0.5f * (1.0f - cosf(2.0f * (float)M_PI * (float)i / den));
    melc_fill_tables(t, power_scale);
    d.alloc(T_TOTAL * sizeof(float));
    QASR_HIP(hipMemcpy(d.p, t.data(), T_TOTAL * sizeof(float), hipMemcpyHostToDevice));
}

NemoMel::NemoMel(int device, int max_streams, long max_samples, float fft_scale)
    : device_(device), max_streams_(max_streams), max_samples_(max_samples) {
    if (max_streams <= 0 || max_samples <= 0 || max_streams > 4096 || max_samples > 16000L * 1200 || !(fft_scale > 0.0f))
        throw std::invalid_argument("nemo mel: max_streams in 1..4096, max_samples in 1..1200 s, fft_scale > 0");
    QASR_HIP(hipSetDevice(device_));
    QASR_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    for (auto& e : ev_) QASR_HIP(hipEventCreate(&e));
    build_table(d_tab_[0], true, 0, fft_scale * fft_scale);
    build_table(d_tab_[1], false, 0, fft_scale * fft_scale);
    // extractRaw multiplies by 0.25 BECAUSE vDSP returns 2 X ("vDSP_fft_zrip scales 2x vs torch.stft -- divide power by 4",
    // NemotronStreamingASR/StreamingMelPreprocessor.swift:100): its power spectrum is the textbook |X|^2 whatever one assumes about vDSP
    build_table(d_tab_[2], false, 56, 1.0f);
    const size_t B = (size_t)max_streams_;
    const size_t pcm_bytes = B * (size_t)(max_samples_ + 2) * sizeof(float);
    h_pcm_.alloc(pcm_bytes);
    d_pcm_.alloc(pcm_bytes);
    const size_t meta = B * (sizeof(long) + 3 * sizeof(int)) + 64;
    h_meta_.alloc(meta);
    d_meta_.alloc(meta);
    d_raw_.alloc(B * (size_t)nemo_num_frames(max_samples_) * NEMO_NMELS * sizeof(float));
    d_stats_.alloc(B * 2 * NEMO_NMELS * sizeof(float));
    d_run_.alloc(B * 257 * sizeof(float));
    QASR_HIP(hipMemset(d_run_.p, 0, d_run_.bytes));
}

NemoMel::~NemoMel() {
    if (stream_) (void)hipStreamSynchronize(stream_);
    drop_graph();
    for (auto& e : ev_) if (e) (void)hipEventDestroy(e);
    if (stream_) (void)hipStreamDestroy(stream_);
}

void NemoMel::drop_graph() {
    if (graph_) (void)hipGraphExecDestroy(graph_);
    graph_ = nullptr;
    for (auto& k : graph_key_) k = -1;
}

void NemoMel::reset_stats(int stream) {
    if (stream >= max_streams_) throw std::invalid_argument("nemo mel: stream outside [0, max_streams)");
    QASR_HIP(hipStreamSynchronize(stream_));
    if (stream < 0) QASR_HIP(hipMemset(d_run_.p, 0, d_run_.bytes));
    else QASR_HIP(hipMemset(d_run_.as<float>() + (size_t)stream * 257, 0, 257 * sizeof(float)));
}

// everything of one call that touches the device, in stream order: meta + PCM up, three kernels, result down
void NemoMel::issue(int variant, int B, int max_frames, int fit, size_t stride, size_t pcm_elems, hipStream_t s) {
    char* dm = d_meta_.as<char>();
    NemoMeta meta;
    meta.pcm_off = reinterpret_cast<const long*>(dm);
    meta.n_samples = reinterpret_cast<const int*>(dm + (size_t)B * sizeof(long));
    meta.frame_off = meta.n_samples + B;
    meta.stream_id = meta.frame_off + B;
    QASR_HIP(hipMemcpyAsync(d_meta_.p, h_meta_.p, (size_t)B * (sizeof(long) + 3 * sizeof(int)), hipMemcpyHostToDevice, s));
    QASR_HIP(hipMemcpyAsync(d_pcm_.p, h_pcm_.p, pcm_elems * sizeof(float), hipMemcpyHostToDevice, s));
    const bool reflect = variant != NEMO_MEL_RAW;
    const float* tab = d_tab_[variant == NEMO_MEL_TDT ? 0 : (variant == NEMO_MEL_RAW ? 2 : 1)].as<float>();
    // frames per wave: one at streaming shapes (every frame its own wave: 64 x 18 frames = 288 workgroups), eight on long clips
    const int fpw = max_frames <= 64 ? 1 : 8;
    dim3 g1(cdiv(max_frames, NM_WAVES * fpw), B);
    if (reflect) hipLaunchKernelGGL(nemo_frames_kernel<true>, g1, dim3(NM_WAVES * 64), 0, s, tab, d_pcm_.as<float>(), meta, d_raw_.as<float>(), fpw);
    else hipLaunchKernelGGL(nemo_frames_kernel<false>, g1, dim3(NM_WAVES * 64), 0, s, tab, d_pcm_.as<float>(), meta, d_raw_.as<float>(), fpw);
    const int norm = variant != NEMO_MEL_RAW;
    if (norm)
        hipLaunchKernelGGL(nemo_stats_kernel, dim3(B), dim3(1024), 0, s, d_raw_.as<float>(), meta, d_stats_.as<float>(), d_run_.as<float>(),
                           variant == NEMO_MEL_TDT ? 0 : (variant == NEMO_MEL_EOU ? 1 : 2));
    hipLaunchKernelGGL(nemo_finalize_kernel, dim3(cdiv(fit, 64), B), dim3(256), 0, s, d_raw_.as<float>(), meta, d_stats_.as<float>(),
                       d_out_.as<float>(), (int)stride, fit, norm, variant == NEMO_MEL_TDT ? 1 : 0);
    QASR_HIP(hipMemcpyAsync(h_out_.p, d_out_.p, (size_t)B * NEMO_NMELS * stride * sizeof(float), hipMemcpyDeviceToHost, s));
}

void NemoMel::extract(int variant, const float* const* pcm, const size_t* n, size_t B, const int32_t* stream_ids, float* out, size_t stride,
                      int32_t* mel_len, int fit) {
    if (variant < 0 || variant > 3) throw std::invalid_argument("nemo mel: unknown variant");
    if (B == 0) return;
    if ((int)B > max_streams_) throw std::length_error("nemo mel: batch exceeds max_streams");
    QASR_HIP(hipSetDevice(device_));
    long* h_off = h_meta_.as<long>();
    int* h_ns = reinterpret_cast<int*>(h_off + B);
    int* h_fo = h_ns + B;
    int* h_sid = h_fo + B;
    long off = 0;
    int frames_total = 0, max_frames = 0;
    bool uniform = true;
    for (size_t b = 0; b < B; ++b) {
        if (!pcm[b] || n[b] == 0) throw std::invalid_argument("nemo mel: empty clip (the host wrapper answers the reference's [1,128,1] zero array)");
        if ((long)n[b] > max_samples_) throw std::length_error("nemo mel: clip longer than max_samples");
        if (variant != NEMO_MEL_RAW && n[b] <= (size_t)NEMO_PAD)
            throw std::invalid_argument("nemo mel: reflect padding needs more than 256 samples (the reference indexes out of bounds)");
        const int sid = stream_ids ? stream_ids[b] : (int)b;
        if (sid < 0 || sid >= max_streams_) throw std::invalid_argument("nemo mel: stream id outside [0, max_streams)");
        if (variant == NEMO_MEL_EOU_STREAMING)
            for (size_t c = 0; c < b; ++c)
                if (h_sid[c] == sid) throw std::invalid_argument("nemo mel: one chunk per stream and call (running statistics are sequential)");
        h_off[b] = off; h_ns[b] = (int)n[b]; h_fo[b] = frames_total; h_sid[b] = sid;
        std::memcpy(h_pcm_.as<float>() + off, pcm[b], n[b] * sizeof(float));
        off += (long)((n[b] + 1) & ~(size_t)1);
        const int nf = nemo_num_frames((long)n[b]);
        frames_total += nf;
        max_frames = nf > max_frames ? nf : max_frames;
        uniform = uniform && n[b] == n[0];
        if (mel_len) mel_len[b] = nemo_mel_length((long)n[b]);
    }
    if (fit <= 0) fit = max_frames;
    if (stride < (size_t)fit) throw std::invalid_argument("nemo mel: stride smaller than the frame count");
    const size_t out_elems = B * NEMO_NMELS * stride;
    if (out_elems > out_cap_) {
        QASR_HIP(hipStreamSynchronize(stream_));
        drop_graph();
        d_out_.alloc(out_elems * sizeof(float));
        h_out_.alloc(out_elems * sizeof(float));
        out_cap_ = out_elems;
    }
    QASR_HIP(hipEventRecord(ev_[0], stream_));
    last_graph_ = false;
    if (uniform) {
        // same variant / batch / chunk length / geometry as the captured call: the pinned staging buffers hold this call's samples and
        // stream ids, everything else is identical -> replay (the 64-stream x 160 ms-hop loop of configs[4])
        const long key[6] = {variant, (long)B, (long)n[0], fit, (long)stride, (long)off};
        if (!graph_ || std::memcmp(key, graph_key_, sizeof(key)) != 0) {
            drop_graph();
            hipGraph_t g = nullptr;
            QASR_HIP(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
            try { issue(variant, (int)B, max_frames, fit, stride, (size_t)off, stream_); }
            catch (...) { (void)hipStreamEndCapture(stream_, &g); if (g) (void)hipGraphDestroy(g); throw; }
            QASR_HIP(hipStreamEndCapture(stream_, &g));
            hipError_t e = hipGraphInstantiate(&graph_, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (e != hipSuccess) { graph_ = nullptr; QASR_HIP(e); }
            std::memcpy(graph_key_, key, sizeof(key));
        }
        QASR_HIP(hipGraphLaunch(graph_, stream_));
        last_graph_ = true;
    } else {
        issue(variant, (int)B, max_frames, fit, stride, (size_t)off, stream_);
    }
    QASR_HIP(hipEventRecord(ev_[1], stream_));
    QASR_HIP(hipStreamSynchronize(stream_));
    QASR_HIP(hipGetLastError());
    QASR_HIP(hipEventElapsedTime(&last_ms_, ev_[0], ev_[1]));
    std::memcpy(out, h_out_.p, out_elems * sizeof(float));
}

}  // namespace qasr

// nemo_mel.h -- batched NeMo-style log-mel front-ends of the Parakeet / Nemotron models (BASELINE configs[4]): kernels in nemo_mel.hip.
//
// The reference computes these on the CPU with Accelerate, one clip or one 160 ms chunk at a time:
//   variant TDT            ParakeetASR/MelPreprocessor.swift:52-202                    periodic Hann, reflect pad, per-feature norm, float16
//   variant EOU            ParakeetStreamingASR/StreamingMelPreprocessor.swift:62-186  symmetric Hann, reflect pad, per-feature norm
//   variant RAW            NemotronStreamingASR/StreamingMelPreprocessor.swift:55-129  symmetric Hann centred in the frame, zero pad, power / 4
//                          (= ParakeetStreamingASR/StreamingMelPreprocessor.swift:193-273)
//   variant EOU_STREAMING  ParakeetStreamingASR/StreamingMelPreprocessor.swift:280-393 as EOU with mean / std from sums kept per stream
// Everything after the front-end (FastConformer encoder, prediction network, joint) is an opaque CoreML bundle in the reference and is
// NOT built here; the greedy loops that drive those networks are host code in transducer.cpp.
#pragma once
#include "engine.h"

namespace qasr {

enum { NEMO_MEL_TDT = 0, NEMO_MEL_EOU = 1, NEMO_MEL_RAW = 2, NEMO_MEL_EOU_STREAMING = 3 };
constexpr int NEMO_HOP = 160, NEMO_NMELS = 128, NEMO_PAD = 256;

inline int nemo_num_frames(long n) { return (int)(n / NEMO_HOP) + 1; }     // (n + 2 * 256 - 512) / 160 + 1
inline int nemo_mel_length(long n) { return (int)(n / NEMO_HOP); }          // NeMo: floor(samples / hop)

class NemoMel {
  public:
    NemoMel(int device, int max_streams, long max_samples, float fft_scale);
    ~NemoMel();
    // B clips / chunks -> out [B][128][stride] float32 in host memory (TDT: values rounded to float16 like the reference's output
    // array), mel_len[b] = n[b] / 160.  Frames t < min(nFrames, fit) are written (normalised variants: zero from melLength on), frames
    // up to `fit` zero-filled (StreamingSession.truncateMel / padMel); fit <= 0: every frame of the longest clip.
    // stream_ids (EOU_STREAMING only): which running-statistics slot each row updates and reads; NULL = row index.
    void extract(int variant, const float* const* pcm, const size_t* n, size_t B, const int32_t* stream_ids, float* out, size_t stride,
                 int32_t* mel_len, int fit);
    void reset_stats(int stream);                      // resetRunningStats; stream < 0: all
    float last_ms() const { return last_ms_; }         // device time of the last extract (H2D + kernels + D2H), HIP events
    bool last_was_graph() const { return last_graph_; }
    int max_streams() const { return max_streams_; }
    long max_samples() const { return max_samples_; }

  private:
    void issue(int variant, int B, int max_frames, int fit, size_t stride, size_t pcm_elems, hipStream_t s);
    void drop_graph();
    int device_, max_streams_;
    long max_samples_;
    hipStream_t stream_ = nullptr;
    hipEvent_t ev_[2] = {};
    DevBuf d_tab_[3];                                  // window tables: periodic left-aligned | symmetric left-aligned | symmetric centred (/4)
    DevBuf d_pcm_, d_meta_, d_raw_, d_stats_, d_run_, d_out_;
    HostBuf h_pcm_, h_meta_, h_out_;
    size_t out_cap_ = 0;
    float last_ms_ = 0.f;
    bool last_graph_ = false;
    // fixed-shape calls (64 streams x one 160 ms chunk) replay one captured graph: H2D, three kernels, D2H
    hipGraphExec_t graph_ = nullptr;
    long graph_key_[6] = {-1, -1, -1, -1, -1, -1};
};

}  // namespace qasr

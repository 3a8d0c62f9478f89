// mel.h -- log-mel front-end launcher (kernels in mel.hip).
#pragma once
#include "common.h"

namespace qasr {

constexpr int MEL_SR = 16000, MEL_NFFT = 400, MEL_HOP = 160, MEL_NMELS = 128, MEL_PADDED = 512;
constexpr int MEL_NBINS = 257, MEL_MAX_FRAMES = 120000;

inline int mel_num_frames_all(long n) { return (int)(n / MEL_HOP) + 1; }              // before drop-last
inline int mel_num_frames(long n) {                                                   // handed to the encoder
    int f = mel_num_frames_all(n) - 1;
    return f > MEL_MAX_FRAMES ? MEL_MAX_FRAMES : f;
}

// Constant tables (Hann window, twiddles, sparse slaney filterbank) resident in HBM.
struct MelTables {
    float* dev = nullptr;      // one allocation, layout documented in mel.hip
    size_t bytes = 0;
    void build(float fft_scale);
    void release();
};

// One batch of ragged clips.  All pointers are device pointers.
struct MelBatch {
    const float* pcm;          // clips back to back, clip b at pcm + pcm_off[b]
    const long* pcm_off;       // [B] element offsets (even)
    const int* n_samples;      // [B]
    const int* frame_off;      // [B] row offset of clip b in `raw` (frames incl. the dropped one)
    int B;
    int max_frames_all;        // max over clips of frames incl. dropped
    float* raw;                // [sum frames_all, 128] log10 mel before clamp
    unsigned* gmax;            // [B] per-clip max (order-preserving uint encoding), zeroed by launcher
    float* out;                // [B][128][out_stride] final log-mel, reference layout per clip
    int out_stride;            // >= max T
};

void mel_launch(const MelTables& t, const MelBatch& b, hipStream_t s);

}  // namespace qasr

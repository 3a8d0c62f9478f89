// engine.hip -- engine lifecycle, batch planning, stage orchestration.
#include "engine.h"
#include <algorithm>
#include <cstring>
#include <thread>

namespace qasr {

static int conv_len(int n) { return (n - 1) / 2 + 1; }
static int tokens_for_chunk(int clen) { return conv_len(conv_len(conv_len(clen))); }

// Integer geometry of one clip.  References: AudioPreprocessing.swift:195,296,304 (frames);
// AudioEncoder.swift:367-380 (chunks), :443-449 (valid tokens), :464-478 (windows);
// Qwen3ASR.swift:199-233 (prompt length = 16 + tokens + context + language).
ClipPlan Engine::plan_clip(const qasr_config& cfg, long n, int extra_prompt) {
    ClipPlan c;
    c.n_samples = n;
    c.frames_all = mel_num_frames_all(n);
    c.frames = mel_num_frames(n);
    const int chunk = 2 * cfg.n_window;
    c.n_chunks = (c.frames + chunk - 1) / chunk;
    int rem = c.frames % chunk;
    c.last_chunk_len = c.n_chunks == 0 ? 0 : (rem == 0 ? chunk : rem);
    c.max_chunk_len = c.n_chunks > 1 ? chunk : c.last_chunk_len;
    c.n_tokens = 0;
    int max_after = 0;
    for (int i = 0; i < c.n_chunks; ++i) {
        int t = tokens_for_chunk(i == c.n_chunks - 1 ? c.last_chunk_len : chunk);
        c.n_tokens += t;
        max_after = std::max(max_after, t);
    }
    if (c.n_tokens > 0) {
        int window = max_after * (cfg.n_window_infer / chunk);
        for (int i = 0; i < c.n_tokens / window; ++i) c.windows.push_back(window);
        if (c.n_tokens % window) c.windows.push_back(c.n_tokens % window);
    }
    c.prompt_len = 16 + c.n_tokens + extra_prompt;
    return c;
}

Engine::Engine(const qasr_config& cfg) : cfg_(cfg) {
    if (cfg_.max_batch <= 0) throw std::invalid_argument("max_batch must be positive");
    // the decode step holds the batch rows of a launch in at most four 16-row MFMA tiles (dec_gemv.hip, dec_lmhead.hip, dec_quant.hip): refuse a
    // capacity the step could not serve here, not at the first qasr_batch_run.  Larger jobs go through the engine in passes (qasr_dp_*,
    // qasr/transcribe_batch.py); the forced aligner has no decode step and is not bounded by this.
    if (cfg_.classify_num == 0 && cfg_.max_batch > 64)
        throw std::invalid_argument("max_batch " + std::to_string(cfg_.max_batch) + " exceeds the decode step's 64 batch rows per engine; run larger jobs in passes of <= 64 clips (qasr_dp_transcribe_batch slices a block over its engines)");
    QASR_HIP(hipSetDevice(cfg_.device));
    QASR_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    mel_tables_.build(cfg_.fft_scale);
    max_samples_ = (long)cfg_.max_audio_seconds * MEL_SR;
    max_frames_all_ = mel_num_frames_all(max_samples_);
    const int B = cfg_.max_batch;
    // pcm: clips back to back, each padded to an even element count
    h_pcm_.alloc((size_t)B * (max_samples_ + 2) * sizeof(float));
    d_pcm_.alloc((size_t)B * (max_samples_ + 2) * sizeof(float));
    h_meta_.alloc((size_t)B * 64);
    d_meta_.alloc((size_t)B * 64);
    d_mel_raw_.alloc((size_t)B * max_frames_all_ * MEL_NMELS * sizeof(float));
    d_gmax_.alloc((size_t)B * sizeof(unsigned));
    mel_stride_ = ((max_frames_all_ + 63) / 64) * 64;
    d_mel_.alloc((size_t)B * MEL_NMELS * mel_stride_ * sizeof(float));
}

Engine::~Engine() {
    if (stream_) (void)hipStreamSynchronize(stream_);
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
    if (graph_exec_n_) (void)hipGraphExecDestroy(graph_exec_n_);
    for (auto& e : ev_) if (e) (void)hipEventDestroy(e);
    if (fork_ev_) (void)hipEventDestroy(fork_ev_);
    for (int i = 0; i < 3; ++i) {
        if (join_ev_[i]) (void)hipEventDestroy(join_ev_[i]);
        if (side_[i]) (void)hipStreamDestroy(side_[i]);
    }
    mel_tables_.release();
    if (copy_stream_) { (void)hipStreamSynchronize(copy_stream_); (void)hipStreamDestroy(copy_stream_); }
    if (ev_mel_done_) (void)hipEventDestroy(ev_mel_done_);
    if (ev_stage_done_) (void)hipEventDestroy(ev_stage_done_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

void Engine::set_tensor(const std::string& name, const void* host, int dtype, const int64_t* shape, int ndim) {
    Tensor& t = tensors_[name];
    t.shape.assign(shape, shape + ndim);
    t.dtype = dtype;
    size_t el = (dtype == QASR_DTYPE_F32 || dtype == QASR_DTYPE_U32) ? 4 : 2;
    // the captured graph and the fused copies may still be reading the buffer this replaces
    QASR_HIP(hipStreamSynchronize(stream_));
    drop_graph();
    finalized_ = false;
    t.buf.alloc(t.numel() * el);
    QASR_HIP(hipMemcpy(t.buf.p, host, t.numel() * el, hipMemcpyHostToDevice));
}

void Engine::drop_graph() {
    if (graph_exec_) { (void)hipGraphExecDestroy(graph_exec_); graph_exec_ = nullptr; }
    if (graph_exec_n_) { (void)hipGraphExecDestroy(graph_exec_n_); graph_exec_n_ = nullptr; }
    graph_n_ = 0;
    graph_key_ = -1;
}

// (Re-)finalize: everything derived from the tensors is rebuilt, so everything that points at the previous build
// goes first -- the captured decode graph holds raw pointers into the KV caches / packed weights, fused_ owns the
// previous packed copies, and a resident batch refers to the old caches.
void Engine::finalize() {
    QASR_HIP(hipStreamSynchronize(stream_));
    drop_graph();
    fused_.clear();
    batch_ = 0;
    staged_valid_ = false;
    run_issued_ = false;
    h_ctx0_.clear();
    forced_ctx_ = 0;
    finalized_ = false;
    finalize_encoder();
    finalize_decoder();
    finalized_ = true;
}
void Engine::unload() {
    QASR_HIP(hipStreamSynchronize(stream_));
    drop_graph();
    tensors_.clear();
    fused_.clear();
    kcache_.clear();
    vfcache_.clear();
    d_wscratch_.release();
    if (copy_stream_) QASR_HIP(hipStreamSynchronize(copy_stream_));
    staged_valid_ = false;
    run_issued_ = false;
    batch_ = 0;
    h_ctx0_.clear();
    forced_ctx_ = 0;
    finalized_ = false;
}
void Engine::require_batch(const char* what) const {
    if (!finalized_) throw NotLoaded(std::string(what) + ": weights not finalized (unloaded, or a tensor was replaced)");
    if (batch_ <= 0 || h_ctx0_.empty()) throw std::runtime_error(std::string(what) + ": no resident batch (call qasr_batch_begin)");
}
// Every weight byte resident in HBM: the uploaded tensors AND what qasr_finalize derived from them (fused q|k|v and gate|up, the
// fragment-major decode-step images, quantised decode images, the quantised engine's one-layer prompt-pass scratch).  Caches and
// activation workspaces are capacity, not parameters, and are not counted.
size_t Engine::memory_footprint() const {
    size_t n = 0;
    for (auto& kv : tensors_) n += kv.second.buf.bytes;
    for (auto& b : fused_) n += b->bytes;
    n += d_wscratch_.bytes;
    return n;
}

// Clips of one batch -> pinned staging `hp` (+ offsets / sample counts / frame offsets in `hm`) -> the device PCM and meta buffers, queued on
// `cs`.  Fills the clip plans; the caller adopts them.
void Engine::stage_pcm(const float* const* pcm, const size_t* n, size_t B, HostBuf& hp, HostBuf& hm, hipStream_t cs, std::vector<ClipPlan>& clips,
                       int& max_frames_all) {
    if ((int)B > cfg_.max_batch) throw std::invalid_argument("batch exceeds max_batch");
    clips.clear();
    long off = 0;
    long* h_off = hm.as<long>();
    int* h_ns = reinterpret_cast<int*>(h_off + B);
    int* h_fo = h_ns + B;
    int frame_off = 0;
    max_frames_all = 0;
    for (size_t b = 0; b < B; ++b) {
        if (!pcm[b] || n[b] == 0) throw std::invalid_argument("empty clip");
        if ((long)n[b] > max_samples_) throw std::length_error("clip longer than max_audio_seconds");
        ClipPlan c = plan_clip(cfg_, (long)n[b], 0);
        h_off[b] = off;
        h_ns[b] = (int)n[b];
        h_fo[b] = frame_off;
        off += ((long)n[b] + 1) & ~1L;
        frame_off += c.frames_all;
        max_frames_all = std::max(max_frames_all, c.frames_all);
        clips.push_back(std::move(c));
    }
    // caller memory (pageable) -> pinned staging -> HBM: 61 MB at 32 x 30 s, ~10 ms on one core and ~2.4 ms over PCIe.  Clips are
    // spread over a few threads (each owns whole clips; the staging buffer is private to this engine) and the batch goes in
    // up to four slices, the H2D copy of a slice queued as soon as it is staged: staging of slice i + 1 overlaps the copy of i.
    {
        const long total = off;
        const int nthr = total > (4L << 20) ? (int)std::min<size_t>(B, 8) : 1;
        const size_t nsl = total > (8L << 20) ? std::min<size_t>(B, 4) : 1;
        for (size_t sidx = 0; sidx < nsl; ++sidx) {
            const size_t s0 = B * sidx / nsl, s1 = B * (sidx + 1) / nsl;
            auto copy_range = [&](size_t b0, size_t b1) {
                for (size_t b = b0; b < b1; ++b) std::memcpy(hp.as<float>() + h_off[b], pcm[b], n[b] * sizeof(float));
            };
            const size_t nb = s1 - s0;
            const int t_n = (int)std::min<size_t>((size_t)nthr, nb);
            if (t_n <= 1) copy_range(s0, s1);
            else {
                std::vector<std::thread> pool;
                for (int t = 0; t < t_n; ++t) pool.emplace_back(copy_range, s0 + nb * t / t_n, s0 + nb * (t + 1) / t_n);
                for (auto& th : pool) th.join();
            }
            const long e0 = h_off[s0], e1 = s1 < B ? h_off[s1] : off;
            QASR_HIP(hipMemcpyAsync(d_pcm_.as<float>() + e0, hp.as<float>() + e0, (size_t)(e1 - e0) * sizeof(float), hipMemcpyHostToDevice, cs));
        }
    }
    size_t meta_bytes = B * (sizeof(long) + 2 * sizeof(int));
    QASR_HIP(hipMemcpyAsync(d_meta_.p, hm.p, meta_bytes, hipMemcpyHostToDevice, cs));
}

void Engine::upload_pcm(const float* const* pcm, const size_t* n, size_t B) {
    staged_valid_ = false;                         // a batch staged ahead would find its device buffers overwritten
    if (copy_stream_) QASR_HIP(hipStreamSynchronize(copy_stream_));   // ... and its copies must not land on top of this batch
    stage_pcm(pcm, n, B, h_pcm_, h_meta_, stream_, clips_, batch_max_frames_all_);
    batch_ = (int)B;
    run_issued_ = false;
    pcm_staged_over_ = false;
    d_pcm_off_ = d_meta_.as<long>();
    d_n_samples_ = reinterpret_cast<int*>(d_pcm_off_ + B);
    d_frame_off_ = d_n_samples_ + B;
}

// qasr_batch_stage: the NEXT batch's clips -> a second pinned buffer -> HBM on a copy stream, behind the current batch's log-mel (the only
// reader of the device PCM / meta buffers), while the current batch's encoder / prompt pass / decode run.
void Engine::batch_stage(const float* const* pcm, const size_t* n, size_t B) {
    if (!finalized_) throw NotLoaded("weights not finalized");
    require_asr("batch_stage");
    if (B == 0) throw std::invalid_argument("empty batch");
    if (batch_ > 0 && !run_issued_)
        throw std::invalid_argument("batch_stage: the current batch has not been started yet (call qasr_batch_run first: its log-mel still needs the PCM buffer)");
    if (!copy_stream_) {
        QASR_HIP(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
        QASR_HIP(hipEventCreateWithFlags(&ev_mel_done_, hipEventDisableTiming));
        QASR_HIP(hipEventCreateWithFlags(&ev_stage_done_, hipEventDisableTiming));
        h_pcm2_.alloc(h_pcm_.bytes);
        h_meta2_.alloc(h_meta_.bytes);
        // The batch already running was issued before this event existed, so its batch_run could not record it after the log-mel:
        // record it now, at the tail of everything that batch has queued (later than needed, never earlier).  Without this the wait
        // below is on a never-recorded event, i.e. no wait, and the copies could land in d_pcm_ / d_meta_ under a log-mel still queued.
        if (run_issued_) QASR_HIP(hipEventRecord(ev_mel_done_, stream_));
    }
    staged_valid_ = false;
    QASR_HIP(hipStreamSynchronize(copy_stream_));  // the previous staged batch's copies have left the second pinned buffer
    if (run_issued_) QASR_HIP(hipStreamWaitEvent(copy_stream_, ev_mel_done_, 0));
    stage_pcm(pcm, n, B, h_pcm2_, h_meta2_, copy_stream_, staged_clips_, staged_max_frames_all_);
    QASR_HIP(hipEventRecord(ev_stage_done_, copy_stream_));
    staged_B_ = (int)B;
    staged_valid_ = true;
    pcm_staged_over_ = true;                       // the device PCM buffer no longer holds the current batch
}
void Engine::run_mel() {
    MelBatch mb;
    mb.pcm = d_pcm_.as<float>();
    mb.pcm_off = d_pcm_off_;
    mb.n_samples = d_n_samples_;
    mb.frame_off = d_frame_off_;
    mb.B = batch_;
    mb.max_frames_all = batch_max_frames_all_;
    mb.raw = d_mel_raw_.as<float>();
    mb.gmax = d_gmax_.as<unsigned>();
    mb.out = d_mel_.as<float>();
    mb.out_stride = mel_stride_;
    mel_launch(mel_tables_, mb, stream_);
}

void Engine::mel_host(const float* pcm, size_t n, float* out) {
    const float* ptrs[1] = {pcm};
    size_t ns[1] = {n};
    upload_pcm(ptrs, ns, 1);
    run_mel();
    int T = clips_[0].frames;
    if (T > 0)
        QASR_HIP(hipMemcpy2DAsync(out, (size_t)T * sizeof(float), d_mel_.p, (size_t)mel_stride_ * sizeof(float),
                                  (size_t)T * sizeof(float), MEL_NMELS, hipMemcpyDeviceToHost, stream_));
    QASR_HIP(hipStreamSynchronize(stream_));
}

}  // namespace qasr

// engine.h -- host-side engine: owns the HIP device/stream, weights, workspaces and the batch plan.
#pragma once
#include "common.h"
#include "mel.h"
#include "enc_kernels.h"
#include "dec_kernels.h"
#include "dec_quant.h"
#include "qasr.h"
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace qasr {

// RAII device buffer
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        QASR_HIP(hipMalloc(&p, n));
        bytes = n;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Pinned host staging buffer
struct HostBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~HostBuf() { if (p) (void)hipHostFree(p); }
    void alloc(size_t n) {
        if (p) (void)hipHostFree(p);
        if (n == 0) n = 16;
        QASR_HIP(hipHostMalloc(&p, n, hipHostMallocDefault));
        bytes = n;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct Tensor {           // a named parameter resident in HBM
    DevBuf buf;
    std::vector<int64_t> shape;
    int dtype = QASR_DTYPE_BF16;
    size_t numel() const { size_t n = 1; for (auto d : shape) n *= (size_t)d; return n; }
};

// Per-clip geometry derived on the host (integer work of R2/R4/R7)
struct ClipPlan {
    long n_samples = 0;
    int frames_all = 0;   // incl. the dropped last frame
    int frames = 0;       // handed to the encoder
    int n_chunks = 0;
    int last_chunk_len = 0;
    int max_chunk_len = 0;      // conv input width for this clip's chunks (100, or <100 for a single short chunk)
    int n_tokens = 0;     // audio tokens
    int prompt_len = 0;
    std::vector<int> windows;   // attention window lengths
};

class Engine {
public:
    explicit Engine(const qasr_config& cfg);
    ~Engine();

    const qasr_config& config() const { return cfg_; }
    void bind_device() const { QASR_HIP(hipSetDevice(cfg_.device)); }
    hipStream_t stream() const { return stream_; }
    std::string last_error;

    // weights
    void set_tensor(const std::string& name, const void* host, int dtype, const int64_t* shape, int ndim);
    void load_directory(const std::string& dir);
    void finalize();
    bool loaded() const { return finalized_; }
    void unload();
    size_t memory_footprint() const;

    // stages (host in / host out; used by the oracle-diff entry points)
    void mel_host(const float* pcm, size_t n, float* out);
    void encode_host(const float* mel, int n_frames, float* out);
    int num_audio_tokens(int n_frames) const;
    void prefill_logits_host(const float* audio_embeds, int n_audio, const qasr_options* opt, float* logits);
    void decode_forced_host(const int32_t* tokens, int n, float* logits);

    // batch pipeline (qasr_batch_*): H2D + plan | mel + encoder + prefill + greedy decode | D2H
    void batch_begin(const float* const* pcm, const size_t* n, size_t B, const qasr_options* opt);
    void batch_stage(const float* const* pcm, const size_t* n, size_t B);
    void batch_begin_staged(const qasr_options* opt);
    void plan_batch(const qasr_options* opt, int max_tokens);
    void batch_run();
    void batch_rewind();
    void batch_sync();
    void batch_tokens(int32_t* tokens, int32_t* lens);
    void batch_timings(float ms[5], int32_t* n_steps);
    void gemm_probe(const uint16_t* A, const uint16_t* W, const float* bias, int M, int N, int K, int form, int reps, float* out,
                    float* avg_ms);
    void kernel_probe(int which, int reps, float* avg_ms, double* bytes_per_launch);
    int batch_size() const { return batch_; }
    void decode_structure(int* fused_qa, int* chain, int* launches_per_layer);
    int step_chain(int r0, int nr) const;           // the chain knob where the persistent launch applies to rows [r0, r0 + nr), else 0
    bool step_qa(int r0, int nr, int chain) const;  // q|k|v + attention as one launch for those rows
    // several engines run concurrently on this GPU: no launch may rely on its whole grid being resident (dec_chain.hip, dec_qa.hip)
    void set_shared_device(bool shared) { if (shared != shared_device_) { shared_device_ = shared; drop_graph(); } }
    void require_asr(const char* what) const {
        if (cfg_.classify_num > 0) throw std::runtime_error(std::string(what) + ": this engine is a forced aligner (no LM head / decode state)");
    }

    // forced aligner (N3): device forward + the host logic of csrc/aligner.cpp
    void align_forward(const float* const* pcm, const size_t* n, size_t B, const std::vector<std::vector<int32_t>>& slotted,
                       const std::vector<std::vector<int32_t>>& ts_pos, std::vector<std::vector<int32_t>>& raw, float* logits);
    struct AlignedWord { std::string text; float start, end; };
    struct SlottedText { std::vector<int32_t> ids, ts_pos; std::vector<std::string> words; };
    SlottedText prepare_alignment(const std::vector<std::pair<std::string, std::string>>& pairs) const;
    int align_words(const float* pcm, size_t n, const std::vector<std::pair<std::string, std::string>>& pairs, bool long_form,
                    const std::string* long_text = nullptr);
    void align_batch(const float* const* pcm, const size_t* n, size_t B,
                     const std::vector<std::vector<std::pair<std::string, std::string>>>& pairs);
    struct AlignResult { std::vector<AlignedWord> words; std::vector<qasr_aligned_word> view; std::vector<int32_t> raw; };
    std::vector<AlignResult> al_batch;         // owned result storage for qasr_align_batch
    std::vector<AlignedWord> al_words;         // owned result storage for qasr_align*
    std::vector<qasr_aligned_word> al_view;
    std::vector<int32_t> al_raw;

    // tokenizer (R9)
    void set_vocab(const int32_t* ids, const char* const* tokens, size_t n);
    void set_merges(const std::string& merges_txt);
    std::vector<int32_t> encode_text(const std::string& text) const;
    void load_vocab_files(const std::string& dir);
    std::string detokenize(const int32_t* tokens, int n, bool strip_asr_prefix) const;
    std::string result_text;              // owned result storage for qasr_transcribe / vtable
    std::vector<int32_t> result_tokens;

    static ClipPlan plan_clip(const qasr_config& cfg, long n_samples, int extra_prompt);

private:
    void upload_pcm(const float* const* pcm, const size_t* n, size_t B);
    void run_mel();
    const Tensor& tensor(const std::string& name) const;
    const bf16_t* wptr(const std::string& name, std::initializer_list<int64_t> shape) const;
    // f32 view of an audio-encoder vector (bias, LayerNorm gain / shift): the reference runs the encoder in f32 with these
    // tensors widened from the checkpoint dtype, so f16 / f32 checkpoints keep every bit and bf16 ones widen exactly
    const float* fvec(const std::string& name, int64_t n);
    void finalize_encoder();
    void alloc_encoder_workspace();
    void plan_encoder();          // chunk / token / window tables of the current batch -> HBM
    void run_encoder();
    void finalize_decoder();
    const bf16_t* packed_copy(const bf16_t* w, int N, int K);
    QuantRaw quant_raw(const std::string& stem, int N, int K) const;
    QuantImg quant_image(const QuantRaw& raw);
    void embed_rows(const int* d_ids, bf16_t* dst, int n, hipStream_t s);      // dst[i] = embedding row ids[i] (either table)
    void plan_prefill(const qasr_options* opt, const std::vector<int>& n_audio,
                      const std::vector<std::vector<int32_t>>* aligner_tails = nullptr);
    void run_prefill(bool want_logits);
    void decode_gemv(DecEpi epi, const DecGemvArgs& a, const QuantImg& qi, const bf16_t* norm_w, bf16_t* h, hipStream_t s);
    void run_decode_step(bool want_logits, bool greedy, int r0, int nr, hipStream_t s, bool with_head);
    void issue_decode_step(int split);
    void sample_and_finalize(int advance_ctx);
    int decode_group_rows() const;
    GreedyState greedy_rows(int r0) const;
    RopeRows rope_rows(int r0) const;
    void run_lm_head(bool want_logits, int r0, int nr, hipStream_t s);
    void reset_greedy_state(int max_tokens, bool ignore_eos);
    void decode_loop();

    qasr_config cfg_;
    hipStream_t stream_ = nullptr;
    MelTables mel_tables_;
    std::map<std::string, Tensor> tensors_;
    bool finalized_ = false;

    // batch state
    std::vector<ClipPlan> clips_;
    long max_samples_ = 0;       // per clip capacity
    int max_frames_all_ = 0;
    HostBuf h_pcm_, h_meta_;
    // batch staged ahead (qasr_batch_stage / qasr_batch_begin_staged): second pinned buffers, copy stream, hand-over events
    HostBuf h_pcm2_, h_meta2_;
    hipStream_t copy_stream_ = nullptr;
    hipEvent_t ev_mel_done_ = nullptr, ev_stage_done_ = nullptr;
    std::vector<ClipPlan> staged_clips_;
    int staged_B_ = 0, staged_max_frames_all_ = 0;
    bool staged_valid_ = false, run_issued_ = false, pcm_staged_over_ = false;
    void stage_pcm(const float* const* pcm, const size_t* n, size_t B, HostBuf& hp, HostBuf& hm, hipStream_t cs, std::vector<ClipPlan>& clips,
                   int& max_frames_all);
    DevBuf d_pcm_, d_meta_, d_mel_raw_, d_gmax_, d_mel_;
    int mel_stride_ = 0;
    // device views into d_meta_ (ints/longs), rebuilt per batch
    long* d_pcm_off_ = nullptr;
    int* d_n_samples_ = nullptr;
    int* d_frame_off_ = nullptr;
    int batch_ = 0;
    int batch_max_frames_all_ = 0;

    // ---- audio encoder ---------------------------------------------------------------------
    struct EncLayerW {
        const bf16_t *wqkv, *wo, *w1, *w2;                       // MFMA operands: bf16
        const float *ln1_g, *ln1_b, *bqkv, *bo, *ln2_g, *ln2_b, *b1, *b2;   // biases / LayerNorm parameters: f32 (see fvec)
    };
    struct EncW {
        const bf16_t *c1w, *c2w, *c3w, *conv_out, *p1w, *p2w;
        const float *c1b, *c2b, *c3b, *lnp_g, *lnp_b, *p1b, *p2b;
        std::vector<EncLayerW> layers;
    } encw_;
    std::vector<std::unique_ptr<DevBuf>> fused_;   // concatenated / permuted copies built by finalize
    int max_win_ = 0;                              // longest attention window of the planned batch
    DevBuf d_pe_;                                  // [W3][d_model] f32 sinusoid table
    int H1_ = 0, W1_ = 0, H2_ = 0, W2_ = 0, H3_ = 0, W3_ = 0;
    int max_chunks_ = 0, max_tokens_ = 0;          // capacity (whole batch)
    DevBuf d_c1_, d_c2_, d_c3_, d_encx_, d_ench_, d_encqkv_, d_enca_, d_encmid_, d_audio_;
    HostBuf h_encmeta_;
    DevBuf d_encmeta_;
    ChunkMeta* d_chunks_ = nullptr;
    long* d_tok_rowoff_ = nullptr;
    int* d_tok_t_ = nullptr;
    int* d_cu_win_ = nullptr;
    int n_img_ = 0, n_tok_ = 0, n_win_ = 0;
    std::vector<int> clip_tok_off_;                // first packed audio token of each clip

    // ---- text decoder ------------------------------------------------------------------------
    struct DecLayerW {
        const bf16_t *ln1, *wqkv, *qn, *kn, *wo, *ln2, *wgu, *wdown;
        const bf16_t *wqkv_p, *wo_p, *wgu_p, *wdown_p;       // fragment-major copies for the decode step
        QuantImg qkv_q, o_q, gu_q, down_q;                   // quantised checkpoints: packed decode-step images instead
        QuantRaw rq{}, rk{}, rv{}, ro{}, rg{}, ru{}, rd{};   // ... and the uploaded triplets, the prompt pass's source (prompt_weights)
    };
    struct PromptW { const bf16_t *wqkv, *wo, *wgu, *wdown; };
    // bf16 matrices of layer l for the prompt-pass GEMMs: the resident fused tensors of a float checkpoint; for an MLX-quantised one
    // bf16(scale * q + bias) written into ONE reusable layer-sized scratch on stream s right before use (the reference's many-row
    // kernel multiplies by exactly that value) -- no bf16 expansion of the decoder stays resident
    PromptW prompt_weights(int l, hipStream_t s);
    DevBuf d_wscratch_;
    struct DecW {
        const bf16_t *embed, *norm, *embed_p;
        const bf16_t *cls_w = nullptr, *cls_b = nullptr;       // aligner: Linear(hidden, classify_num) `lm_head.{weight,bias}`
        bool quant = false;                                    // MLX 4 / 8 bit checkpoint (QuantizedTextModel): packed in HBM
        QuantRaw embed_raw{};                                  // quantised tied embedding as uploaded (row gather)
        QuantImg embed_q{};                                    // its LM-head image
        std::vector<DecLayerW> layers;
    } decw_;
    int max_prompt_ = 0, max_ctx_ = 0, max_pos_ = 0, vt_stride_ = 0;
    DevBuf d_rope_cos_, d_rope_sin_, d_rope_rows_;    // tables [max_ctx][hd/2]; per-row copies for the next step
    std::vector<std::unique_ptr<DevBuf>> kcache_, vfcache_;     // per layer: keys row-major, values fragment-major
    unsigned long long* stamp_buf_ = nullptr;                   // diagnostic in-situ stamps (kernel_probe, make DIAG=1)
    int stamp_layer_ = -1;
    DevBuf d_al_rows_, d_al_x_, d_al_logits_, d_al_idx_;         // aligner head workspace (grown on demand)
    DevBuf d_vt_;
    DevBuf d_px_, d_ph_, d_pqkv_, d_pqr_, d_pattn_, d_pact_;   // prefill (packed prompt positions)
    DevBuf d_dx_, d_dh_, d_dqkv_, d_dattn_, d_dact_, d_logits_, d_part_val_, d_part_idx_;   // decode rows
    bool shared_device_ = false;
    unsigned long long* qa_dbg_ = nullptr;
    unsigned long long* chain_dbg_ = nullptr;                  // diagnostic stamps of the middle layer's chain launch (kernel_probe 6)
    DevBuf d_qa_part_;                                         // ... and the per-wave attention partials of a unit spread over several workgroups (QA_PART_BYTES)
    DevBuf d_qa_gran_;                                         // the fused q|k|v + attention launch's hand-off granules (dec_chain.h QA_GRAN_BYTES)
    DevBuf d_chain_ctr_;                                       // arrival counters of the persistent layer launch (dec_chain.h)
    HostBuf h_pmeta_;
    DevBuf d_pmeta_;
    int *d_p_ids_ = nullptr, *d_p_audio_src_ = nullptr, *d_p_slot_ = nullptr, *d_p_pos_ = nullptr;
    int *d_p_cu_ = nullptr, *d_p_slotclip_ = nullptr, *d_p_last_ = nullptr;
    int n_pos_ = 0, max_len_ = 0;
    std::vector<int> prompt_len_;
    DevBuf d_gstate_;                                          // tokens | lens | finished | ctx_len | n_active
    GreedyState gstate_{};
    int cur_max_tokens_ = 448;
    bool cur_ignore_eos_ = false;
    int n_parts_ = 0;
    int steps_done_ = 0;
    // decode-step graph, keyed by (B, max_tokens, ignore_eos)
    hipGraphExec_t graph_exec_ = nullptr;
    hipGraphExec_t graph_exec_n_ = nullptr;                      // the same step captured tuning().graph_steps times in a row
    int graph_n_ = 0;
    long graph_key_ = -1;
    unsigned graph_epoch_ = 0;                                 // tuning().epoch the graph was captured under
    int* d_err_flag_ = nullptr;                                // device word: set when a greedy step sees a non-finite best logit
    int forced_ctx_ = 0;                                       // host copy of slot 0's context length (decode_forced capacity)
    void drop_graph();
    void require_batch(const char* what) const;
    hipEvent_t ev_[6] = {};
    hipStream_t side_[3] = {};                                 // parallel decode row groups
    hipEvent_t fork_ev_ = nullptr, join_ev_[3] = {};
    std::vector<int> h_ctx0_;
    HostBuf h_ginit_;                                          // pinned: ctx_len init [max_batch] | n_active

    // ---- tokenizer -----------------------------------------------------------------------------
    std::unordered_map<int32_t, std::string> id_to_token_;
    std::unordered_map<std::string, int32_t> token_to_id_;
    std::unordered_map<std::string, int> merge_rank_;

    // ---- slow decoding path (Qwen3DecodingOptions) ---------------------------------------------
    bool slow_path_ = false;
    float opt_rep_penalty_ = 1.0f, opt_temperature_ = 0.0f;
    int opt_ngram_ = 0;
    uint64_t opt_seed_ = 0;
    void decode_loop_slow();
};

// host logic of the forced aligner (aligner.cpp)
bool aligner_needs_nl_tokenizer(const std::string& language);
std::vector<std::pair<std::string, std::string>> aligner_split_word_pairs(const std::string& text);     // (surface, cleaned)
std::vector<int32_t> aligner_lis_positions(const int32_t* values, size_t n);
std::vector<int32_t> aligner_enforce_monotonicity(const int32_t* raw, size_t n);
int aligner_find_trailing_plateau(const float* start_times, size_t n, float tolerance, int min_size);

}  // namespace qasr

// the C ABI's engine handle (api.cpp, api_dp.cpp)
struct qasr_engine {
    std::unique_ptr<qasr::Engine> impl;
};

// ctc_engine.h -- Omnilingual ASR (wav2vec2 encoder + CTC head) engine: BASELINE configs[3], SURVEY.md section 8f N4.
// Mirrors OmnilingualASRMLXModel (Sources/OmnilingualASR/MLX/OmnilingualMLXModel.swift): raw 16 kHz samples ->
// utterance layer-norm -> 7-layer conv feature extractor -> LayerNorm -> Linear -> grouped-conv positional encoder ->
// N pre-norm transformer layers -> LayerNorm -> CTC head -> per-frame argmax -> duplicate collapse -> SentencePiece text.
// One engine = one GPU + one stream; clips of a batch are independent (packed frames, per-clip attention).
#pragma once
#include "engine.h"
#include "ctc_kernels.h"
#include <string>
#include <vector>

namespace qasr {

class CtcEngine {
public:
    explicit CtcEngine(const qasr_ctc_config& cfg);
    ~CtcEngine();
    const qasr_ctc_config& config() const { return cfg_; }
    std::string last_error;

    void set_tensor(const std::string& name, const void* host, int dtype, const int64_t* shape, int ndim);
    void load_directory(const std::string& dir);
    void finalize();
    bool loaded() const { return finalized_; }
    void unload();
    size_t memory_footprint() const;

    static int num_frames(long n_samples);               // Wav2Vec2FeatureExtractor.outputLength
    int max_frames() const { return num_frames(max_samples_); }

    // raw argmax ids per frame of every clip (packed by clip), and optionally the logits [total frames][vocab] (host)
    void forward(const float* const* pcm, const size_t* n, size_t B, std::vector<std::vector<int32_t>>& frame_ids, float* logits);
    void timings(float ms[4]);

    // vocabulary (OmnilingualVocabulary)
    void set_pieces(const char* const* texts, const int32_t* types, size_t n);
    void load_sentencepiece(const std::string& path);
    bool has_pieces() const { return !pieces_.empty(); }
    std::string detokenize(const int32_t* ids, int n) const;
    std::string result_text;

private:
    struct Lin { const bf16_t* w = nullptr; const float* b = nullptr; };      // bf16 [N][K] weight + f32 bias
    struct Layer { const float *ln1_g, *ln1_b, *ln2_g, *ln2_b; Lin qkv, o, f1, f2; };
    const Tensor& tensor(const std::string& name) const;
    const float* f32_param(const std::string& name, std::initializer_list<int64_t> shape);
    void linear_weight(const std::string& stem, int N, int K, bf16_t* dst);    // float or MLX triplet -> bf16 rows
    void* new_buf(size_t bytes);

    qasr_ctc_config cfg_;
    hipStream_t stream_ = nullptr;
    std::map<std::string, Tensor> tensors_;
    std::vector<std::unique_ptr<DevBuf>> built_;          // everything finalize derives from the tensors
    bool finalized_ = false;
    long max_samples_ = 0;
    // weights as the kernels take them
    struct Conv { const void* w; const float *b, *ln_g, *ln_b; } conv_[7]{};
    const float *post_g_ = nullptr, *post_b_ = nullptr, *pos_b_ = nullptr, *final_g_ = nullptr, *final_b_ = nullptr;
    Lin proj_, head_;
    const bf16_t* pos_w_ = nullptr;                       // [D][KP][cpg] bf16, weight norm fused
    std::vector<Layer> layers_;
    // workspaces
    HostBuf h_pcm_, h_meta_;
    DevBuf d_pcm_, d_meta_, d_stats_, d_act_[2], d_convf_, d_rows_, d_x_, d_y_, d_h_, d_qkv_, d_att_, d_mid_, d_logits_, d_ids_, d_info_;
    int cap_frames_ = 0;                                  // packed transformer frames
    long cap_conv_rows_ = 0;                              // packed rows of the widest conv layer
    hipEvent_t ev_[4] = {};
    // tokenizer
    std::vector<std::pair<std::string, int>> pieces_;
};

int ctc_greedy_decode(const float* logits, int T, int V, int valid_frames, int32_t* out);
void layer_normalize_host(const float* x, size_t n, float eps, float* out);

}  // namespace qasr

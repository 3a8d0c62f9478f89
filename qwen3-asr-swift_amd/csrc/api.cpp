// api.cpp -- extern "C" boundary (include/qasr.h).  Exceptions never cross it.
#include "engine.h"
#include "ctc_engine.h"
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using qasr::Engine;

static thread_local std::string g_create_error;

static int fail(qasr_engine* e, int code, const std::string& msg) {
    // a reported HIP failure must not stay behind as the runtime's sticky "last error" (a later launch check would blame itself for it)
    if (code == QASR_ERR_HIP) (void)hipGetLastError();
    if (e && e->impl) e->impl->last_error = msg; else g_create_error = msg;
    return code;
}

// every guarded entry first makes the engine's device current for the calling thread (the HIP current device is per thread: an engine
// per GPU may be driven from any thread, e.g. the worker threads of qasr_dp_*)
#define QASR_GUARD(e, body)                                                          \
    try { (e)->impl->bind_device(); body; return QASR_OK; }                          \
    catch (const qasr::HipError& ex) { return fail(e, QASR_ERR_HIP, ex.what()); }    \
    catch (const qasr::NotLoaded& ex) { return fail(e, QASR_ERR_NOT_LOADED, ex.what()); } \
    catch (const std::invalid_argument& ex) { return fail(e, QASR_ERR_INVALID, ex.what()); } \
    catch (const std::length_error& ex) { return fail(e, QASR_ERR_CAPACITY, ex.what()); }    \
    catch (const std::exception& ex) { return fail(e, QASR_ERR_INVALID, ex.what()); }

static bool contains(const std::string& s, const char* sub) { return s.find(sub) != std::string::npos; }

extern "C" {

int qasr_default_config(const char* preset, qasr_config* c) {
    if (!c) return QASR_ERR_INVALID;
    std::string p = preset ? preset : "0.6B";
    std::memset(c, 0, sizeof(*c));
    // Qwen3AudioEncoderConfig.small (AudioEncoder.swift:28-45), TextDecoderConfig.small (Configuration.swift:68-79)
    c->enc_d_model = 896; c->enc_heads = 14; c->enc_ffn = 3584; c->enc_layers = 18; c->n_mels = 128;
    c->enc_out_dim = 1024; c->conv_channels = 480; c->n_window = 50; c->n_window_infer = 800; c->ln_eps = 1e-5f;
    c->vocab = 151936; c->hidden = 1024; c->dec_layers = 28; c->heads = 16; c->kv_heads = 8; c->head_dim = 128;
    c->inter = 3072; c->rms_eps = 1e-6f; c->rope_theta = 1000000.0f; c->group_size = 64; c->bits = 4;
    c->tok_im_start = 151644; c->tok_im_end = 151645; c->tok_audio_start = 151669; c->tok_audio_end = 151670;
    c->tok_audio_pad = 151676; c->tok_asr_text = 151704; c->tok_newline = 198; c->tok_system = 8948;
    c->tok_user = 872; c->tok_assistant = 77091;
    c->fft_scale = 2.0f;
    c->device = 0; c->max_batch = 32; c->max_audio_seconds = 30; c->max_new_tokens = 448; c->max_prompt_extra = 64;
    c->classify_num = 0; c->tok_timestamp = 151705; c->timestamp_segment_time = 0.08f;
    if (p == "tiny" || p == "tiny-aligner") {   // test geometry (oracle/config.py AUDIO_TINY / TEXT_TINY / TOKENS_TINY)
        c->enc_d_model = 64; c->enc_heads = 2; c->enc_ffn = 128; c->enc_layers = 2; c->enc_out_dim = 64;
        c->conv_channels = 32; c->n_window_infer = 200;
        c->vocab = 512; c->hidden = 64; c->dec_layers = 2; c->heads = 4; c->kv_heads = 2; c->head_dim = 32; c->inter = 128;
        c->tok_im_start = 500; c->tok_im_end = 501; c->tok_audio_start = 502; c->tok_audio_end = 503;
        c->tok_audio_pad = 504; c->tok_asr_text = 505; c->tok_newline = 198; c->tok_system = 300;
        c->tok_user = 301; c->tok_assistant = 302;
        c->bits = 16; c->max_batch = 8;
        if (p == "tiny-aligner") { c->classify_num = 40; c->tok_timestamp = 506; c->max_prompt_extra = 256; c->max_new_tokens = 1; }
        return QASR_OK;
    }
    std::string lower_p = p;
    for (auto& ch : lower_p) ch = (char)tolower(ch);
    if (contains(lower_p, "aligner")) {
        // Qwen3ForcedAligner (ForcedAligner.swift:63-83): encoder = Qwen3AudioEncoderConfig.forcedAligner
        // (AudioEncoder.swift:71-88: the large encoder projecting to 1024), text decoder = .small, 5000 classes
        c->enc_d_model = 1024; c->enc_heads = 16; c->enc_ffn = 4096; c->enc_layers = 24; c->enc_out_dim = 1024;
        c->classify_num = 5000;
        c->bits = contains(lower_p, "bf16") || contains(lower_p, "float") ? 16 : contains(lower_p, "8bit") ? 8 : 4;   // ForcedAlignerVariant.detect :17-26
        c->max_batch = 1; c->max_audio_seconds = 1200; c->max_new_tokens = 1; c->max_prompt_extra = 4096;
        return QASR_OK;
    }
    // ASRModelSize.detect / detectBits (Qwen3ASR.swift:581-601)
    bool large = contains(p, "1.7B") || contains(p, "1.7b");
    std::string lower = p;
    for (auto& ch : lower) ch = (char)tolower(ch);
    int bits = (contains(lower, "8bit") || contains(lower, "8-bit")) ? 8
             : (contains(lower, "4bit") || contains(lower, "4-bit")) ? 4 : (large ? 8 : 4);
    c->bits = bits;
    if (large) {         // .large presets (AudioEncoder.swift:51-68, Configuration.swift:89-100)
        c->enc_d_model = 1024; c->enc_heads = 16; c->enc_ffn = 4096; c->enc_layers = 24; c->enc_out_dim = 2048;
        c->hidden = 2048; c->inter = 6144;
    }
    return QASR_OK;
}

int qasr_create(const char* model_dir, const qasr_config* cfg, qasr_engine** out) {
    if (!cfg || !out) return QASR_ERR_INVALID;
    *out = nullptr;
    qasr_engine* e = new qasr_engine();
    try {
        e->impl.reset(new Engine(*cfg));
        if (model_dir) { e->impl->load_directory(model_dir); e->impl->finalize(); }
    } catch (const qasr::HipError& ex) { g_create_error = ex.what(); delete e; (void)hipGetLastError(); return QASR_ERR_HIP; }
    catch (const std::exception& ex) { g_create_error = ex.what(); delete e; return model_dir ? QASR_ERR_IO : QASR_ERR_INVALID; }
    *out = e;
    return QASR_OK;
}

int qasr_set_tensor(qasr_engine* e, const char* name, const void* host, int dtype, const int64_t* shape, int ndim) {
    if (!e || !name || !host || !shape || ndim <= 0) return QASR_ERR_INVALID;
    QASR_GUARD(e, e->impl->set_tensor(name, host, dtype, shape, ndim));
}
int qasr_finalize(qasr_engine* e) { if (!e) return QASR_ERR_INVALID; QASR_GUARD(e, e->impl->finalize()); }
int qasr_is_loaded(const qasr_engine* e) { return e && e->impl->loaded(); }
int qasr_unload(qasr_engine* e) { if (!e) return QASR_ERR_INVALID; QASR_GUARD(e, e->impl->unload()); }
size_t qasr_memory_footprint(const qasr_engine* e) { return e ? e->impl->memory_footprint() : 0; }
void qasr_destroy(qasr_engine* e) { delete e; }
const char* qasr_last_error(const qasr_engine* e) { return e ? e->impl->last_error.c_str() : g_create_error.c_str(); }
int qasr_input_sample_rate(const qasr_engine*) { return 16000; }

int qasr_num_mel_frames(size_t n) { return qasr::mel_num_frames((long)n); }

int qasr_mel(qasr_engine* e, const float* pcm, size_t n, float* out) {
    if (!e || !pcm || !out) return QASR_ERR_INVALID;
    if (n == 0) return fail(e, QASR_ERR_EMPTY_AUDIO, "empty clip");
    QASR_GUARD(e, e->impl->mel_host(pcm, n, out));
}

int qasr_num_audio_tokens(const qasr_engine* e, int n_frames) { return e ? e->impl->num_audio_tokens(n_frames) : -1; }

int qasr_encode(qasr_engine* e, const float* mel, int n_frames, float* out) {
    if (!e || !mel || !out) return QASR_ERR_INVALID;
    if (!e->impl->loaded()) return fail(e, QASR_ERR_NOT_LOADED, "weights not finalized");
    QASR_GUARD(e, e->impl->encode_host(mel, n_frames, out));
}

int qasr_set_vocab(qasr_engine* e, const int32_t* ids, const char* const* tokens, size_t n) {
    if (!e || !ids || !tokens) return QASR_ERR_INVALID;
    QASR_GUARD(e, e->impl->set_vocab(ids, tokens, n));
}

int qasr_set_merges(qasr_engine* e, const char* merges_txt) {
    if (!e || !merges_txt) return QASR_ERR_INVALID;
    QASR_GUARD(e, e->impl->set_merges(merges_txt));
}

int qasr_encode_text(qasr_engine* e, const char* utf8, int32_t* ids, int32_t cap) {
    if (!e || !utf8 || !ids || cap < 0) return -1;
    try {
        std::vector<int32_t> v = e->impl->encode_text(utf8);
        if ((int32_t)v.size() > cap) { fail(e, QASR_ERR_CAPACITY, "encode_text: buffer too small"); return -1; }
        std::memcpy(ids, v.data(), v.size() * sizeof(int32_t));
        return (int32_t)v.size();
    } catch (const std::exception& ex) { fail(e, QASR_ERR_INVALID, ex.what()); return -1; }
}

int qasr_detokenize(qasr_engine* e, const int32_t* tokens, int32_t n, char* buf, size_t cap) {
    if (!e || !tokens || !buf || cap == 0 || n < 0) return -1;
    try {
        std::string t = e->impl->detokenize(tokens, n, true);
        if (t.size() + 1 > cap) { fail(e, QASR_ERR_CAPACITY, "detokenize: buffer too small"); return -1; }
        std::memcpy(buf, t.c_str(), t.size() + 1);
        return (int)t.size();
    } catch (const std::exception& ex) { fail(e, QASR_ERR_INVALID, ex.what()); return -1; }
}

int qasr_batch_begin(qasr_engine* e, const float* const* pcm, const size_t* n, size_t B, const qasr_options* opt) {
    if (!e || !pcm || !n) return QASR_ERR_INVALID;
    if (!e->impl->loaded()) return fail(e, QASR_ERR_NOT_LOADED, "weights not finalized");
    for (size_t b = 0; b < B; ++b) if (n[b] == 0 || !pcm[b]) return fail(e, QASR_ERR_EMPTY_AUDIO, "empty clip in batch");
    QASR_GUARD(e, e->impl->batch_begin(pcm, n, B, opt));
}
int qasr_batch_stage(qasr_engine* e, const float* const* pcm, const size_t* n, size_t B) {
    if (!e || !pcm || !n) return QASR_ERR_INVALID;
    QASR_GUARD(e, e->impl->batch_stage(pcm, n, B));
}
int qasr_batch_begin_staged(qasr_engine* e, const qasr_options* opt) {
    if (!e) return QASR_ERR_INVALID;
    QASR_GUARD(e, e->impl->batch_begin_staged(opt));
}
int qasr_batch_run(qasr_engine* e) { if (!e) return QASR_ERR_INVALID; QASR_GUARD(e, e->impl->batch_run()); }
int qasr_batch_rewind(qasr_engine* e) { if (!e) return QASR_ERR_INVALID; QASR_GUARD(e, e->impl->batch_rewind()); }
int qasr_batch_sync(qasr_engine* e) { if (!e) return QASR_ERR_INVALID; QASR_GUARD(e, e->impl->batch_sync()); }
int qasr_batch_tokens(qasr_engine* e, int32_t* tokens, int32_t* lens) {
    if (!e || !tokens || !lens) return QASR_ERR_INVALID;
    if (!e->impl->loaded()) return fail(e, QASR_ERR_NOT_LOADED, "weights not finalized");
    QASR_GUARD(e, e->impl->batch_tokens(tokens, lens));
}
int qasr_batch_timings(qasr_engine* e, float ms[5], int32_t* n_steps) {
    if (!e || !ms) return QASR_ERR_INVALID;
    QASR_GUARD(e, e->impl->batch_timings(ms, n_steps));
}
int qasr_kernel_probe(qasr_engine* e, int which, int reps, float* avg_ms, double* bytes_per_launch) {
    if (!e || !avg_ms || !bytes_per_launch || reps <= 0 || which < 0 || which > 7) return QASR_ERR_INVALID;
    QASR_GUARD(e, e->impl->kernel_probe(which, reps, avg_ms, bytes_per_launch));
}

int qasr_gemm_probe(qasr_engine* e, const uint16_t* A, const uint16_t* W, const float* bias, int M, int N, int K, int form,
                    int reps, float* out, float* avg_ms) {
    if (!e || !A || !W || !out || M <= 0 || N <= 0 || K <= 0 || K % 8 || N % 4 || form < -1 || form > 2 || reps <= 0)
        return QASR_ERR_INVALID;
    QASR_GUARD(e, e->impl->gemm_probe(A, W, bias, M, N, K, form, reps, out, avg_ms));
}

int qasr_transcribe_batch(qasr_engine* e, const float* const* pcm, const size_t* n, size_t B, int sample_rate,
                          const qasr_options* opt, int32_t* tokens, int32_t* lens) {
    if (!e || !tokens || !lens) return QASR_ERR_INVALID;
    // AudioPreprocessing.swift:327-329 resamples with AVAudioConverter (closed source); inputs must be 16 kHz here
    if (sample_rate != 16000) return fail(e, QASR_ERR_INVALID, "only 16 kHz input is supported (resampler is out of scope)");
    int rc = qasr_batch_begin(e, pcm, n, B, opt);
    if (rc) return rc;
    if ((rc = qasr_batch_run(e))) return rc;
    return qasr_batch_tokens(e, tokens, lens);
}

int qasr_transcribe(qasr_engine* e, const float* pcm, size_t n, int sample_rate, const qasr_options* opt, qasr_result* out) {
    if (!e || !out) return QASR_ERR_INVALID;
    const int stride = e->impl->config().max_new_tokens + 1;
    std::vector<int32_t> toks((size_t)stride);
    int32_t len = 0;
    const float* ptrs[1] = {pcm};
    size_t ns[1] = {n};
    int rc = qasr_transcribe_batch(e, ptrs, ns, 1, sample_rate, opt, toks.data(), &len);
    if (rc) return rc;
    try {
        e->impl->result_tokens.assign(toks.begin(), toks.begin() + len);
        e->impl->result_text = e->impl->detokenize(toks.data(), len, true);
    } catch (const std::exception& ex) { return fail(e, QASR_ERR_INVALID, ex.what()); }
    out->text = e->impl->result_text.c_str();
    out->tokens = e->impl->result_tokens.data();
    out->n_tokens = len;
    return QASR_OK;
}

// ---- forced aligner ---------------------------------------------------------------------------------
static char* dup_joined(const std::vector<std::string>& v) {
    size_t n = 1;
    for (auto& s : v) n += s.size() + 1;
    char* out = (char*)malloc(n);
    if (!out) return nullptr;
    char* q = out;
    for (size_t i = 0; i < v.size(); ++i) {
        if (i) *q++ = '\n';
        std::memcpy(q, v[i].data(), v[i].size());
        q += v[i].size();
    }
    *q = 0;
    return out;
}

int qasr_split_words(const char* text, const char* language, char** surfaces, char** cleaned) {
    if (!text) return -QASR_ERR_INVALID;
    if (surfaces) *surfaces = nullptr;
    if (cleaned) *cleaned = nullptr;
    try {
        if (qasr::aligner_needs_nl_tokenizer(language ? language : "English")) return -QASR_ERR_UNSUPPORTED;
        auto pairs = qasr::aligner_split_word_pairs(text);
        std::vector<std::string> a, b;
        for (auto& p : pairs) { a.push_back(p.first); b.push_back(p.second); }
        if (surfaces && !(*surfaces = dup_joined(a))) return -QASR_ERR_INVALID;
        if (cleaned && !(*cleaned = dup_joined(b))) return -QASR_ERR_INVALID;
        return (int)pairs.size();
    } catch (const std::exception&) { return -QASR_ERR_INVALID; }
}

int qasr_lis_positions(const int32_t* values, size_t n, int32_t* positions) {
    if ((!values && n) || !positions) return -QASR_ERR_INVALID;
    try {
        auto p = qasr::aligner_lis_positions(values, n);
        std::memcpy(positions, p.data(), p.size() * sizeof(int32_t));
        return (int)p.size();
    } catch (const std::exception&) { return -QASR_ERR_INVALID; }
}

int qasr_enforce_monotonicity(const int32_t* raw, size_t n, int32_t* out) {
    if ((!raw || !out) && n) return QASR_ERR_INVALID;
    try {
        auto v = qasr::aligner_enforce_monotonicity(raw, n);
        std::memcpy(out, v.data(), v.size() * sizeof(int32_t));
        return QASR_OK;
    } catch (const std::exception&) { return QASR_ERR_INVALID; }
}

int qasr_find_trailing_plateau(const float* start_times, size_t n, float tolerance, int32_t min_size) {
    if (!start_times && n) return -QASR_ERR_INVALID;
    return qasr::aligner_find_trailing_plateau(start_times, n, tolerance, min_size);
}

static int split_for(qasr_engine* e, const char* text, const char* language, std::vector<std::pair<std::string, std::string>>& pairs) {
    if (qasr::aligner_needs_nl_tokenizer(language ? language : "English"))
        return fail(e, QASR_ERR_UNSUPPORTED, "the reference splits this language with Apple's NLTokenizer: pass words to qasr_align_words");
    pairs = qasr::aligner_split_word_pairs(text);
    return QASR_OK;
}

int qasr_align_prepare(qasr_engine* e, const char* text, const char* language, int32_t* ids, int32_t ids_cap,
                       int32_t* ts_positions, int32_t ts_cap, int32_t* n_ts, int32_t* n_words) {
    if (!e || !text || !ids || !ts_positions) return -1;
    try {
        std::vector<std::pair<std::string, std::string>> pairs;
        if (split_for(e, text, language, pairs)) return -1;
        auto st = e->impl->prepare_alignment(pairs);
        if ((int32_t)st.ids.size() > ids_cap || (int32_t)st.ts_pos.size() > ts_cap) { fail(e, QASR_ERR_CAPACITY, "align_prepare: buffer too small"); return -1; }
        std::memcpy(ids, st.ids.data(), st.ids.size() * sizeof(int32_t));
        std::memcpy(ts_positions, st.ts_pos.data(), st.ts_pos.size() * sizeof(int32_t));
        if (n_ts) *n_ts = (int32_t)st.ts_pos.size();
        if (n_words) *n_words = (int32_t)st.words.size();
        return (int)st.ids.size();
    } catch (const std::exception& ex) { fail(e, QASR_ERR_INVALID, ex.what()); return -1; }
}

int qasr_align_raw(qasr_engine* e, const float* pcm, size_t n, const int32_t* slotted_ids, int32_t n_ids,
                   const int32_t* ts_positions, int32_t n_ts, int32_t* raw_indices, float* logits) {
    if (!e || !pcm || !slotted_ids || !ts_positions || !raw_indices || n_ids < 0 || n_ts < 0) return QASR_ERR_INVALID;
    if (n == 0) return fail(e, QASR_ERR_EMPTY_AUDIO, "empty clip");
    if (!e->impl->loaded()) return fail(e, QASR_ERR_NOT_LOADED, "weights not finalized");
    QASR_GUARD(e, {
        std::vector<std::vector<int32_t>> raw;
        e->impl->bind_device();
        e->impl->align_forward(&pcm, &n, 1, {std::vector<int32_t>(slotted_ids, slotted_ids + n_ids)},
                               {std::vector<int32_t>(ts_positions, ts_positions + n_ts)}, raw, logits);
        if (!raw.empty()) std::memcpy(raw_indices, raw[0].data(), raw[0].size() * sizeof(int32_t));
    });
}

static int align_common(qasr_engine* e, const float* pcm, size_t n, int sample_rate,
                        const std::vector<std::pair<std::string, std::string>>& pairs, bool long_form, qasr_alignment* out,
                        const char* long_text = nullptr) {
    if (sample_rate != 16000) return fail(e, QASR_ERR_INVALID, "input must be 16 kHz mono (no resampler: AVAudioConverter is not reproducible)");
    if (n == 0) return fail(e, QASR_ERR_EMPTY_AUDIO, "empty clip");
    if (!e->impl->loaded()) return fail(e, QASR_ERR_NOT_LOADED, "weights not finalized");
    QASR_GUARD(e, {
        const std::string text = long_text ? long_text : "";
        e->impl->bind_device();
        const int passes = e->impl->align_words(pcm, n, pairs, long_form, long_text ? &text : nullptr);
        out->words = e->impl->al_view.data();
        out->n_words = e->impl->al_view.size();
        out->raw_indices = e->impl->al_raw.data();
        out->n_indices = e->impl->al_raw.size();
        out->passes = passes;
    });
}

int qasr_align(qasr_engine* e, const float* pcm, size_t n, int sample_rate, const char* text, const char* language,
               qasr_alignment* out) {
    if (!e || !pcm || !text || !out) return QASR_ERR_INVALID;
    std::vector<std::pair<std::string, std::string>> pairs;
    try { if (int rc = split_for(e, text, language, pairs)) return rc; }
    catch (const std::exception& ex) { return fail(e, QASR_ERR_INVALID, ex.what()); }
    return align_common(e, pcm, n, sample_rate, pairs, false, out);
}

int qasr_align_long(qasr_engine* e, const float* pcm, size_t n, int sample_rate, const char* text, const char* language,
                    qasr_alignment* out) {
    if (!e || !pcm || !text || !out) return QASR_ERR_INVALID;
    std::vector<std::pair<std::string, std::string>> pairs;
    try { if (int rc = split_for(e, text, language, pairs)) return rc; }
    catch (const std::exception& ex) { return fail(e, QASR_ERR_INVALID, ex.what()); }
    return align_common(e, pcm, n, sample_rate, pairs, true, out, text);
}

int qasr_align_batch(qasr_engine* e, const float* const* pcm, const size_t* n, size_t B, int sample_rate,
                     const char* const* texts, const char* language, qasr_alignment* out) {
    if (!e || !pcm || !n || !texts || !out || B == 0) return QASR_ERR_INVALID;
    if (sample_rate != 16000) return fail(e, QASR_ERR_INVALID, "input must be 16 kHz mono (no resampler: AVAudioConverter is not reproducible)");
    if (!e->impl->loaded()) return fail(e, QASR_ERR_NOT_LOADED, "weights not finalized");
    std::vector<std::vector<std::pair<std::string, std::string>>> pairs(B);
    for (size_t b = 0; b < B; ++b) {
        if (!pcm[b] || !texts[b]) return fail(e, QASR_ERR_INVALID, "align_batch: null clip or text");
        if (n[b] == 0) return fail(e, QASR_ERR_EMPTY_AUDIO, "empty clip");
        try { if (int rc = split_for(e, texts[b], language, pairs[b])) return rc; }
        catch (const std::exception& ex) { return fail(e, QASR_ERR_INVALID, ex.what()); }
    }
    QASR_GUARD(e, {
        e->impl->bind_device();
        e->impl->align_batch(pcm, n, B, pairs);
        for (size_t b = 0; b < B; ++b) {
            auto& r = e->impl->al_batch[b];
            out[b].words = r.view.data();
            out[b].n_words = r.view.size();
            out[b].raw_indices = r.raw.data();
            out[b].n_indices = r.raw.size();
            out[b].passes = 1;
        }
    });
}

int qasr_align_words(qasr_engine* e, const float* pcm, size_t n, int sample_rate, const char* const* surfaces,
                     const char* const* cleaned, size_t n_words, qasr_alignment* out) {
    if (!e || !pcm || !surfaces || !cleaned || !out) return QASR_ERR_INVALID;
    std::vector<std::pair<std::string, std::string>> pairs;
    for (size_t i = 0; i < n_words; ++i) {
        if (!surfaces[i] || !cleaned[i]) return fail(e, QASR_ERR_INVALID, "align_words: null word");
        pairs.push_back({surfaces[i], cleaned[i]});
    }
    return align_common(e, pcm, n, sample_rate, pairs, false, out);
}

// speech-core bridge (VoicePipeline.swift:374-410): strings stay valid until the next transcribe
static sc_transcription_result_t vt_transcribe(void* ctx, const float* audio, size_t length, int sample_rate) {
    qasr_engine* e = static_cast<qasr_engine*>(ctx);
    qasr_result r{};
    sc_transcription_result_t out{};
    int rc = qasr_transcribe(e, audio, length, sample_rate, nullptr, &r);
    if (rc != QASR_OK) {
        e->impl->result_text = std::string("[qasr error: ") + e->impl->last_error + "]";
        out.text = e->impl->result_text.c_str();
    } else {
        out.text = r.text;
    }
    out.language = "";
    out.confidence = 0.0f;      // TranscriptionResult default (Protocols.swift:141)
    out.start_time = 0.0f;
    out.end_time = 0.0f;
    return out;
}
static int32_t vt_rate(void*) { return 16000; }

int qasr_stt_vtable(qasr_engine* e, sc_stt_vtable_t* out) {
    if (!e || !out) return QASR_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    out->context = e;
    out->transcribe = vt_transcribe;
    out->input_sample_rate = vt_rate;
    return QASR_OK;
}

int qasr_decode_structure(qasr_engine* e, int* fused_qa, int* chain, int* launches_per_layer) {
    if (!e || !fused_qa || !chain || !launches_per_layer) return QASR_ERR_INVALID;
    QASR_GUARD(e, e->impl->decode_structure(fused_qa, chain, launches_per_layer));
}

int qasr_set_shared_device(qasr_engine* e, int shared) {
    if (!e) return QASR_ERR_INVALID;
    QASR_GUARD(e, e->impl->set_shared_device(shared != 0));
}

int qasr_set_tuning(const char* key, int value) {
    if (!key) return QASR_ERR_INVALID;
    return qasr::tuning_set(key, value) ? QASR_OK : QASR_ERR_INVALID;
}
int qasr_get_tuning(const char* key, int* value) {
    if (!key || !value) return QASR_ERR_INVALID;
    return qasr::tuning_get(key, value) ? QASR_OK : QASR_ERR_INVALID;
}

int qasr_prefill_logits(qasr_engine* e, const float* audio_embeds, int n_audio, const qasr_options* opt, float* logits) {
    if (!e || !logits || (n_audio > 0 && !audio_embeds)) return QASR_ERR_INVALID;
    if (!e->impl->loaded()) return fail(e, QASR_ERR_NOT_LOADED, "weights not finalized");
    QASR_GUARD(e, e->impl->prefill_logits_host(audio_embeds, n_audio, opt, logits));
}

int qasr_decode_forced(qasr_engine* e, const int32_t* tokens, int n, float* logits) {
    if (!e || !tokens || !logits || n < 0) return QASR_ERR_INVALID;
    QASR_GUARD(e, e->impl->decode_forced_host(tokens, n, logits));
}

// ---- Omnilingual ASR (wav2vec2 + CTC) -------------------------------------------------------------------
}  // extern "C"

struct qasr_ctc_engine {
    std::unique_ptr<qasr::CtcEngine> impl;
};
static int cfail(qasr_ctc_engine* e, int code, const std::string& msg) {
    if (code == QASR_ERR_HIP) (void)hipGetLastError();
    if (e && e->impl) e->impl->last_error = msg; else g_create_error = msg;
    return code;
}
#define QASR_CGUARD(e, body)                                                          \
    try { body; return QASR_OK; }                                                     \
    catch (const qasr::HipError& ex) { return cfail(e, QASR_ERR_HIP, ex.what()); }    \
    catch (const qasr::NotLoaded& ex) { return cfail(e, QASR_ERR_NOT_LOADED, ex.what()); } \
    catch (const std::invalid_argument& ex) { return cfail(e, QASR_ERR_INVALID, ex.what()); } \
    catch (const std::length_error& ex) { return cfail(e, QASR_ERR_CAPACITY, ex.what()); }    \
    catch (const std::exception& ex) { return cfail(e, QASR_ERR_INVALID, ex.what()); }

extern "C" {

int qasr_ctc_default_config(const char* variant, qasr_ctc_config* c) {
    if (!c) return QASR_ERR_INVALID;
    const std::string v = variant ? variant : "300M";
    std::memset(c, 0, sizeof(*c));
    c->feature_dim = 512; c->pos_kernel = 128; c->pos_groups = 16; c->vocab = 10288; c->group_size = 64; c->bits = 4;
    c->ln_eps = 1e-5f; c->device = 0; c->max_batch = 32; c->max_audio_seconds = 40;
    if (v == "tiny") {               // oracle/omnilingual.py OMNI_TINY
        c->model_dim = 64; c->layers = 2; c->heads = 2; c->ffn_dim = 128; c->feature_dim = 32; c->pos_kernel = 16; c->pos_groups = 4;
        c->vocab = 40; c->bits = 16; c->max_batch = 8; c->max_audio_seconds = 10;
        return QASR_OK;
    }
    // OmnilingualMLXConfig.variant (:88-103); detectVariant looks for "CTC-<size>-" in a model id (OmnilingualMLXModel.swift:121-126)
    struct V { const char* name; int d, l, h, f; };
    static const V table[] = {{"300M", 1024, 24, 16, 4096}, {"1B", 1280, 48, 20, 5120}, {"3B", 2048, 60, 32, 8192}, {"7B", 2048, 128, 32, 8192}};
    const V* pick = nullptr;
    for (const V& t : table)
        if (v == t.name || contains(v, (std::string("CTC-") + t.name + "-").c_str())) pick = &t;
    if (!pick) return QASR_ERR_INVALID;
    c->model_dim = pick->d; c->layers = pick->l; c->heads = pick->h; c->ffn_dim = pick->f;
    if (contains(v, "8bit")) c->bits = 8;      // detectBits (:128-132)
    return QASR_OK;
}

int qasr_ctc_create(const char* model_dir, const qasr_ctc_config* cfg, qasr_ctc_engine** out) {
    if (!cfg || !out) return QASR_ERR_INVALID;
    *out = nullptr;
    if (cfg->max_audio_seconds > 40 || cfg->max_audio_seconds <= 0 || cfg->max_batch <= 0) {
        g_create_error = "omnilingual: max_audio_seconds must be in 1..40 (the reference's cap), max_batch positive";
        return QASR_ERR_INVALID;
    }
    qasr_ctc_engine* e = new qasr_ctc_engine();
    try {
        e->impl.reset(new qasr::CtcEngine(*cfg));
        if (model_dir) { e->impl->load_directory(model_dir); e->impl->finalize(); }
    } catch (const qasr::HipError& ex) { g_create_error = ex.what(); delete e; (void)hipGetLastError(); return QASR_ERR_HIP; }
    catch (const std::exception& ex) { g_create_error = ex.what(); delete e; return model_dir ? QASR_ERR_IO : QASR_ERR_INVALID; }
    *out = e;
    return QASR_OK;
}
int qasr_ctc_set_tensor(qasr_ctc_engine* e, const char* name, const void* host, int dtype, const int64_t* shape, int ndim) {
    if (!e || !name || !host || !shape || ndim <= 0) return QASR_ERR_INVALID;
    QASR_CGUARD(e, e->impl->set_tensor(name, host, dtype, shape, ndim));
}
int qasr_ctc_finalize(qasr_ctc_engine* e) { if (!e) return QASR_ERR_INVALID; QASR_CGUARD(e, e->impl->finalize()); }
int qasr_ctc_set_pieces(qasr_ctc_engine* e, const char* const* texts, const int32_t* types, size_t n) {
    if (!e || (!texts && n)) return QASR_ERR_INVALID;
    QASR_CGUARD(e, e->impl->set_pieces(texts, types, n));
}
int qasr_ctc_is_loaded(const qasr_ctc_engine* e) { return e && e->impl->loaded(); }
int qasr_ctc_unload(qasr_ctc_engine* e) { if (!e) return QASR_ERR_INVALID; QASR_CGUARD(e, e->impl->unload()); }
size_t qasr_ctc_memory_footprint(const qasr_ctc_engine* e) { return e ? e->impl->memory_footprint() : 0; }
void qasr_ctc_destroy(qasr_ctc_engine* e) { delete e; }
const char* qasr_ctc_last_error(const qasr_ctc_engine* e) { return e ? e->impl->last_error.c_str() : g_create_error.c_str(); }
int qasr_ctc_num_frames(size_t n) { return qasr::CtcEngine::num_frames((long)n); }

static int ctc_run(qasr_ctc_engine* e, const float* const* pcm, const size_t* n, size_t B, int sample_rate,
                   std::vector<std::vector<int32_t>>& collapsed, float* logits) {
    if (sample_rate != 16000) return cfail(e, QASR_ERR_INVALID, "only 16 kHz input is supported (resampler is out of scope)");
    if (!e->impl->loaded()) return cfail(e, QASR_ERR_NOT_LOADED, "weights not finalized");
    for (size_t b = 0; b < B; ++b) {
        if (!pcm[b] || n[b] == 0) return cfail(e, QASR_ERR_EMPTY_AUDIO, "empty clip in batch");
        // OmnilingualMLXModel.swift:154-159: the 40 s cap is an error, not a truncation
        if ((double)n[b] / 16000.0 > 40.0) return cfail(e, QASR_ERR_CAPACITY, "input exceeds the Omnilingual cap of 40 s");
    }
    QASR_CGUARD(e, {
        std::vector<std::vector<int32_t>> frames;
        e->impl->forward(pcm, n, B, frames, logits);
        collapsed.assign(B, {});
        for (size_t b = 0; b < B; ++b) {                    // collapseConsecutiveDuplicates (:195-209)
            int prev = -1;
            for (int32_t id : frames[b]) if (id != prev) { collapsed[b].push_back(id); prev = id; }
        }
    });
}

int qasr_ctc_transcribe_batch(qasr_ctc_engine* e, const float* const* pcm, const size_t* n, size_t B, int sample_rate,
                              int32_t* ids, size_t stride, int32_t* lens) {
    if (!e || !pcm || !n || !ids || !lens || B == 0) return QASR_ERR_INVALID;
    std::vector<std::vector<int32_t>> col;
    if (int rc = ctc_run(e, pcm, n, B, sample_rate, col, nullptr)) return rc;
    for (size_t b = 0; b < B; ++b) {
        if (col[b].size() > stride) return cfail(e, QASR_ERR_CAPACITY, "ctc_transcribe_batch: id buffer stride too small");
        std::memcpy(ids + b * stride, col[b].data(), col[b].size() * sizeof(int32_t));
        lens[b] = (int32_t)col[b].size();
    }
    return QASR_OK;
}

int qasr_ctc_transcribe(qasr_ctc_engine* e, const float* pcm, size_t n, int sample_rate, const char** text) {
    if (!e || !text) return QASR_ERR_INVALID;
    if (n == 0) { e->impl->result_text.clear(); *text = e->impl->result_text.c_str(); return QASR_OK; }     // :160-162
    if (!pcm) return QASR_ERR_INVALID;
    std::vector<std::vector<int32_t>> col;
    if (int rc = ctc_run(e, &pcm, &n, 1, sample_rate, col, nullptr)) return rc;
    // no SentencePiece vocabulary: an error, not "" for every clip (the reference cannot exist without one, OmnilingualMLXModel.swift:86-98)
    if (!e->impl->has_pieces()) return cfail(e, QASR_ERR_NOT_LOADED, "no SentencePiece vocabulary: load tokenizer.model or call qasr_ctc_set_pieces");
    try { e->impl->result_text = e->impl->detokenize(col[0].data(), (int)col[0].size()); }
    catch (const std::exception& ex) { return cfail(e, QASR_ERR_INVALID, ex.what()); }
    *text = e->impl->result_text.c_str();
    return QASR_OK;
}

int qasr_ctc_logits(qasr_ctc_engine* e, const float* pcm, size_t n, float* logits) {
    if (!e || !pcm || !logits) return QASR_ERR_INVALID;
    std::vector<std::vector<int32_t>> col;
    return ctc_run(e, &pcm, &n, 1, 16000, col, logits);
}

int qasr_ctc_detokenize(qasr_ctc_engine* e, const int32_t* ids, int32_t n, char* buf, size_t cap) {
    if (!e || (!ids && n) || !buf || cap == 0 || n < 0) return -1;
    if (!e->impl->has_pieces()) { cfail(e, QASR_ERR_NOT_LOADED, "no SentencePiece vocabulary"); return -1; }
    try {
        std::string t = e->impl->detokenize(ids, n);
        if (t.size() + 1 > cap) { cfail(e, QASR_ERR_CAPACITY, "detokenize: buffer too small"); return -1; }
        std::memcpy(buf, t.c_str(), t.size() + 1);
        return (int)t.size();
    } catch (const std::exception& ex) { cfail(e, QASR_ERR_INVALID, ex.what()); return -1; }
}

int qasr_ctc_timings(qasr_ctc_engine* e, float ms[4]) {
    if (!e || !ms) return QASR_ERR_INVALID;
    QASR_CGUARD(e, e->impl->timings(ms));
}

int qasr_ctc_greedy(const float* logits, int32_t T, int32_t V, int32_t valid_frames, int32_t* out) {
    if (T < 0 || V <= 0 || (T > 0 && (!logits || !out))) return -QASR_ERR_INVALID;
    return qasr::ctc_greedy_decode(logits, T, V, valid_frames, out);
}

int qasr_layer_normalize(const float* x, size_t n, float eps, float* out) {
    if ((!x || !out) && n) return QASR_ERR_INVALID;
    qasr::layer_normalize_host(x, n, eps, out);
    return QASR_OK;
}

}  // extern "C"

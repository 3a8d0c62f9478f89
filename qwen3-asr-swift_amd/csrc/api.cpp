// api.cpp -- extern "C" boundary (include/qasr.h).  Exceptions never cross it.
#include "engine.h"
#include <cstring>
#include <string>

using qasr::Engine;

struct qasr_engine {
    std::unique_ptr<Engine> impl;
};

static thread_local std::string g_create_error;

static int fail(qasr_engine* e, int code, const std::string& msg) {
    if (e && e->impl) e->impl->last_error = msg; else g_create_error = msg;
    return code;
}

#define QASR_GUARD(e, body)                                                          \
    try { body; return QASR_OK; }                                                    \
    catch (const qasr::HipError& ex) { return fail(e, QASR_ERR_HIP, ex.what()); }    \
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
    if (p == "tiny") {   // test geometry (oracle/config.py AUDIO_TINY / TEXT_TINY / TOKENS_TINY)
        c->enc_d_model = 64; c->enc_heads = 2; c->enc_ffn = 128; c->enc_layers = 2; c->enc_out_dim = 64;
        c->conv_channels = 32; c->n_window_infer = 200;
        c->vocab = 512; c->hidden = 64; c->dec_layers = 2; c->heads = 4; c->kv_heads = 2; c->head_dim = 32; c->inter = 128;
        c->tok_im_start = 500; c->tok_im_end = 501; c->tok_audio_start = 502; c->tok_audio_end = 503;
        c->tok_audio_pad = 504; c->tok_asr_text = 505; c->tok_newline = 198; c->tok_system = 300;
        c->tok_user = 301; c->tok_assistant = 302;
        c->bits = 16; c->max_batch = 8;
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
    } catch (const qasr::HipError& ex) { g_create_error = ex.what(); delete e; return QASR_ERR_HIP; }
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

}  // extern "C"

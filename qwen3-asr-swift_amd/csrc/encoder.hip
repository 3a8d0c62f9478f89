// encoder.hip -- audio encoder orchestration (Qwen3AudioEncoder.callAsFunction,
// Sources/Qwen3ASR/AudioEncoder.swift:362-511) on top of gemm.h / enc_kernels.h.
//
// Data layout in HBM for one batch (img = 100-frame chunk, tokens packed over the whole batch):
//   mel    f32  [B][128][stride]                 (reference layout per clip)
//   c1     bf16 [img][H1=64][W1=50][C]           conv2d1 + GELU          (NHWC)
//   c2     bf16 [img][H2=32][W2=25][C]           conv2d2 + GELU          (implicit GEMM, K = 9C)
//   c3     bf16 [img][W3=13][H3=16][C]           conv2d3 + GELU, pixel order (t, f) so that one
//                                                token's conv_out operand is 16*C contiguous
//   x      f32  [tokens][d_model]                residual stream (valid tokens only, packed)
//   h/qkv/a/mid bf16                             LayerNorm out / fused QKV / attention / FFN mid
//   audio  bf16 [tokens][out_dim]                proj2 output = embeddings spliced into the prompt
#include "engine.h"
#include "ctc_kernels.h"   // f32-parameter LayerNorm and GEMM epilogues (shared with the wav2vec2 path)
#include <cmath>
#include <cstring>

namespace qasr {

static int conv_len(int n) { return (n - 1) / 2 + 1; }

__global__ void permute_conv_out_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int N, int C, int F) {
    // dst[n][f*C + c] = src[n][c*F + f]      (AudioEncoder.swift:423-424 flatten order is c*16 + f)
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)N * C * F;
    if (i >= total) return;
    int n = (int)(i / (C * F)), r = (int)(i - (long)n * C * F);
    int f = r / C, c = r - f * C;
    dst[i] = src[(long)n * C * F + c * F + f];
}

const Tensor& Engine::tensor(const std::string& name) const {
    auto it = tensors_.find(name);
    if (it == tensors_.end()) throw std::runtime_error("missing tensor: " + name);
    return it->second;
}

const bf16_t* Engine::wptr(const std::string& name, std::initializer_list<int64_t> shape) const {
    const Tensor& t = tensor(name);
    if (t.dtype != QASR_DTYPE_BF16) throw std::runtime_error("tensor " + name + ": expected bf16");
    if (t.shape != std::vector<int64_t>(shape)) {
        std::string got;
        for (auto d : t.shape) got += std::to_string(d) + ",";
        throw std::runtime_error("tensor " + name + ": unexpected shape [" + got + "]");
    }
    return t.buf.as<bf16_t>();
}

__global__ void widen_vec_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = bf16_to_f32(src[i]);
}

const float* Engine::fvec(const std::string& name, int64_t n) {
    const Tensor& t = tensor(name);
    if (t.shape != std::vector<int64_t>{n}) throw std::runtime_error("tensor " + name + ": unexpected shape");
    if (t.dtype == QASR_DTYPE_F32) return t.buf.as<float>();
    if (t.dtype != QASR_DTYPE_BF16) throw std::runtime_error("tensor " + name + ": expected bf16 or f32");
    auto buf = std::make_unique<DevBuf>();
    buf->alloc((size_t)n * sizeof(float));
    hipLaunchKernelGGL(widen_vec_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream_, t.buf.as<bf16_t>(), buf->as<float>(), (long)n);
    const float* p = buf->as<float>();
    fused_.push_back(std::move(buf));
    return p;
}

void Engine::finalize_encoder() {
    const int D = cfg_.enc_d_model, C = cfg_.conv_channels, F = cfg_.enc_ffn;
    H1_ = conv_len(cfg_.n_mels); H2_ = conv_len(H1_); H3_ = conv_len(H2_);
    W1_ = conv_len(2 * cfg_.n_window); W2_ = conv_len(W1_); W3_ = conv_len(W2_);
    const std::string a = "audio_tower.";
    encw_.c1w = wptr(a + "conv2d1.weight", {C, 3, 3, 1});
    encw_.c1b = fvec(a + "conv2d1.bias", C);
    encw_.c2w = wptr(a + "conv2d2.weight", {C, 3, 3, C});
    encw_.c2b = fvec(a + "conv2d2.bias", C);
    encw_.c3w = wptr(a + "conv2d3.weight", {C, 3, 3, C});
    encw_.c3b = fvec(a + "conv2d3.bias", C);
    const bf16_t* co = wptr(a + "conv_out.weight", {D, (int64_t)C * H3_});
    {
        auto buf = std::make_unique<DevBuf>();
        buf->alloc((size_t)D * C * H3_ * sizeof(bf16_t));
        long total = (long)D * C * H3_;
        hipLaunchKernelGGL(permute_conv_out_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream_, co,
                           buf->as<bf16_t>(), D, C, H3_);
        encw_.conv_out = buf->as<bf16_t>();
        fused_.push_back(std::move(buf));
    }
    encw_.lnp_g = fvec(a + "ln_post.weight", D);
    encw_.lnp_b = fvec(a + "ln_post.bias", D);
    encw_.p1w = wptr(a + "proj1.weight", {D, D});
    encw_.p1b = fvec(a + "proj1.bias", D);
    encw_.p2w = wptr(a + "proj2.weight", {cfg_.enc_out_dim, D});
    encw_.p2b = fvec(a + "proj2.bias", cfg_.enc_out_dim);
    encw_.layers.clear();
    for (int i = 0; i < cfg_.enc_layers; ++i) {
        const std::string p = a + "layers." + std::to_string(i) + ".";
        EncLayerW L;
        L.ln1_g = fvec(p + "self_attn_layer_norm.weight", D);
        L.ln1_b = fvec(p + "self_attn_layer_norm.bias", D);
        auto wq = std::make_unique<DevBuf>();
        auto bq = std::make_unique<DevBuf>();
        wq->alloc((size_t)3 * D * D * sizeof(bf16_t));
        bq->alloc((size_t)3 * D * sizeof(float));
        const char* names[3] = {"q_proj", "k_proj", "v_proj"};
        for (int j = 0; j < 3; ++j) {
            const bf16_t* w = wptr(p + "self_attn." + names[j] + ".weight", {D, D});
            const float* b = fvec(p + "self_attn." + names[j] + ".bias", D);
            QASR_HIP(hipMemcpyAsync(wq->as<bf16_t>() + (size_t)j * D * D, w, (size_t)D * D * sizeof(bf16_t),
                                    hipMemcpyDeviceToDevice, stream_));
            QASR_HIP(hipMemcpyAsync(bq->as<float>() + (size_t)j * D, b, (size_t)D * sizeof(float),
                                    hipMemcpyDeviceToDevice, stream_));
        }
        L.wqkv = wq->as<bf16_t>();
        L.bqkv = bq->as<float>();
        fused_.push_back(std::move(wq));
        fused_.push_back(std::move(bq));
        L.wo = wptr(p + "self_attn.out_proj.weight", {D, D});
        L.bo = fvec(p + "self_attn.out_proj.bias", D);
        L.ln2_g = fvec(p + "final_layer_norm.weight", D);
        L.ln2_b = fvec(p + "final_layer_norm.bias", D);
        L.w1 = wptr(p + "fc1.weight", {F, D});
        L.b1 = fvec(p + "fc1.bias", F);
        L.w2 = wptr(p + "fc2.weight", {D, F});
        L.b2 = fvec(p + "fc2.bias", D);
        encw_.layers.push_back(L);
    }
    // sinusoid table (AudioEncoder.swift:171-199), float32 arithmetic
    std::vector<float> pe((size_t)W3_ * D);
    const int half = D / 2;
    const float inc = logf(10000.0f) / (float)(half - 1);
    for (int t = 0; t < W3_; ++t)
        for (int i = 0; i < half; ++i) {
            float inv = expf((float)i * (-inc));
            float ang = (float)t * inv;
            pe[(size_t)t * D + i] = sinf(ang);
            pe[(size_t)t * D + half + i] = cosf(ang);
        }
    d_pe_.alloc(pe.size() * sizeof(float));
    QASR_HIP(hipMemcpyAsync(d_pe_.p, pe.data(), pe.size() * sizeof(float), hipMemcpyHostToDevice, stream_));
    QASR_HIP(hipStreamSynchronize(stream_));
    alloc_encoder_workspace();
}

void Engine::alloc_encoder_workspace() {
    const int chunk = 2 * cfg_.n_window, D = cfg_.enc_d_model, C = cfg_.conv_channels;
    const int max_frames = mel_num_frames(max_samples_);
    max_chunks_ = cfg_.max_batch * ((max_frames + chunk - 1) / chunk);
    max_tokens_ = max_chunks_ * W3_;
    d_c1_.alloc((size_t)max_chunks_ * H1_ * W1_ * C * sizeof(bf16_t));
    d_c2_.alloc((size_t)max_chunks_ * H2_ * W2_ * C * sizeof(bf16_t));
    d_c3_.alloc((size_t)max_chunks_ * H3_ * W3_ * C * sizeof(bf16_t));
    // f32 residual stream; also the f32 staging area of the stage entry points (encode_host widens [tokens, enc_out_dim]
    // into it, prefill_logits_host stages [tokens, hidden]), so it is sized for the wider of the two
    d_encx_.alloc((size_t)max_tokens_ * std::max(D, cfg_.enc_out_dim) * sizeof(float));
    d_ench_.alloc((size_t)max_tokens_ * D * sizeof(bf16_t));
    d_encqkv_.alloc((size_t)max_tokens_ * 3 * D * sizeof(bf16_t));
    d_enca_.alloc((size_t)max_tokens_ * D * sizeof(bf16_t));
    d_encmid_.alloc((size_t)max_tokens_ * cfg_.enc_ffn * sizeof(bf16_t));
    d_audio_.alloc((size_t)max_tokens_ * cfg_.enc_out_dim * sizeof(bf16_t));
    size_t meta = (size_t)max_chunks_ * sizeof(ChunkMeta) + (size_t)max_tokens_ * (sizeof(long) + sizeof(int)) +
                  (size_t)(max_tokens_ + 2) * sizeof(int) + 64;
    h_encmeta_.alloc(meta);
    d_encmeta_.alloc(meta);
}

int Engine::num_audio_tokens(int n_frames) const {
    const int chunk = 2 * cfg_.n_window;
    int full = n_frames / chunk, rem = n_frames % chunk;
    int t = full * conv_len(conv_len(conv_len(chunk)));
    if (rem) t += conv_len(conv_len(conv_len(rem)));
    return t;
}

// Host-side integer planning of the current batch (clips_ must be filled).
void Engine::plan_encoder() {
    const int chunk = 2 * cfg_.n_window;
    char* hp = h_encmeta_.as<char>();
    ChunkMeta* hc = reinterpret_cast<ChunkMeta*>(hp);
    n_img_ = 0;
    for (auto& c : clips_) n_img_ += c.n_chunks;
    n_tok_ = 0;
    for (auto& c : clips_) n_tok_ += c.n_tokens;
    if (n_img_ > max_chunks_ || n_tok_ > max_tokens_) throw std::length_error("batch exceeds encoder workspace");
    size_t off = (size_t)max_chunks_ * sizeof(ChunkMeta);
    long* h_rowoff = reinterpret_cast<long*>(hp + off);
    off += (size_t)max_tokens_ * sizeof(long);
    int* h_tok_t = reinterpret_cast<int*>(hp + off);
    size_t off_tok_t = off;
    off += (size_t)max_tokens_ * sizeof(int);
    int* h_cu = reinterpret_cast<int*>(hp + off);
    size_t off_cu = off;
    int img = 0, tok = 0, nwin = 0;
    max_win_ = 0;
    h_cu[0] = 0;
    clip_tok_off_.clear();
    const long tok_stride = (long)H3_ * cfg_.conv_channels;
    for (size_t b = 0; b < clips_.size(); ++b) {
        const ClipPlan& c = clips_[b];
        clip_tok_off_.push_back(tok);
        for (int i = 0; i < c.n_chunks; ++i) {
            ChunkMeta m;
            m.clip = (int)b;
            m.t0 = i * chunk;
            m.clen = (i == c.n_chunks - 1) ? c.last_chunk_len : chunk;
            m.w0 = c.max_chunk_len;
            m.w1 = conv_len(m.w0); m.w2 = conv_len(m.w1); m.w3 = conv_len(m.w2);
            m.tok_off = tok;
            m.n_tok = conv_len(conv_len(conv_len(m.clen)));
            for (int t = 0; t < m.n_tok; ++t) {
                h_rowoff[tok] = ((long)img * W3_ + t) * tok_stride;
                h_tok_t[tok] = t;
                ++tok;
            }
            hc[img++] = m;
        }
        for (int wl : c.windows) { h_cu[nwin + 1] = h_cu[nwin] + wl; ++nwin; max_win_ = std::max(max_win_, wl); }
    }
    n_win_ = nwin;
    if (h_cu[nwin] != n_tok_) throw std::runtime_error("window plan does not cover the packed tokens");
    QASR_HIP(hipMemcpyAsync(d_encmeta_.p, h_encmeta_.p, off_cu + (size_t)(nwin + 1) * sizeof(int),
                            hipMemcpyHostToDevice, stream_));
    char* dp = d_encmeta_.as<char>();
    d_chunks_ = reinterpret_cast<ChunkMeta*>(dp);
    d_tok_rowoff_ = reinterpret_cast<long*>(dp + (size_t)max_chunks_ * sizeof(ChunkMeta));
    d_tok_t_ = reinterpret_cast<int*>(dp + off_tok_t);
    d_cu_win_ = reinterpret_cast<int*>(dp + off_cu);
}

void Engine::run_encoder() {
    if (n_tok_ == 0) return;
    const int D = cfg_.enc_d_model, C = cfg_.conv_channels, F = cfg_.enc_ffn, K9 = 9 * C;
    hipStream_t s = stream_;
    bf16_t *c1 = d_c1_.as<bf16_t>(), *c2 = d_c2_.as<bf16_t>(), *c3 = d_c3_.as<bf16_t>();
    bf16_t *h = d_ench_.as<bf16_t>(), *qkv = d_encqkv_.as<bf16_t>(), *at = d_enca_.as<bf16_t>();
    bf16_t* mid = d_encmid_.as<bf16_t>();
    float* x = d_encx_.as<float>();
    conv1_launch(d_mel_.as<float>(), mel_stride_, cfg_.n_mels, d_chunks_, n_img_, encw_.c1w, encw_.c1b, c1, H1_, W1_, C, s);
    {
        AConv3x3s2 a{c1, H1_, W1_, C, H2_, W2_, n_img_ * H2_ * W2_, K9, true};
        EpiConvGelu e{c2, C, encw_.c2b, d_chunks_, H2_, W2_, true, 2};
        // C >= 64: a K-tile touches at most two taps -> the functor with the per-K-tile scalar decomposition (gemm.h AConv3x3s2W, same bits)
        if (C >= GEMM_BK && tuning().conv_ktile) gemm_nt(AConv3x3s2W{a}, encw_.c2w, K9, a.M, C, K9, e, s);
        else gemm_nt(a, encw_.c2w, K9, a.M, C, K9, e, s);
    }
    {
        AConv3x3s2 a{c2, H2_, W2_, C, H3_, W3_, n_img_ * H3_ * W3_, K9, false};
        EpiConvGelu e{c3, C, encw_.c3b, d_chunks_, H3_, W3_, false, 3};
        if (C >= GEMM_BK && tuning().conv_ktile) gemm_nt(AConv3x3s2W{a}, encw_.c3w, K9, a.M, C, K9, e, s);
        else gemm_nt(a, encw_.c3w, K9, a.M, C, K9, e, s);
    }
    {
        ARowTable a{c3, d_tok_rowoff_, n_tok_, H3_ * C};
        EpiPosF32 e{x, D, d_pe_.as<float>(), d_tok_t_};
        gemm_nt(a, encw_.conv_out, (long)H3_ * C, n_tok_, D, H3_ * C, e, s);
    }
    const int hd = D / cfg_.enc_heads;
    for (const EncLayerW& L : encw_.layers) {
        layernorm_f32p_launch(x, L.ln1_g, L.ln1_b, h, n_tok_, D, cfg_.ln_eps, 0, s);
        gemm_nt(ADense{h, D, n_tok_, D}, L.wqkv, D, n_tok_, 3 * D, D, EpiBiasActBf16F<0>{qkv, 3L * D, L.bqkv}, s);
        // head_dim 64 (both published sizes): the transposed-score 32x32x16 kernel of the wav2vec2 path, a window = a "clip"
        if (hd == 64 && tuning().enc_attn != 0) mha_attention_launch(qkv, d_cu_win_, n_win_, max_win_, cfg_.enc_heads, hd, at, s);
        else window_attention_launch(qkv, d_cu_win_, n_win_, cfg_.enc_heads, hd, at, s);
        gemm_nt(ADense{at, D, n_tok_, D}, L.wo, D, n_tok_, D, D, EpiResidF32F{x, D, L.bo}, s);
        layernorm_f32p_launch(x, L.ln2_g, L.ln2_b, h, n_tok_, D, cfg_.ln_eps, 0, s);
        gemm_nt(ADense{h, D, n_tok_, D}, L.w1, D, n_tok_, F, D, EpiBiasActBf16F<1>{mid, F, L.b1}, s);
        gemm_nt(ADense{mid, F, n_tok_, F}, L.w2, F, n_tok_, D, F, EpiResidF32F{x, D, L.b2}, s);
    }
    layernorm_f32p_launch(x, encw_.lnp_g, encw_.lnp_b, h, n_tok_, D, cfg_.ln_eps, 0, s);
    gemm_nt(ADense{h, D, n_tok_, D}, encw_.p1w, D, n_tok_, D, D, EpiBiasActBf16F<1>{at, D, encw_.p1b}, s);
    gemm_nt(ADense{at, D, n_tok_, D}, encw_.p2w, D, n_tok_, cfg_.enc_out_dim, D,
            EpiBiasActBf16F<0>{d_audio_.as<bf16_t>(), cfg_.enc_out_dim, encw_.p2b}, s);
    QASR_HIP(hipGetLastError());
}

__global__ void widen_bf16_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = bf16_to_f32(src[i]);
}

// Oracle-diff entry: one clip's [n_mels, T] log-mel (host) -> [tokens, out_dim] (host f32).
void Engine::encode_host(const float* mel, int n_frames, float* out) {
    if (!finalized_) throw std::runtime_error("weights not finalized");
    if (n_frames <= 0 || n_frames > mel_num_frames(max_samples_)) throw std::length_error("encode: bad frame count");
    ClipPlan c = plan_clip(cfg_, (long)n_frames * MEL_HOP, 0);
    if (c.frames != n_frames) throw std::runtime_error("encode: plan mismatch");
    clips_.assign(1, c);
    batch_ = 1;
    QASR_HIP(hipMemcpy2DAsync(d_mel_.p, (size_t)mel_stride_ * sizeof(float), mel, (size_t)n_frames * sizeof(float),
                              (size_t)n_frames * sizeof(float), cfg_.n_mels, hipMemcpyHostToDevice, stream_));
    plan_encoder();
    run_encoder();
    long n = (long)n_tok_ * cfg_.enc_out_dim;
    // reuse the f32 residual buffer as the widened staging area
    float* stage = d_encx_.as<float>();
    if ((size_t)n * sizeof(float) > d_encx_.bytes) throw std::length_error("encode: staging too small");
    hipLaunchKernelGGL(widen_bf16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream_, d_audio_.as<bf16_t>(), stage, n);
    QASR_HIP(hipMemcpyAsync(out, stage, n * sizeof(float), hipMemcpyDeviceToHost, stream_));
    QASR_HIP(hipStreamSynchronize(stream_));
}

// qasr_gemm_probe: the GEMM by itself, operands and result through host memory
void Engine::gemm_probe(const uint16_t* A, const uint16_t* W, const float* bias, int M, int N, int K, int form, int reps, float* out,
                        float* avg_ms) {
    DevBuf dA, dW, dB, dO;
    dA.alloc((size_t)M * K * 2); dW.alloc((size_t)N * K * 2); dB.alloc((size_t)N * 4); dO.alloc((size_t)M * N * 4);
    hipStream_t s = stream_;
    QASR_HIP(hipMemcpyAsync(dA.p, A, (size_t)M * K * 2, hipMemcpyHostToDevice, s));
    QASR_HIP(hipMemcpyAsync(dW.p, W, (size_t)N * K * 2, hipMemcpyHostToDevice, s));
    if (bias) QASR_HIP(hipMemcpyAsync(dB.p, bias, (size_t)N * 4, hipMemcpyHostToDevice, s));
    else QASR_HIP(hipMemsetAsync(dB.p, 0, (size_t)N * 4, s));
    hipEvent_t e0, e1;
    QASR_HIP(hipEventCreate(&e0));
    QASR_HIP(hipEventCreate(&e1));
    try {
        const ADense a{dA.as<bf16_t>(), K, M, K};
        const EpiBiasF32 epi{dO.as<float>(), N, dB.as<float>()};
        gemm_nt(a, dW.as<bf16_t>(), K, M, N, K, epi, s, form);            // warm
        QASR_HIP(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) gemm_nt(a, dW.as<bf16_t>(), K, M, N, K, epi, s, form);
        QASR_HIP(hipEventRecord(e1, s));
        QASR_HIP(hipMemcpyAsync(out, dO.p, (size_t)M * N * 4, hipMemcpyDeviceToHost, s));
        QASR_HIP(hipStreamSynchronize(s));
        QASR_HIP(hipGetLastError());
        float ms = 0.f;
        QASR_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (avg_ms) *avg_ms = ms / (float)reps;
    } catch (...) {
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        throw;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

}  // namespace qasr

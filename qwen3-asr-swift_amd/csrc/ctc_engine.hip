// ctc_engine.hip -- Omnilingual ASR engine (see ctc_engine.h).
#include <thread>
#include "ctc_engine.h"
#include "safetensors.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>

namespace qasr {

static const int kKernels[7] = {10, 3, 3, 3, 3, 2, 2};      // OmnilingualMLXConfig.swift:57-58
static const int kStrides[7] = {5, 2, 2, 2, 2, 2, 2};

int CtcEngine::num_frames(long n) {                         // Wav2Vec2Frontend.swift:47-54
    long L = n;
    for (int i = 0; i < 7; ++i) {
        L = (L - kKernels[i]) / kStrides[i] + 1;
        if (L <= 0) return 0;
    }
    return (int)L;
}

static long conv_len(long L, int i) { return L < kKernels[i] ? 0 : (L - kKernels[i]) / kStrides[i] + 1; }

CtcEngine::CtcEngine(const qasr_ctc_config& cfg) : cfg_(cfg) {
    // every dimension positive BEFORE any division or modulo (a zeroed / garbage config is QASR_ERR_INVALID, never SIGFPE)
    if (cfg_.model_dim <= 0 || cfg_.layers <= 0 || cfg_.heads <= 0 || cfg_.ffn_dim <= 0 || cfg_.feature_dim <= 0 || cfg_.pos_kernel <= 0 ||
        cfg_.pos_groups <= 0 || cfg_.vocab <= 0 || cfg_.max_batch <= 0 || cfg_.max_audio_seconds <= 0 || cfg_.max_audio_seconds > 40 ||
        !(cfg_.ln_eps > 0.0f))
        throw std::invalid_argument("omnilingual config: every dimension and capacity must be positive (max_audio_seconds <= 40, the reference's cap)");
    if (cfg_.model_dim % cfg_.heads || cfg_.model_dim % cfg_.pos_groups || (cfg_.model_dim / cfg_.pos_groups) % 8 ||
        cfg_.feature_dim % 8 || cfg_.vocab % 4 || cfg_.model_dim % 8 || cfg_.ffn_dim % 8)
        throw std::invalid_argument("omnilingual config: widths must be multiples of 8 (vocab of 4), heads / groups must divide model_dim");
    const int hd = cfg_.model_dim / cfg_.heads;
    if (hd != 64 && hd != 32) throw std::invalid_argument("omnilingual config: head_dim must be 64 (32 for test geometries)");
    QASR_HIP(hipSetDevice(cfg_.device));
    QASR_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    for (auto& e : ev_) QASR_HIP(hipEventCreate(&e));
    max_samples_ = (long)cfg_.max_audio_seconds * 16000;
    const int B = cfg_.max_batch, C = cfg_.feature_dim, D = cfg_.model_dim;
    long L[7];
    long n = max_samples_;
    for (int i = 0; i < 7; ++i) { n = conv_len(n, i); L[i] = n; }
    cap_frames_ = (int)(B * std::max<long>(L[6], 1));
    cap_conv_rows_ = B * std::max<long>(L[1], 1);
    h_pcm_.alloc((size_t)B * (max_samples_ + 2) * sizeof(float));
    d_pcm_.alloc((size_t)B * (max_samples_ + 2) * sizeof(float));
    const size_t meta = (size_t)B * (sizeof(long) + sizeof(int)) + (size_t)7 * (2 * B + 1) * sizeof(int) + 64;
    h_meta_.alloc(meta);
    d_meta_.alloc(meta);
    d_stats_.alloc((size_t)2 * B * sizeof(float));
    d_act_[0].alloc((size_t)B * std::max<long>(L[0], 1) * C * sizeof(bf16_t));
    d_act_[1].alloc((size_t)B * std::max<long>(L[1], 1) * C * sizeof(bf16_t));
    d_convf_.alloc((size_t)cap_conv_rows_ * C * sizeof(float));
    d_rows_.alloc((size_t)cap_conv_rows_ * sizeof(long));
    d_x_.alloc((size_t)cap_frames_ * D * sizeof(float));
    d_y_.alloc((size_t)cap_frames_ * std::max(D, C) * sizeof(float));   // also holds the last extractor layer's [frames][C] f32 output
    d_h_.alloc((size_t)cap_frames_ * std::max(D, C) * sizeof(bf16_t));
    d_qkv_.alloc((size_t)cap_frames_ * 3 * D * sizeof(bf16_t));
    d_att_.alloc((size_t)cap_frames_ * D * sizeof(bf16_t));
    d_mid_.alloc((size_t)cap_frames_ * cfg_.ffn_dim * sizeof(bf16_t));
    d_logits_.alloc((size_t)cap_frames_ * cfg_.vocab * sizeof(float));
    d_ids_.alloc(((size_t)cap_frames_ + 1) * sizeof(int));   // + the non-finite flag
    d_info_.alloc((size_t)cap_frames_ * sizeof(int2));
}

CtcEngine::~CtcEngine() {
    if (stream_) (void)hipStreamSynchronize(stream_);
    for (auto& e : ev_) if (e) (void)hipEventDestroy(e);
    if (stream_) (void)hipStreamDestroy(stream_);
}

// The reference widens every non-uint32 tensor to float32 at load (OmnilingualMLXWeightLoader.swift:26-37): so does this.
void CtcEngine::set_tensor(const std::string& name, const void* host, int dtype, const int64_t* shape, int ndim) {
    QASR_HIP(hipStreamSynchronize(stream_));
    finalized_ = false;
    Tensor& t = tensors_[name];
    t.shape.assign(shape, shape + ndim);
    const size_t n = t.numel();
    if (dtype == QASR_DTYPE_U32) {
        t.dtype = QASR_DTYPE_U32;
        t.buf.alloc(n * 4);
        QASR_HIP(hipMemcpy(t.buf.p, host, n * 4, hipMemcpyHostToDevice));
        return;
    }
    std::vector<float> wide;
    const float* src = reinterpret_cast<const float*>(host);
    if (dtype == QASR_DTYPE_BF16) {
        wide.resize(n);
        const bf16_t* s = reinterpret_cast<const bf16_t*>(host);
        for (size_t i = 0; i < n; ++i) wide[i] = bf16_to_f32_host(s[i]);
        src = wide.data();
    } else if (dtype == QASR_DTYPE_F16) {
        wide.resize(n);
        SafeEntry e{"F16", {}, reinterpret_cast<const uint8_t*>(host), n * 2};
        for (size_t i = 0; i < n; ++i) wide[i] = safe_elem_f32(e, i);
        src = wide.data();
    } else if (dtype != QASR_DTYPE_F32) throw std::invalid_argument("tensor " + name + ": unsupported dtype");
    t.dtype = QASR_DTYPE_F32;
    t.buf.alloc(n * 4);
    QASR_HIP(hipMemcpy(t.buf.p, src, n * 4, hipMemcpyHostToDevice));
}

void CtcEngine::load_directory(const std::string& dir) {
    SafeTensorsDir st(dir);
    std::vector<float> wide;
    for (auto& kv : st.entries) {
        const SafeEntry& e = kv.second;
        const size_t n = e.numel();
        if (e.dtype == "U32") {
            if (n * 4 != e.bytes) throw std::runtime_error("tensor " + kv.first + ": byte size does not match shape");
            set_tensor(kv.first, e.data, QASR_DTYPE_U32, e.shape.data(), (int)e.shape.size());
        } else if (e.dtype == "F32" || e.dtype == "F16" || e.dtype == "BF16") {
            if (n * (e.dtype == "F32" ? 4 : 2) != e.bytes) throw std::runtime_error("tensor " + kv.first + ": byte size does not match shape");
            wide.resize(n);
            for (size_t i = 0; i < n; ++i) wide[i] = safe_elem_f32(e, i);
            set_tensor(kv.first, wide.data(), QASR_DTYPE_F32, e.shape.data(), (int)e.shape.size());
        }
    }
    // the reference refuses to load without the SentencePiece model (OmnilingualMLXModel.swift:86-92): a mis-staged directory is an
    // error here too, not an engine that answers "" for every clip
    std::ifstream probe(dir + "/tokenizer.model", std::ios::binary);
    if (!probe.good()) throw std::runtime_error("tokenizer.model not found at " + dir + "/tokenizer.model");
    load_sentencepiece(dir + "/tokenizer.model");
}

const Tensor& CtcEngine::tensor(const std::string& name) const {
    auto it = tensors_.find(name);
    if (it == tensors_.end()) throw std::runtime_error("missing tensor " + name);
    return it->second;
}

const float* CtcEngine::f32_param(const std::string& name, std::initializer_list<int64_t> shape) {
    const Tensor& t = tensor(name);
    if (t.dtype != QASR_DTYPE_F32) throw std::runtime_error("tensor " + name + ": expected a float tensor");
    size_t want = 1;
    for (auto d : shape) want *= (size_t)d;
    if (t.numel() != want) throw std::runtime_error("tensor " + name + ": unexpected element count");
    return t.buf.as<float>();
}

void* CtcEngine::new_buf(size_t bytes) {
    auto b = std::make_unique<DevBuf>();
    b->alloc(bytes);
    void* p = b->p;
    built_.push_back(std::move(b));
    return p;
}

// `stem.weight` float [N][K], or an MLX triplet (uint32 words + f32 scales / biases) -> bf16 [N][K] rows at dst.
// Quantised: bf16(scale * q + bias): the reference multiplies by the unrounded f32 value; rounding the MFMA operand to bf16
// is this path's stated deviation (oracle policy DEVICE).
void CtcEngine::linear_weight(const std::string& stem, int N, int K, bf16_t* dst) {
    if (tensors_.count(stem + ".scales")) {
        const Tensor &w = tensor(stem + ".weight"), &sc = tensor(stem + ".scales"), &bi = tensor(stem + ".biases");
        const int bits = cfg_.bits;
        if ((bits != 4 && bits != 8) || cfg_.group_size != 64 || K % 64) throw std::runtime_error(stem + ": quantised weights need bits 4 / 8, group 64");
        if (w.dtype != QASR_DTYPE_U32 || w.shape != std::vector<int64_t>{N, (int64_t)K * bits / 32} ||
            sc.shape != std::vector<int64_t>{N, (int64_t)K / 64} || bi.shape != sc.shape || sc.dtype != QASR_DTYPE_F32 || bi.dtype != QASR_DTYPE_F32)
            throw std::runtime_error(stem + ": quantised triplet has unexpected shapes (bits?)");
        QuantRaw q;
        q.wq = w.buf.as<uint32_t>(); q.scales = sc.buf.p; q.biases = bi.buf.p; q.sb_f32 = 1; q.N = N; q.K = K; q.bits = bits;
        quant_dequant_rows_launch(q, 0, N, dst, stream_);
    } else {
        const float* w = f32_param(stem + ".weight", {N, K});
        cast_f32_bf16_launch(w, dst, (long)N * K, stream_);
    }
}

// PyTorch Conv1d weight [O][I][K] f32 -> [O][K][I] bf16 (CommonWeightLoader.applyConv1dWeights transpose: true)
__global__ void conv_w_transpose_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int O, int I, int K) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)O * I * K) return;
    const int i = (int)(idx % I), k = (int)((idx / I) % K);
    const long o = idx / ((long)I * K);
    dst[idx] = f32_to_bf16(src[(o * I + i) * K + k]);
}
// weight_norm(dim = 2): norm[k] = sqrt(sum_{o,i} v[o][i][k]^2)   (OmnilingualMLXWeightLoader.swift:92-103)
__global__ void wn_norm_kernel(const float* __restrict__ v, float* __restrict__ norm, long OI, int K) {
    __shared__ float s[4];
    const int k = blockIdx.x;
    float acc = 0.0f;
    for (long j = threadIdx.x; j < OI; j += blockDim.x) { const float x = v[j * K + k]; acc = fmaf(x, x, acc); }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) norm[k] = sqrtf(s[0] + s[1] + s[2] + s[3]);
}
__global__ void wn_fuse_kernel(const float* __restrict__ g, const float* __restrict__ v, const float* __restrict__ norm,
                               bf16_t* __restrict__ dst, int O, int I, int K) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;       // dst index [o][k][i]
    if (idx >= (long)O * I * K) return;
    const int i = (int)(idx % I), k = (int)((idx / I) % K);
    const long o = idx / ((long)I * K);
    dst[idx] = f32_to_bf16(g[k] * v[(o * I + i) * K + k] / fmaxf(norm[k], 1e-12f));
}

void CtcEngine::finalize() {
    QASR_HIP(hipStreamSynchronize(stream_));
    built_.clear();
    finalized_ = false;
    const int C = cfg_.feature_dim, D = cfg_.model_dim, F = cfg_.ffn_dim, V = cfg_.vocab, KP = cfg_.pos_kernel, cpg = D / cfg_.pos_groups;
    const std::string fe = "encoder_frontend.feature_extractor.layers.";
    for (int i = 0; i < 7; ++i) {
        const std::string p = fe + std::to_string(i);
        const int in = i == 0 ? 1 : C, k = kKernels[i];
        const float* w = f32_param(p + ".conv.weight", {C, in, k});
        if (i == 0) conv_[i].w = w;                                      // [C][1][10] == [C][10]
        else {
            bf16_t* d = (bf16_t*)new_buf((size_t)C * k * in * sizeof(bf16_t));
            hipLaunchKernelGGL(conv_w_transpose_kernel, dim3(cdiv((long)C * in * k, 256)), dim3(256), 0, stream_, w, d, C, in, k);
            conv_[i].w = d;
        }
        conv_[i].b = f32_param(p + ".conv.bias", {C});
        conv_[i].ln_g = f32_param(p + ".layer_norm.weight", {C});
        conv_[i].ln_b = f32_param(p + ".layer_norm.bias", {C});
    }
    post_g_ = f32_param("encoder_frontend.post_extract_layer_norm.weight", {C});
    post_b_ = f32_param("encoder_frontend.post_extract_layer_norm.bias", {C});
    {
        bf16_t* w = (bf16_t*)new_buf((size_t)D * C * sizeof(bf16_t));
        linear_weight("encoder_frontend.model_dim_proj", D, C, w);
        proj_ = {w, f32_param("encoder_frontend.model_dim_proj.bias", {D})};
    }
    {
        const std::string p = "encoder_frontend.pos_encoder.conv";
        const float* g = f32_param(p + ".weight_g", {1, 1, KP});
        const float* v = f32_param(p + ".weight_v", {D, cpg, KP});
        float* norm = (float*)new_buf((size_t)KP * sizeof(float));
        hipLaunchKernelGGL(wn_norm_kernel, dim3(KP), dim3(256), 0, stream_, v, norm, (long)D * cpg, KP);
        bf16_t* w = (bf16_t*)new_buf((size_t)D * KP * cpg * sizeof(bf16_t));
        hipLaunchKernelGGL(wn_fuse_kernel, dim3(cdiv((long)D * cpg * KP, 256)), dim3(256), 0, stream_, g, v, norm, w, D, cpg, KP);
        pos_w_ = w;
        if (tensors_.count(p + ".bias")) pos_b_ = f32_param(p + ".bias", {D});
        else {   // optional in the reference (OmnilingualMLXWeightLoader.swift:80-82): the Conv1d keeps its zero-initialised bias
            float* z = (float*)new_buf((size_t)D * sizeof(float));
            QASR_HIP(hipMemsetAsync(z, 0, (size_t)D * sizeof(float), stream_));
            pos_b_ = z;
        }
    }
    layers_.clear();
    for (int l = 0; l < cfg_.layers; ++l) {
        const std::string p = "encoder.layers." + std::to_string(l);
        Layer L{};
        L.ln1_g = f32_param(p + ".self_attn_layer_norm.weight", {D});
        L.ln1_b = f32_param(p + ".self_attn_layer_norm.bias", {D});
        L.ln2_g = f32_param(p + ".ffn_layer_norm.weight", {D});
        L.ln2_b = f32_param(p + ".ffn_layer_norm.bias", {D});
        bf16_t* wqkv = (bf16_t*)new_buf((size_t)3 * D * D * sizeof(bf16_t));
        float* bqkv = (float*)new_buf((size_t)3 * D * sizeof(float));
        const char* names[3] = {"q_proj", "k_proj", "v_proj"};
        for (int j = 0; j < 3; ++j) {
            const std::string s = p + ".self_attn." + names[j];
            linear_weight(s, D, D, wqkv + (size_t)j * D * D);
            QASR_HIP(hipMemcpyAsync(bqkv + (size_t)j * D, f32_param(s + ".bias", {D}), (size_t)D * 4, hipMemcpyDeviceToDevice, stream_));
        }
        L.qkv = {wqkv, bqkv};
        bf16_t* wo = (bf16_t*)new_buf((size_t)D * D * sizeof(bf16_t));
        linear_weight(p + ".self_attn.output_proj", D, D, wo);
        L.o = {wo, f32_param(p + ".self_attn.output_proj.bias", {D})};
        bf16_t* w1 = (bf16_t*)new_buf((size_t)F * D * sizeof(bf16_t));
        linear_weight(p + ".ffn.inner_proj", F, D, w1);
        L.f1 = {w1, f32_param(p + ".ffn.inner_proj.bias", {F})};
        bf16_t* w2 = (bf16_t*)new_buf((size_t)D * F * sizeof(bf16_t));
        linear_weight(p + ".ffn.output_proj", D, F, w2);
        L.f2 = {w2, f32_param(p + ".ffn.output_proj.bias", {D})};
        layers_.push_back(L);
    }
    final_g_ = f32_param("encoder.layer_norm.weight", {D});
    final_b_ = f32_param("encoder.layer_norm.bias", {D});
    {
        bf16_t* w = (bf16_t*)new_buf((size_t)V * D * sizeof(bf16_t));
        linear_weight("final_proj", V, D, w);
        head_ = {w, f32_param("final_proj.bias", {V})};
    }
    QASR_HIP(hipStreamSynchronize(stream_));
    QASR_HIP(hipGetLastError());
    finalized_ = true;
}

void CtcEngine::unload() {
    QASR_HIP(hipStreamSynchronize(stream_));
    tensors_.clear();
    built_.clear();
    layers_.clear();
    finalized_ = false;
}

size_t CtcEngine::memory_footprint() const {
    size_t n = 0;
    for (auto& kv : tensors_) n += kv.second.buf.bytes;
    return n;
}

void CtcEngine::forward(const float* const* pcm, const size_t* n, size_t Bz, std::vector<std::vector<int32_t>>& frame_ids, float* logits) {
    if (!finalized_) throw NotLoaded("omnilingual: weights not finalized");
    const int B = (int)Bz, C = cfg_.feature_dim, D = cfg_.model_dim, F = cfg_.ffn_dim, V = cfg_.vocab;
    if (B <= 0) throw std::invalid_argument("empty batch");
    if (B > cfg_.max_batch) throw std::length_error("batch exceeds max_batch");
    QASR_HIP(hipStreamSynchronize(stream_));              // pinned staging is reused
    // ---- plan (host, integer): per clip output length of every conv layer ------------------------------------
    char* hm = h_meta_.as<char>();
    long* h_off = reinterpret_cast<long*>(hm);
    int* h_ns = reinterpret_cast<int*>(h_off + B);
    int* h_lv = h_ns + B;                                   // 7 levels x ([B + 1] offsets | [B] lengths)
    auto lv_off = [&](int i) { return h_lv + (size_t)i * (2 * B + 1); };
    auto lv_n = [&](int i) { return lv_off(i) + B + 1; };
    long off = 0;
    for (int b = 0; b < B; ++b) {
        if (n[b] == 0) throw std::invalid_argument("empty clip");
        if ((long)n[b] > max_samples_) throw std::length_error("clip longer than max_audio_seconds");
        h_off[b] = off;
        h_ns[b] = (int)n[b];
        off += ((long)n[b] + 1) & ~1L;
        long L = (long)n[b];
        for (int i = 0; i < 7; ++i) { L = conv_len(L, i); lv_n(i)[b] = (int)L; }
    }
    // caller memory (pageable) -> pinned staging -> HBM in up to four slices of whole clips: a slice is staged on a few threads and
    // its H2D copy queued at once, so staging of slice i + 1 overlaps the copy of slice i (61 MB at 32 x 30 s)
    {
        const int nthr = off > (1L << 20) ? std::min(B, 8) : 1;
        const int nsl = off > (8L << 20) ? std::min(B, 4) : 1;
        for (int sidx = 0; sidx < nsl; ++sidx) {
            const int s0 = B * sidx / nsl, s1 = B * (sidx + 1) / nsl, nb = s1 - s0;
            auto copy_range = [&](int b0, int b1) {
                for (int b = b0; b < b1; ++b) std::memcpy(h_pcm_.as<float>() + h_off[b], pcm[b], n[b] * sizeof(float));
            };
            const int t_n = std::min(nthr, nb);
            if (t_n <= 1) copy_range(s0, s1);
            else {
                std::vector<std::thread> pool;
                for (int t = 0; t < t_n; ++t) pool.emplace_back(copy_range, s0 + nb * t / t_n, s0 + nb * (t + 1) / t_n);
                for (auto& th : pool) th.join();
            }
            const long e0 = h_off[s0], e1 = s1 < B ? h_off[s1] : off;
            QASR_HIP(hipMemcpyAsync(d_pcm_.as<float>() + e0, h_pcm_.as<float>() + e0, (size_t)(e1 - e0) * sizeof(float), hipMemcpyHostToDevice,
                                    stream_));
        }
    }
    int max_n0 = 0, max_frames = 0;
    for (int i = 0; i < 7; ++i) {
        int acc = 0;
        for (int b = 0; b < B; ++b) { lv_off(i)[b] = acc; acc += lv_n(i)[b]; }
        lv_off(i)[B] = acc;
    }
    for (int b = 0; b < B; ++b) { max_n0 = std::max(max_n0, lv_n(0)[b]); max_frames = std::max(max_frames, lv_n(6)[b]); }
    const int Ftot = lv_off(6)[B];
    const size_t meta_bytes = (size_t)B * (sizeof(long) + sizeof(int)) + (size_t)7 * (2 * B + 1) * sizeof(int);
    hipStream_t s = stream_;
    QASR_HIP(hipMemcpyAsync(d_meta_.p, h_meta_.p, meta_bytes, hipMemcpyHostToDevice, s));
    const long* d_off = d_meta_.as<long>();
    const int* d_ns = reinterpret_cast<const int*>(d_off + B);
    const int* d_lv = d_ns + B;
    auto dv_off = [&](int i) { return d_lv + (size_t)i * (2 * B + 1); };
    auto dv_n = [&](int i) { return dv_off(i) + B + 1; };
    frame_ids.assign(B, {});
    QASR_HIP(hipEventRecord(ev_[0], s));
    if (Ftot > 0) {
        // ---- feature extractor -------------------------------------------------------------------------------
        wave_stats_launch(d_pcm_.as<float>(), d_off, d_ns, B, 1e-5f, d_stats_.as<float>(), s);       // layerNormEpsilon, :30
        w2v_conv0_launch(d_pcm_.as<float>(), d_off, d_stats_.as<float>(), dv_off(0), dv_n(0), B, max_n0, (const float*)conv_[0].w,
                         conv_[0].b, conv_[0].ln_g, conv_[0].ln_b, 1e-5f, d_act_[0].as<bf16_t>(), C, s);
        float* feats = d_convf_.as<float>();
        for (int i = 1; i < 7; ++i) {
            const int tot = lv_off(i)[B], K = kKernels[i] * C;
            bf16_t* in = d_act_[(i - 1) & 1].as<bf16_t>();
            w2v_conv_rows_launch(dv_off(i - 1), dv_off(i), dv_n(i), B, tot, kStrides[i], C, d_rows_.as<long>(), s);
            gemm_nt(ARowTable{in, d_rows_.as<long>(), tot, K}, (const bf16_t*)conv_[i].w, K, tot, C, K, EpiBiasF32{feats, C, conv_[i].b}, s);
            // LayerNorm + GELU; the last layer's output feeds another LayerNorm: keep it in f32 through d_x_ (scratch here)
            if (i < 6) layernorm_f32p_launch(feats, conv_[i].ln_g, conv_[i].ln_b, d_act_[i & 1].as<bf16_t>(), tot, C, 1e-5f, 1, s);
            else layernorm_gelu_f32_launch(feats, conv_[i].ln_g, conv_[i].ln_b, d_y_.as<float>(), tot, C, 1e-5f, s);
        }
        // ---- post-extract LayerNorm, Linear(C -> D), positional encoder -----------------------------------------
        bf16_t* h = d_h_.as<bf16_t>();
        float *x = d_x_.as<float>(), *y = d_y_.as<float>();
        layernorm_f32p_launch(y, post_g_, post_b_, h, Ftot, C, 1e-5f, 0, s);
        gemm_nt(ADense{h, C, Ftot, C}, proj_.w, C, Ftot, D, C, EpiBiasF32{x, D, proj_.b}, s);
        cast_f32_bf16_launch(x, h, (long)Ftot * D, s);
        w2v_frame_info_launch(dv_off(6), dv_n(6), B, Ftot, d_info_.as<int2>(), s);
        const int KP = cfg_.pos_kernel, cpg = D / cfg_.pos_groups;
        // all groups in one launch: a group alone is 375 tiles at 300M, a third of what the chip holds at once
        gemm_nt_groups(AGroupConv1d{h, d_info_.as<int2>(), D, cpg, KP, 0, Ftot}, pos_w_, (long)KP * cpg, (long)cpg * KP * cpg, cfg_.pos_groups,
                       Ftot, cpg, KP * cpg, EpiPosConv{y, x, D, pos_b_, 0, cpg}, s);
        std::swap(x, y);                                     // x = frontend output
        QASR_HIP(hipEventRecord(ev_[1], s));
        // ---- transformer encoder --------------------------------------------------------------------------------
        bf16_t *qkv = d_qkv_.as<bf16_t>(), *att = d_att_.as<bf16_t>(), *mid = d_mid_.as<bf16_t>();
        for (const Layer& L : layers_) {
            layernorm_f32p_launch(x, L.ln1_g, L.ln1_b, h, Ftot, D, cfg_.ln_eps, 0, s);
            gemm_nt(ADense{h, D, Ftot, D}, L.qkv.w, D, Ftot, 3 * D, D, EpiBiasActBf16F<0>{qkv, 3L * D, L.qkv.b}, s);
            mha_attention_launch(qkv, dv_off(6), B, max_frames, cfg_.heads, D / cfg_.heads, att, s);
            gemm_nt(ADense{att, D, Ftot, D}, L.o.w, D, Ftot, D, D, EpiResidF32F{x, D, L.o.b}, s);
            layernorm_f32p_launch(x, L.ln2_g, L.ln2_b, h, Ftot, D, cfg_.ln_eps, 0, s);
            gemm_nt(ADense{h, D, Ftot, D}, L.f1.w, D, Ftot, F, D, EpiBiasActBf16F<1>{mid, F, L.f1.b}, s);
            gemm_nt(ADense{mid, F, Ftot, F}, L.f2.w, F, Ftot, D, F, EpiResidF32F{x, D, L.f2.b}, s);
        }
        QASR_HIP(hipEventRecord(ev_[2], s));
        // ---- final LayerNorm, CTC head, per-frame argmax -------------------------------------------------------------
        layernorm_f32p_launch(x, final_g_, final_b_, h, Ftot, D, cfg_.ln_eps, 0, s);
        gemm_nt(ADense{h, D, Ftot, D}, head_.w, D, Ftot, V, D, EpiBiasF32{d_logits_.as<float>(), V, head_.b}, s);
        int* d_err = d_ids_.as<int>() + cap_frames_;
        QASR_HIP(hipMemsetAsync(d_err, 0, sizeof(int), s));
        argmax_f32_launch(d_logits_.as<float>(), V, Ftot, V, d_ids_.as<int>(), d_err, s);
    } else {
        QASR_HIP(hipEventRecord(ev_[1], s));
        QASR_HIP(hipEventRecord(ev_[2], s));
    }
    QASR_HIP(hipEventRecord(ev_[3], s));
    std::vector<int> ids((size_t)std::max(Ftot, 1));
    int bad = 0;
    if (Ftot > 0) {
        QASR_HIP(hipMemcpyAsync(ids.data(), d_ids_.p, (size_t)Ftot * sizeof(int), hipMemcpyDeviceToHost, s));
        QASR_HIP(hipMemcpyAsync(&bad, d_ids_.as<int>() + cap_frames_, sizeof(int), hipMemcpyDeviceToHost, s));
        if (logits) QASR_HIP(hipMemcpyAsync(logits, d_logits_.p, (size_t)Ftot * V * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    QASR_HIP(hipStreamSynchronize(s));
    QASR_HIP(hipGetLastError());
    if (bad) throw HipError("omnilingual: non-finite logits (corrupt weights or input)");
    for (int b = 0; b < B; ++b) frame_ids[b].assign(ids.begin() + lv_off(6)[b], ids.begin() + lv_off(6)[b] + lv_n(6)[b]);
}

void CtcEngine::timings(float ms[4]) {
    QASR_HIP(hipStreamSynchronize(stream_));
    for (int i = 0; i < 3; ++i) QASR_HIP(hipEventElapsedTime(&ms[i], ev_[i], ev_[i + 1]));
    QASR_HIP(hipEventElapsedTime(&ms[3], ev_[0], ev_[3]));
}

// ---- host logic ---------------------------------------------------------------------------------------
// CTCGreedyDecoder.decode (CTCGreedyDecoder.swift:28-55): strict '>' keeps the first maximum; duplicates collapse; blank stays
int ctc_greedy_decode(const float* logits, int T, int V, int valid_frames, int32_t* out) {
    const int frames = valid_frames >= 0 && valid_frames < T ? valid_frames : T;
    int n = 0, prev = -1;
    for (int t = 0; t < frames; ++t) {
        const float* row = logits + (size_t)t * V;
        int best = 0;
        float bv = row[0];
        for (int v = 1; v < V; ++v)
            if (row[v] > bv) { bv = row[v]; best = v; }
        if (best != prev) { out[n++] = best; prev = best; }
    }
    return n;
}

// OmnilingualASRModel.layerNormalize (OmnilingualASR.swift:305-325): sequential f32 sums
void layer_normalize_host(const float* x, size_t n, float eps, float* out) {
    if (n == 0) return;
    float sum = 0.0f, sq = 0.0f;
    for (size_t i = 0; i < n; ++i) { sum += x[i]; sq += x[i] * x[i]; }
    const float mean = sum / (float)n;
    const float var = std::max(0.0f, sq / (float)n - mean * mean);
    const float inv = 1.0f / std::sqrt(var + eps);
    for (size_t i = 0; i < n; ++i) out[i] = (x[i] - mean) * inv;
}

void CtcEngine::set_pieces(const char* const* texts, const int32_t* types, size_t n) {
    pieces_.clear();
    for (size_t i = 0; i < n; ++i) pieces_.push_back({texts[i] ? texts[i] : "", types ? types[i] : 1});
}

// tokenizer.model = serialized sentencepiece ModelProto: field 1 (repeated, length-delimited) SentencePiece
// { 1: piece (string), 2: score (float, fixed32), 3: type (varint enum, default NORMAL = 1) }
void CtcEngine::load_sentencepiece(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    f.seekg(0, std::ios::end);
    const std::streamoff size = f.tellg();
    if (size < 0 || size > (std::streamoff)(256 << 20)) throw std::runtime_error("sentencepiece model: implausible file size");   // published: 0.4 MB
    f.seekg(0);
    std::string d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    pieces_.clear();
    size_t p = 0;
    auto varint = [&](const std::string& s, size_t& q, size_t end) {
        uint64_t v = 0;
        for (int sh = 0; q < end && sh < 64; sh += 7) {
            const uint8_t c = (uint8_t)s[q++];
            v |= (uint64_t)(c & 0x7F) << sh;
            if (!(c & 0x80)) return v;
        }
        throw std::runtime_error("sentencepiece model: bad varint");
    };
    auto skip = [&](const std::string& s, size_t& q, size_t end, int wire) {
        if (wire == 0) (void)varint(s, q, end);
        else if (wire == 1) q += 8;
        else if (wire == 5) q += 4;
        else if (wire == 2) { const uint64_t n = varint(s, q, end); if (n > end - q) throw std::runtime_error("sentencepiece model: truncated field"); q += n; }
        else throw std::runtime_error("sentencepiece model: unsupported wire type");
        if (q > end) throw std::runtime_error("sentencepiece model: truncated");
    };
    while (p < d.size()) {
        const uint64_t key = varint(d, p, d.size());
        const int field = (int)(key >> 3), wire = (int)(key & 7);
        if (field == 1 && wire == 2) {
            const uint64_t len = varint(d, p, d.size());
            if (len > d.size() - p) throw std::runtime_error("sentencepiece model: truncated piece");
            const size_t end = p + len;
            std::string text;
            int type = 1;
            while (p < end) {
                const uint64_t k2 = varint(d, p, end);
                const int f2 = (int)(k2 >> 3), w2 = (int)(k2 & 7);
                if (f2 == 1 && w2 == 2) {
                    const uint64_t n = varint(d, p, end);
                    if (n > end - p) throw std::runtime_error("sentencepiece model: truncated string");
                    text.assign(d, p, n);
                    p += n;
                } else if (f2 == 3 && w2 == 0) type = (int)varint(d, p, end);
                else skip(d, p, end, w2);
            }
            pieces_.push_back({text, type});
        } else skip(d, p, d.size(), wire);
    }
    if (pieces_.empty()) throw std::runtime_error("sentencepiece model at " + path + " contained no pieces");
}

// OmnilingualVocabulary.decode (SentencePieceVocabulary.swift:38-57): ids {bos 0, pad 1, eos 2, unk 3} and control (3) /
// unknown (2) / unused (5) / byte (6) pieces are dropped, U+2581 -> space, surrounding whitespace trimmed
std::string CtcEngine::detokenize(const int32_t* ids, int n) const {
    std::string out;
    for (int i = 0; i < n; ++i) {
        const int id = ids[i];
        if (id < 0 || id >= (int)pieces_.size()) continue;
        if (id <= 3) continue;
        const int t = pieces_[id].second;
        if (t == 2 || t == 3 || t == 5 || t == 6) continue;
        out += pieces_[id].first;
    }
    std::string sp;
    for (size_t i = 0; i < out.size();) {
        if (i + 2 < out.size() && (uint8_t)out[i] == 0xE2 && (uint8_t)out[i + 1] == 0x96 && (uint8_t)out[i + 2] == 0x81) { sp += ' '; i += 3; }
        else sp += out[i++];
    }
    size_t a = 0, b = sp.size();
    while (a < b && (sp[a] == ' ' || sp[a] == '\t')) ++a;
    while (b > a && (sp[b - 1] == ' ' || sp[b - 1] == '\t')) --b;
    return sp.substr(a, b - a);
}

}  // namespace qasr

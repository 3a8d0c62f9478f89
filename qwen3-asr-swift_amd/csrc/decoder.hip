// decoder.hip -- text decoder orchestration: prompt build + splice, prompt pass (packed, varlen),
// greedy decode loop with a hipGraph-captured step, batch pipeline and stage entry points.
//
// Reference: Sources/Qwen3ASR/Qwen3ASR.swift:173-294 (generateText), :317-390 (greedy loop),
// QuantizedTextDecoder.swift / FloatTextDecoder.swift (model), PreQuantizedEmbedding.swift:35-49.
//
// HBM layout (per engine):
//   K cache    bf16 [layer][slot][kv_head][max_ctx][head_dim]               static, appended in place
//   V cache    bf16 [layer][slot][kv_head][max_ctx/32][head_dim/16][64][8]  MFMA-fragment order (vfrag_index), appended in place
//   V rows     bf16 [slot][kv_head][max_ctx][head_dim]                      prompt pass only (one layer live)
//   V^T        bf16 [slot][kv_head][head_dim][vt_stride]                    prompt pass only (one layer live)
//   prompt pass activations are packed over all clips: row p = cu[clip] + position
//   decode activations are [batch row][features]
#include "engine.h"
#include "dec_sampler.h"
#include "dec_chain.h"
#include "dec_gemv_wide.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <cstdio>

namespace qasr {

__global__ void interleave_gate_up_kernel(const bf16_t* __restrict__ gate, const bf16_t* __restrict__ up,
                                          bf16_t* __restrict__ dst, int inter, int H) {
    // dst rows in blocks of 32: 16 gate rows j0..j0+15 then the 16 up rows j0..j0+15
    const int r = blockIdx.x;                      // 0 .. 2*inter-1
    const int blk = r >> 5, w = r & 31;
    const bf16_t* src = (w < 16 ? gate : up) + (long)(blk * 16 + (w & 15)) * H;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst + (long)r * H);
    for (int i = threadIdx.x; i < H / 8; i += blockDim.x) d4[i] = s4[i];
}

// dst rows in blocks of 32: 16 rows of `a` then the 16 matching rows of `b` (byte rows: packed q, scales, biases)
__global__ void interleave_rows_bytes_kernel(const char* __restrict__ a, const char* __restrict__ b, char* __restrict__ dst,
                                             int row_bytes) {
    const int r = blockIdx.x, blk = r >> 5, w = r & 31;
    const char* src = (w < 16 ? a : b) + (long)(blk * 16 + (w & 15)) * row_bytes;
    char* d = dst + (long)r * row_bytes;
    for (int i = threadIdx.x; i < row_bytes; i += blockDim.x) d[i] = src[i];
}

__global__ void narrow_f32_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = f32_to_bf16(src[i]);
}

__global__ void add_scalar_kernel(int* p, int n, int v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += v;
}

// teacher-forced path (no greedy_finalize): copy the rope rows of the current positions
__global__ void refresh_rope_rows_kernel(const int* __restrict__ ctx_len, RopeRows rr) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < rr.half) {
        const int p = ctx_len[b];
        rr.cos_rows[(long)b * rr.half + t] = rr.cos_table[(long)p * rr.half + t];
        rr.sin_rows[(long)b * rr.half + t] = rr.sin_table[(long)p * rr.half + t];
    }
}

// decode-step weights are streamed as MFMA A fragments: keep a fragment-major copy (1.19 GB extra for 0.6B)
const bf16_t* Engine::packed_copy(const bf16_t* w, int N, int K) {
    if (N % 16 != 0 || K % 32 != 0) return nullptr;
    auto buf = std::make_unique<DevBuf>();
    buf->alloc((size_t)N * K * sizeof(bf16_t));
    pack_mfma_a_launch(w, buf->as<bf16_t>(), N, K, stream_);
    const bf16_t* p = buf->as<bf16_t>();
    fused_.push_back(std::move(buf));
    return p;
}

// A quantised Linear / embedding of the checkpoint: `stem.weight` uint32 [N][K * bits / 32], `stem.scales`, `stem.biases`
// [N][K / 64] in bf16 or f32 (Sources/MLXCommon/WeightLoading.swift:48-96; bits from the preset, Qwen3ASR.swift:581-601)
QuantRaw Engine::quant_raw(const std::string& stem, int N, int K) const {
    const int bits = cfg_.bits;
    if (bits != 4 && bits != 8) throw std::runtime_error("quantised tensors need qasr_config.bits = 4 or 8 (got " + std::to_string(bits) + ")");
    if (cfg_.group_size != 64 || K % 64 != 0) throw std::runtime_error(stem + ": only group size 64 is supported");
    const Tensor &w = tensor(stem + ".weight"), &sc = tensor(stem + ".scales"), &bi = tensor(stem + ".biases");
    if (w.dtype != QASR_DTYPE_U32 || w.shape != std::vector<int64_t>{N, (int64_t)K * bits / 32})
        throw std::runtime_error(stem + ".weight: expected uint32 [" + std::to_string(N) + ", " + std::to_string(K * bits / 32) + "] (bits / shape mismatch)");
    for (const Tensor* t : {&sc, &bi}) {
        if (t->shape != std::vector<int64_t>{N, (int64_t)K / 64}) throw std::runtime_error(stem + ": scales / biases must be [out, in / 64]");
        if (t->dtype != QASR_DTYPE_BF16 && t->dtype != QASR_DTYPE_F32) throw std::runtime_error(stem + ": scales / biases must be bf16 or f32");
    }
    if (sc.dtype != bi.dtype) throw std::runtime_error(stem + ": scales and biases differ in dtype");
    QuantRaw q;
    q.wq = w.buf.as<uint32_t>(); q.scales = sc.buf.p; q.biases = bi.buf.p;
    q.sb_f32 = sc.dtype == QASR_DTYPE_F32; q.N = N; q.K = K; q.bits = bits;
    return q;
}

// decode-step images of a quantised matrix (dec_quant.h); shapes the tuned kernels cannot take keep only the triplet
QuantImg Engine::quant_image(const QuantRaw& raw) {
    QuantImg img;
    img.raw = raw; img.bits = raw.bits; img.sb_f32 = raw.sb_f32;
    if (raw.N % 16 != 0 || raw.K % 128 != 0) return img;
    auto qp = std::make_unique<DevBuf>(), sb = std::make_unique<DevBuf>();
    qp->alloc(quant_q_bytes(raw.N, raw.K, raw.bits));
    sb->alloc(quant_sb_bytes(raw.N, raw.K, raw.sb_f32));
    quant_pack_launch(raw, qp->as<uint32_t>(), sb->p, stream_);
    img.qp = qp->as<uint32_t>(); img.sb = sb->p;
    fused_.push_back(std::move(qp));
    fused_.push_back(std::move(sb));
    return img;
}

void Engine::embed_rows(const int* d_ids, bf16_t* dst, int n, hipStream_t s) {
    if (decw_.quant) gather_rows_q_launch(decw_.embed_raw, d_ids, dst, n, s);
    else gather_rows_launch(decw_.embed, d_ids, dst, n, cfg_.hidden, s);
}

void Engine::finalize_decoder() {
    const int H = cfg_.hidden, hd = cfg_.head_dim, nq = cfg_.heads * hd, nkv = cfg_.kv_heads * hd, I = cfg_.inter;
    if (cfg_.inter % 16 != 0 || H % 32 != 0) throw std::invalid_argument("decoder widths must be multiples of 16/32");
    // an aligner engine runs the prompt pass only: no decode-step weight images, one K cache shared by all layers
    const bool aligner = cfg_.classify_num > 0;
    // MLX-quantised checkpoint (QuantizedTextModel: every decoder Linear + the tied embedding are triplets) or float
    decw_ = DecW{};
    decw_.quant = tensors_.count("model.embed_tokens.scales") > 0;
    decw_.norm = wptr("model.norm.weight", {H});
    if (decw_.quant) {
        decw_.embed_raw = quant_raw("model.embed_tokens", cfg_.vocab, H);
        if (!aligner) decw_.embed_q = quant_image(decw_.embed_raw);
    } else {
        decw_.embed = wptr("model.embed_tokens.weight", {cfg_.vocab, H});
        decw_.embed_p = aligner ? nullptr : packed_copy(decw_.embed, cfg_.vocab, H);
    }
    if (aligner) {
        if (cfg_.classify_num % 4 != 0) throw std::invalid_argument("classify_num must be a multiple of 4");
        decw_.cls_w = wptr("lm_head.weight", {cfg_.classify_num, H});
        decw_.cls_b = wptr("lm_head.bias", {cfg_.classify_num});
    }
    decw_.layers.clear();
    for (int i = 0; i < cfg_.dec_layers; ++i) {
        const std::string p = "model.layers." + std::to_string(i) + ".";
        DecLayerW L{};
        L.ln1 = wptr(p + "input_layernorm.weight", {H});
        L.ln2 = wptr(p + "post_attention_layernorm.weight", {H});
        L.qn = wptr(p + "self_attn.q_norm.weight", {hd});
        L.kn = wptr(p + "self_attn.k_norm.weight", {hd});
        if (decw_.quant) {
            // The prompt pass multiplies by bf16(scale * q + bias) like the reference's many-row kernel (dec_quant.h), so
            // it keeps its bf16 GEMMs on dequantised copies; the decode step reads the packed images.  q|k|v and gate|up
            // are fused exactly like the float path: rows concatenated / interleaved in blocks of 16, triplet by triplet.
            const int bits = cfg_.bits;
            const QuantRaw rq = quant_raw(p + "self_attn.q_proj", nq, H), rk = quant_raw(p + "self_attn.k_proj", nkv, H),
                           rv = quant_raw(p + "self_attn.v_proj", nkv, H), ro = quant_raw(p + "self_attn.o_proj", H, nq),
                           rg = quant_raw(p + "mlp.gate_proj", I, H), ru = quant_raw(p + "mlp.up_proj", I, H),
                           rd = quant_raw(p + "mlp.down_proj", H, I);
            L.rq = rq; L.rk = rk; L.rv = rv; L.ro = ro; L.rg = rg; L.ru = ru; L.rd = rd;
            // q|k|v rows concatenated and gate|up interleaved triplet by triplet (temporaries: only the decode-step images built from
            // them stay resident, unless a shape has no tuned image and the generic kernel needs the fused triplet itself)
            std::vector<std::unique_ptr<DevBuf>> tmp;
            auto tmp_buf = [&](size_t bytes) {
                tmp.push_back(std::make_unique<DevBuf>());
                tmp.back()->alloc(bytes);
                return tmp.back()->p;
            };
            if (!aligner) {
                const size_t esz = rq.sb_f32 ? 4 : 2;
                auto cat3 = [&](const void* a, size_t na, const void* b, size_t nb, const void* c, size_t nc) {
                    char* d = (char*)tmp_buf(na + nb + nc);
                    QASR_HIP(hipMemcpyAsync(d, a, na, hipMemcpyDeviceToDevice, stream_));
                    QASR_HIP(hipMemcpyAsync(d + na, b, nb, hipMemcpyDeviceToDevice, stream_));
                    QASR_HIP(hipMemcpyAsync(d + na + nb, c, nc, hipMemcpyDeviceToDevice, stream_));
                    return (void*)d;
                };
                QuantRaw rqkv = rq;
                rqkv.N = nq + 2 * nkv;
                const size_t wrow = (size_t)H * bits / 8, srow = (size_t)(H / 64) * esz;
                rqkv.wq = (const uint32_t*)cat3(rq.wq, nq * wrow, rk.wq, nkv * wrow, rv.wq, nkv * wrow);
                rqkv.scales = cat3(rq.scales, nq * srow, rk.scales, nkv * srow, rv.scales, nkv * srow);
                rqkv.biases = cat3(rq.biases, nq * srow, rk.biases, nkv * srow, rv.biases, nkv * srow);
                QuantRaw rgu = rg;
                rgu.N = 2 * I;
                auto inter2 = [&](const void* a, const void* b, size_t row_bytes) {
                    char* d = (char*)tmp_buf((size_t)2 * I * row_bytes);
                    hipLaunchKernelGGL(interleave_rows_bytes_kernel, dim3(2 * I), dim3(128), 0, stream_, (const char*)a, (const char*)b, d, (int)row_bytes);
                    return (void*)d;
                };
                rgu.wq = (const uint32_t*)inter2(rg.wq, ru.wq, wrow);
                rgu.scales = inter2(rg.scales, ru.scales, srow);
                rgu.biases = inter2(rg.biases, ru.biases, srow);
                L.qkv_q = quant_image(rqkv); L.o_q = quant_image(ro); L.gu_q = quant_image(rgu); L.down_q = quant_image(rd);
                if (!L.qkv_q.qp || !L.gu_q.qp) for (auto& b : tmp) fused_.push_back(std::move(b));   // generic kernel reads img.raw
                else QASR_HIP(hipStreamSynchronize(stream_));                                           // images built: drop the temporaries
            }
            decw_.layers.push_back(L);
            continue;
        }
        L.wo = wptr(p + "self_attn.o_proj.weight", {H, nq});
        L.wdown = wptr(p + "mlp.down_proj.weight", {H, I});
        const bf16_t* wq = wptr(p + "self_attn.q_proj.weight", {nq, H});
        const bf16_t* wk = wptr(p + "self_attn.k_proj.weight", {nkv, H});
        const bf16_t* wv = wptr(p + "self_attn.v_proj.weight", {nkv, H});
        auto qkv = std::make_unique<DevBuf>();
        qkv->alloc((size_t)(nq + 2 * nkv) * H * sizeof(bf16_t));
        QASR_HIP(hipMemcpyAsync(qkv->as<bf16_t>(), wq, (size_t)nq * H * 2, hipMemcpyDeviceToDevice, stream_));
        QASR_HIP(hipMemcpyAsync(qkv->as<bf16_t>() + (size_t)nq * H, wk, (size_t)nkv * H * 2, hipMemcpyDeviceToDevice, stream_));
        QASR_HIP(hipMemcpyAsync(qkv->as<bf16_t>() + (size_t)(nq + nkv) * H, wv, (size_t)nkv * H * 2, hipMemcpyDeviceToDevice, stream_));
        L.wqkv = qkv->as<bf16_t>();
        fused_.push_back(std::move(qkv));
        const bf16_t* wg = wptr(p + "mlp.gate_proj.weight", {I, H});
        const bf16_t* wu = wptr(p + "mlp.up_proj.weight", {I, H});
        auto gu = std::make_unique<DevBuf>();
        gu->alloc((size_t)2 * I * H * sizeof(bf16_t));
        hipLaunchKernelGGL(interleave_gate_up_kernel, dim3(2 * I), dim3(128), 0, stream_, wg, wu, gu->as<bf16_t>(), I, H);
        L.wgu = gu->as<bf16_t>();
        fused_.push_back(std::move(gu));
        L.wqkv_p = aligner ? nullptr : packed_copy(L.wqkv, nq + 2 * nkv, H);
        L.wo_p = aligner ? nullptr : packed_copy(L.wo, H, nq);
        L.wgu_p = aligner ? nullptr : packed_copy(L.wgu, 2 * I, H);
        L.wdown_p = aligner ? nullptr : packed_copy(L.wdown, H, I);
        decw_.layers.push_back(L);
    }
    if (decw_.quant) d_wscratch_.alloc(((size_t)(nq + 2 * nkv) * H + (size_t)H * nq + (size_t)2 * I * H + (size_t)H * I) * sizeof(bf16_t));
    // capacity: prompt = 16 fixed ids + audio tokens + context/language extras (Qwen3ASR.swift:199-233)
    const int max_audio_tok = num_audio_tokens(mel_num_frames(max_samples_));
    max_prompt_ = 16 + max_audio_tok + cfg_.max_prompt_extra;
    max_ctx_ = ((max_prompt_ + cfg_.max_new_tokens + 63) / 64) * 64;
    vt_stride_ = ((max_prompt_ + 63) / 64) * 64;
    max_pos_ = cfg_.max_batch * max_prompt_;
    // RoPE tables: theta_i = base^(-i/half), f32 like MLXNN.RoPE(traditional: false)
    {
        const int half = hd / 2;
        std::vector<float> c((size_t)max_ctx_ * half), sn((size_t)max_ctx_ * half);
        const float k = (float)(-std::log((double)cfg_.rope_theta) / (double)half);
        for (int i = 0; i < half; ++i) {
            const float inv = expf((float)i * k);
            for (int p = 0; p < max_ctx_; ++p) {
                const float ang = (float)p * inv;
                c[(size_t)p * half + i] = cosf(ang);
                sn[(size_t)p * half + i] = sinf(ang);
            }
        }
        d_rope_cos_.alloc(c.size() * sizeof(float));
        d_rope_sin_.alloc(sn.size() * sizeof(float));
        QASR_HIP(hipMemcpyAsync(d_rope_cos_.p, c.data(), c.size() * sizeof(float), hipMemcpyHostToDevice, stream_));
        QASR_HIP(hipMemcpyAsync(d_rope_sin_.p, sn.data(), sn.size() * sizeof(float), hipMemcpyHostToDevice, stream_));
        QASR_HIP(hipStreamSynchronize(stream_));
    }
    const int B = cfg_.max_batch, nh = cfg_.heads + 2 * cfg_.kv_heads;
    kcache_.clear();
    vfcache_.clear();
    const size_t cache_bytes = (size_t)B * cfg_.kv_heads * max_ctx_ * hd * sizeof(bf16_t);
    for (int i = 0; i < (aligner ? 1 : cfg_.dec_layers); ++i) {
        kcache_.push_back(std::make_unique<DevBuf>());
        kcache_.back()->alloc(cache_bytes);
        if (aligner) break;
        vfcache_.push_back(std::make_unique<DevBuf>());
        vfcache_.back()->alloc(cache_bytes);
    }
    d_rope_rows_.alloc((size_t)2 * B * (hd / 2) * sizeof(float));
    QASR_HIP(hipMemsetAsync(d_rope_rows_.p, 0, d_rope_rows_.bytes, stream_));
    d_vt_.alloc((size_t)B * cfg_.kv_heads * hd * vt_stride_ * sizeof(bf16_t));
    QASR_HIP(hipMemsetAsync(d_vt_.p, 0, d_vt_.bytes, stream_));   // masked keys multiply stale bytes by P = 0
    d_px_.alloc((size_t)max_pos_ * H * 2);
    d_ph_.alloc((size_t)max_pos_ * H * 2);
    d_pqkv_.alloc((size_t)max_pos_ * nh * hd * 2);
    d_pqr_.alloc((size_t)max_pos_ * nq * 2);
    d_pattn_.alloc((size_t)max_pos_ * nq * 2);
    d_pact_.alloc((size_t)max_pos_ * I * 2);
    d_dx_.alloc((size_t)B * H * 2);
    d_dh_.alloc((size_t)B * H * 2);
    d_dqkv_.alloc((size_t)B * nh * hd * 2);
    d_dattn_.alloc((size_t)B * nq * 2);
    d_dact_.alloc((size_t)B * I * 2);
    d_chain_ctr_.alloc(CHAIN_STATE_BYTES);
    QASR_HIP(hipMemsetAsync(d_chain_ctr_.p, 0, CHAIN_STATE_BYTES, stream_));
    d_qa_gran_.alloc(QA_GRAN_BYTES);
    QASR_HIP(hipMemsetAsync(d_qa_gran_.p, 0, QA_GRAN_BYTES, stream_));
    d_qa_part_.alloc(QA_PART_BYTES);
    QASR_HIP(hipMemsetAsync(d_qa_part_.p, 0, QA_PART_BYTES, stream_));
    d_logits_.alloc((size_t)B * cfg_.vocab * sizeof(float));
    n_parts_ = decw_.quant ? lm_head_q_parts(cfg_.vocab, H, cfg_.bits) : lm_head_parts(cfg_.vocab, H);
    const int parts_cap = std::max(n_parts_, decode_gemv_blocks(DEC_EPI_LOGITS, cfg_.vocab));
    d_part_val_.alloc((size_t)B * parts_cap * sizeof(float));
    d_part_idx_.alloc((size_t)B * parts_cap * sizeof(int));
    const size_t pmeta = (size_t)max_pos_ * 4 * sizeof(int) + (size_t)(3 * B + 2) * sizeof(int);
    h_pmeta_.alloc(pmeta);
    d_pmeta_.alloc(pmeta);
    // greedy state: tokens [B][max_new+1] | lens [B] | finished [B] | ctx_len [B] | n_active [1]
    const size_t gs = ((size_t)B * (cfg_.max_new_tokens + 1) + 3 * B + 4) * sizeof(int);   // + n_active, err, 2 spare
    d_gstate_.alloc(gs);
    int* g = d_gstate_.as<int>();
    gstate_.tokens = g;
    gstate_.lens = g + (size_t)B * (cfg_.max_new_tokens + 1);
    gstate_.finished = gstate_.lens + B;
    gstate_.ctx_len = gstate_.finished + B;
    gstate_.n_active = gstate_.ctx_len + B;
    gstate_.err = gstate_.n_active + 1;
    d_err_flag_ = gstate_.err;
    gstate_.max_new = cfg_.max_new_tokens;
    gstate_.eos = cfg_.tok_im_end;
    gstate_.vocab = cfg_.vocab;
    gstate_.clear = d_chain_ctr_.as<unsigned>();
    gstate_.clear_words = (int)(CHAIN_CTR_BYTES / sizeof(unsigned));
    for (auto& e : ev_)
        if (!e) QASR_HIP(hipEventCreate(&e));
    QASR_HIP(hipStreamSynchronize(stream_));
}

// ---- prompt planning (integer work, R7) ----------------------------------------------------------
void Engine::plan_prefill(const qasr_options* opt, const std::vector<int>& n_audio,
                          const std::vector<std::vector<int32_t>>* aligner_tails) {
    const int B = (int)n_audio.size();
    if (opt && (opt->n_context < 0 || opt->n_language < 0)) throw std::invalid_argument("negative context / language id count");
    const int n_ctx = opt && opt->context_ids ? opt->n_context : 0;
    const int n_lang = opt && opt->language_ids ? opt->n_language : 0;
    if (n_ctx + n_lang > cfg_.max_prompt_extra) throw std::length_error("context + language ids exceed max_prompt_extra");
    // every id becomes a row index of the embedding gather (embed_splice_kernel): refuse what is not a vocabulary row
    auto check_ids = [&](const int32_t* ids, int n, const char* what) {
        for (int i = 0; i < n; ++i)
            if (ids[i] < 0 || ids[i] >= cfg_.vocab)
                throw std::invalid_argument(std::string(what) + " id " + std::to_string(ids[i]) + " outside [0, vocab)");
    };
    if (n_ctx) check_ids(opt->context_ids, n_ctx, "context");
    if (n_lang) check_ids(opt->language_ids, n_lang, "language");
    if (aligner_tails)
        for (auto& t : *aligner_tails) check_ids(t.data(), (int)t.size(), "slotted text");
    if (aligner_tails) {
        if ((int)aligner_tails->size() != B) throw std::invalid_argument("aligner: one slotted text per clip");
        for (auto& t : *aligner_tails)
            if ((int)t.size() > cfg_.max_prompt_extra) throw std::length_error("slotted text exceeds max_prompt_extra");
    }
    char* hp = h_pmeta_.as<char>();
    int* ids = reinterpret_cast<int*>(hp);
    int* asrc = ids + max_pos_;
    int* slot = asrc + max_pos_;
    int* pos = slot + max_pos_;
    int* cu = pos + max_pos_;
    int* slotclip = cu + (B + 1);
    int* last = slotclip + B;
    int p = 0;
    max_len_ = 0;
    prompt_len_.assign(B, 0);
    cu[0] = 0;
    for (int b = 0; b < B; ++b) {
        const int start = p;
        auto push = [&](int id, int a) { ids[p] = id; asrc[p] = a; slot[p] = b; pos[p] = p - start; ++p; };
        // Qwen3ASR.swift:199-233
        push(cfg_.tok_im_start, -1); push(cfg_.tok_system, -1); push(cfg_.tok_newline, -1);
        for (int i = 0; i < n_ctx; ++i) push(opt->context_ids[i], -1);
        push(cfg_.tok_im_end, -1); push(cfg_.tok_newline, -1);
        push(cfg_.tok_im_start, -1); push(cfg_.tok_user, -1); push(cfg_.tok_newline, -1); push(cfg_.tok_audio_start, -1);
        for (int i = 0; i < n_audio[b]; ++i) push(cfg_.tok_audio_pad, clip_tok_off_[b] + i);
        push(cfg_.tok_audio_end, -1); push(cfg_.tok_im_end, -1); push(cfg_.tok_newline, -1);
        push(cfg_.tok_im_start, -1); push(cfg_.tok_assistant, -1); push(cfg_.tok_newline, -1);
        if (aligner_tails) {
            // ForcedAligner.swift:337-378: the same template, then the text with <timestamp> slots; no <asr_text>
            for (int32_t id : (*aligner_tails)[b]) push(id, -1);
        } else {
            for (int i = 0; i < n_lang; ++i) push(opt->language_ids[i], -1);
            push(cfg_.tok_asr_text, -1);
        }
        prompt_len_[b] = p - start;
        if (prompt_len_[b] > max_prompt_) throw std::length_error("prompt longer than engine capacity");
        max_len_ = std::max(max_len_, prompt_len_[b]);
        cu[b + 1] = p;
        slotclip[b] = b;
        last[b] = p - 1;
    }
    n_pos_ = p;
    // one upload: the five per-position arrays are not contiguous (max_pos_ stride) -> copy whole block
    const size_t bytes = (size_t)max_pos_ * 4 * sizeof(int) + (size_t)(3 * B + 1) * sizeof(int);
    QASR_HIP(hipMemcpyAsync(d_pmeta_.p, h_pmeta_.p, bytes, hipMemcpyHostToDevice, stream_));
    int* dp = d_pmeta_.as<int>();
    d_p_ids_ = dp;
    d_p_audio_src_ = dp + max_pos_;
    d_p_slot_ = dp + 2 * (size_t)max_pos_;
    d_p_pos_ = dp + 3 * (size_t)max_pos_;
    d_p_cu_ = dp + 4 * (size_t)max_pos_;
    d_p_slotclip_ = d_p_cu_ + (B + 1);
    d_p_last_ = d_p_slotclip_ + B;
    h_ctx0_ = prompt_len_;
}

void Engine::reset_greedy_state(int max_tokens, bool ignore_eos) {
    const int B = batch_;
    cur_max_tokens_ = max_tokens;
    cur_ignore_eos_ = ignore_eos;
    gstate_.max_tokens = max_tokens;
    gstate_.ignore_eos = ignore_eos ? 1 : 0;
    // tokens = -1, lens = finished = 0, ctx_len = prompt_len, n_active = B
    QASR_HIP(hipMemsetAsync(gstate_.tokens, 0xff, (size_t)cfg_.max_batch * (cfg_.max_new_tokens + 1) * sizeof(int), stream_));
    QASR_HIP(hipMemsetAsync(gstate_.lens, 0, (size_t)2 * cfg_.max_batch * sizeof(int), stream_));
    // ctx_len [max_batch] and n_active are contiguous in the state block: one copy from pinned memory
    if (!h_ginit_.p) h_ginit_.alloc((size_t)(cfg_.max_batch + 2) * sizeof(int));
    int* init = h_ginit_.as<int>();
    for (int b = 0; b < cfg_.max_batch; ++b) init[b] = b < B ? h_ctx0_[b] : 0;
    init[cfg_.max_batch] = B;
    init[cfg_.max_batch + 1] = 0;                                  // err flag
    QASR_HIP(hipMemcpyAsync(gstate_.ctx_len, init, (size_t)(cfg_.max_batch + 2) * sizeof(int), hipMemcpyHostToDevice, stream_));
    steps_done_ = 0;
    forced_ctx_ = B > 0 ? h_ctx0_[0] : 0;
}

void Engine::run_lm_head(bool want_logits, int r0, int nr, hipStream_t s) {
    const int H = cfg_.hidden;
    if (decw_.quant) {
        // the generic (untuned-shape) quantised head writes every logit and reduces them afterwards: it always needs the buffer
        const bool need = want_logits || n_parts_ == 1;
        lm_head_q_launch(decw_.embed_q, d_dx_.as<bf16_t>() + (size_t)r0 * H, decw_.norm, cfg_.rms_eps, nr, cfg_.vocab, H,
                         need ? d_logits_.as<float>() + (size_t)r0 * cfg_.vocab : nullptr,
                         d_part_val_.as<float>() + (size_t)r0 * n_parts_, d_part_idx_.as<int>() + (size_t)r0 * n_parts_,
                         d_dh_.as<bf16_t>() + (size_t)r0 * H, s);
        return;
    }
    lm_head_launch(decw_.embed, decw_.embed_p, d_dx_.as<bf16_t>() + (size_t)r0 * H, decw_.norm, cfg_.rms_eps, nr, cfg_.vocab, H,
                   want_logits ? d_logits_.as<float>() + (size_t)r0 * cfg_.vocab : nullptr,
                   d_part_val_.as<float>() + (size_t)r0 * n_parts_, d_part_idx_.as<int>() + (size_t)r0 * n_parts_,
                   d_dh_.as<bf16_t>() + (size_t)r0 * H, s);
}

Engine::PromptW Engine::prompt_weights(int l, hipStream_t s) {
    const DecLayerW& L = decw_.layers[l];
    if (!decw_.quant) return {L.wqkv, L.wo, L.wgu, L.wdown};
    const int H = cfg_.hidden, hd = cfg_.head_dim, nq = cfg_.heads * hd, nkv = cfg_.kv_heads * hd, I = cfg_.inter;
    bf16_t* wqkv = d_wscratch_.as<bf16_t>();
    bf16_t* wo = wqkv + (size_t)(nq + 2 * nkv) * H;
    bf16_t* wgu = wo + (size_t)H * nq;
    bf16_t* wdown = wgu + (size_t)2 * I * H;
    const DequantJob jobs[7] = {{L.rq, 0, nq, wqkv},
                                {L.rk, 0, nkv, wqkv + (size_t)nq * H},
                                {L.rv, 0, nkv, wqkv + (size_t)(nq + nkv) * H},
                                {L.ro, 0, H, wo},
                                {L.rg, 0, I, wgu, 16, 32, 0},       // gate | up in 32-row blocks: 16 gate rows + the 16 matching up rows
                                {L.ru, 0, I, wgu, 16, 32, 16},
                                {L.rd, 0, H, wdown}};
    quant_dequant_multi_launch(jobs, 7, s);                          // one launch per layer
    return {wqkv, wo, wgu, wdown};
}

void Engine::run_prefill(bool want_logits) {
    const int H = cfg_.hidden, hd = cfg_.head_dim, nq = cfg_.heads * hd, nh = cfg_.heads + 2 * cfg_.kv_heads, I = cfg_.inter;
    hipStream_t s = stream_;
    bf16_t *x = d_px_.as<bf16_t>(), *h = d_ph_.as<bf16_t>(), *qkv = d_pqkv_.as<bf16_t>(), *qr = d_pqr_.as<bf16_t>();
    bf16_t *at = d_pattn_.as<bf16_t>(), *act = d_pact_.as<bf16_t>();
    const int P = n_pos_;
    if (decw_.quant) embed_splice_q_launch(d_p_ids_, d_p_audio_src_, decw_.embed_raw, d_audio_.as<bf16_t>(), x, P, H, s);
    else embed_splice_launch(d_p_ids_, d_p_audio_src_, decw_.embed, d_audio_.as<bf16_t>(), x, P, H, s);
    for (int l = 0; l < cfg_.dec_layers; ++l) {
        const DecLayerW& L = decw_.layers[l];
        const bool aligner = cfg_.classify_num > 0;
        KVLayout kv{kcache_[aligner ? 0 : l]->as<bf16_t>(), nullptr, max_ctx_, cfg_.kv_heads, hd,
                    aligner ? nullptr : vfcache_[l]->as<bf16_t>()};
        const PromptW W = prompt_weights(l, s);
        rmsnorm_rows_launch(x, L.ln1, h, P, H, cfg_.rms_eps, s);
        const bool fuse_qk = qk_norm_rope_fusable(cfg_.heads, cfg_.kv_heads, hd);
        if (fuse_qk)     // q/k RMSNorm + RoPE + cache write in the projection's epilogue (head tiles), same bits as the separate launch
            gemm_nt_headtiles(ADense{h, H, P, H}, W.wqkv, H, P, nh * hd, H,
                              EpiQkHeads{qkv, (long)nh * hd, qr, kv, d_p_slot_, d_p_pos_, L.qn, L.kn, cfg_.rms_eps, d_rope_cos_.as<float>(),
                                         d_rope_sin_.as<float>(), cfg_.heads, cfg_.kv_heads}, s);
        else
            gemm_nt(ADense{h, H, P, H}, W.wqkv, H, P, nh * hd, H, EpiStoreBf16{qkv, (long)nh * hd}, s);
        qk_norm_rope_launch(qkv, d_p_slot_, d_p_pos_, P, cfg_.heads, cfg_.kv_heads, hd, L.qn, L.kn, cfg_.rms_eps,
                            d_rope_cos_.as<float>(), d_rope_sin_.as<float>(), qr, kv, d_vt_.as<bf16_t>(), vt_stride_,
                            d_p_cu_, d_p_slotclip_, batch_, max_len_, s, fuse_qk);
        prefill_attention_launch(qr, kv, d_vt_.as<bf16_t>(), vt_stride_, d_p_cu_, d_p_slotclip_, batch_, max_len_,
                                 cfg_.heads, at, s);
        gemm_nt(ADense{at, nq, P, nq}, W.wo, nq, P, H, nq, EpiResidBf16{x, H}, s);
        rmsnorm_rows_launch(x, L.ln2, h, P, H, cfg_.rms_eps, s);
        gemm_nt_swiglu(ADense{h, H, P, H}, W.wgu, H, P, 2 * I, H, EpiStoreBf16{act, I}, s);
        gemm_nt(ADense{act, I, P, I}, W.wdown, I, P, H, I, EpiResidBf16{x, H}, s);
    }
    if (cfg_.classify_num > 0) { QASR_HIP(hipGetLastError()); return; }      // aligner: the caller reads d_px_ rows
    // last position of every clip -> decode rows (Qwen3ASR.swift:254-256)
    gather_rows_launch(x, d_p_last_, d_dx_.as<bf16_t>(), batch_, H, s);
    run_lm_head(want_logits, 0, batch_, s);
    QASR_HIP(hipGetLastError());
}

// Forced aligner forward (ForcedAligner.swift:236-299) for a batch of clips: mel -> encoder -> one decoder pass over
// [template + audio + slotted text] -> final RMSNorm + Linear(hidden, classify_num) at the timestamp slots -> argmax.
// raw[b] receives one class index per slot of clip b; logits (optional) [sum n_ts, classify_num] f32.
void Engine::align_forward(const float* const* pcm, const size_t* n, size_t B,
                           const std::vector<std::vector<int32_t>>& slotted,
                           const std::vector<std::vector<int32_t>>& ts_pos, std::vector<std::vector<int32_t>>& raw,
                           float* logits) {
    if (!finalized_) throw std::runtime_error("weights not finalized");
    if (cfg_.classify_num <= 0) throw std::runtime_error("this engine has no timestamp head (create it from an aligner preset)");
    if (B == 0 || slotted.size() != B || ts_pos.size() != B) throw std::invalid_argument("align: one slotted text per clip");
    const int H = cfg_.hidden, C = cfg_.classify_num;
    upload_pcm(pcm, n, B);
    plan_encoder();
    std::vector<int> n_audio;
    for (auto& c : clips_) n_audio.push_back(c.n_tokens);
    plan_prefill(nullptr, n_audio, &slotted);
    // packed row of slot j of clip b: cu[b] + 9 template ids + audio + 6 template ids + position inside the text
    std::vector<int> rows;
    int p0 = 0;
    for (size_t b = 0; b < B; ++b) {
        const int text0 = p0 + 9 + n_audio[b] + 6;
        for (int32_t q : ts_pos[b]) {
            if (q < 0 || q >= (int)slotted[b].size()) throw std::invalid_argument("align: timestamp position outside the text");
            rows.push_back(text0 + q);
        }
        p0 += prompt_len_[b];
    }
    const int R = (int)rows.size();
    if (R == 0) { raw.assign(B, {}); return; }
    if (R > max_pos_) throw std::length_error("align: more timestamp slots than prompt positions");
    hipStream_t s = stream_;
    if (d_al_rows_.bytes < (size_t)R * sizeof(int)) d_al_rows_.alloc((size_t)R * sizeof(int));
    if (d_al_x_.bytes < (size_t)R * H * 2 * 2) d_al_x_.alloc((size_t)R * H * 2 * 2);
    if (d_al_logits_.bytes < (size_t)R * C * 2) d_al_logits_.alloc((size_t)R * C * 2);
    if (d_al_idx_.bytes < (size_t)R * sizeof(int)) d_al_idx_.alloc((size_t)R * sizeof(int));
    QASR_HIP(hipMemcpyAsync(d_al_rows_.p, rows.data(), (size_t)R * sizeof(int), hipMemcpyHostToDevice, s));
    QASR_HIP(hipEventRecord(ev_[0], s));
    run_mel();
    QASR_HIP(hipEventRecord(ev_[1], s));
    if (ev_mel_done_) QASR_HIP(hipEventRecord(ev_mel_done_, s));       // the device PCM / meta buffers are free for the next staged batch
    run_issued_ = true;
    run_encoder();
    QASR_HIP(hipEventRecord(ev_[2], s));
    run_prefill(false);
    bf16_t* xr = d_al_x_.as<bf16_t>();
    bf16_t* xn = xr + (size_t)R * H;
    gather_rows_launch(d_px_.as<bf16_t>(), d_al_rows_.as<int>(), xr, R, H, s);
    rmsnorm_rows_launch(xr, decw_.norm, xn, R, H, cfg_.rms_eps, s);
    // MLX Linear = addmm(bias, x, W^T): one rounding of acc + bias to the decoder dtype (ForcedAligner.swift:283)
    gemm_nt(ADense{xn, H, R, H}, decw_.cls_w, H, R, C, H, EpiBiasActBf16<0>{d_al_logits_.as<bf16_t>(), (long)C, decw_.cls_b}, s);
    argmax_rows_launch(d_al_logits_.as<bf16_t>(), C, R, C, d_al_idx_.as<int>(), s);
    QASR_HIP(hipEventRecord(ev_[3], s));
    QASR_HIP(hipEventRecord(ev_[4], s));
    std::vector<int> idx(R);
    QASR_HIP(hipMemcpyAsync(idx.data(), d_al_idx_.p, (size_t)R * sizeof(int), hipMemcpyDeviceToHost, s));
    std::vector<bf16_t> lg;
    if (logits) {
        lg.resize((size_t)R * C);
        QASR_HIP(hipMemcpyAsync(lg.data(), d_al_logits_.p, lg.size() * 2, hipMemcpyDeviceToHost, s));
    }
    QASR_HIP(hipStreamSynchronize(s));
    QASR_HIP(hipGetLastError());
    if (logits)
        for (size_t i = 0; i < lg.size(); ++i) logits[i] = bf16_to_f32_host(lg[i]);
    raw.assign(B, {});
    int k = 0;
    for (size_t b = 0; b < B; ++b)
        for (size_t j = 0; j < ts_pos[b].size(); ++j) raw[b].push_back(idx[k++]);
}

RopeRows Engine::rope_rows(int r0) const {
    const int half = cfg_.head_dim / 2;
    return RopeRows{d_rope_cos_.as<float>(), d_rope_sin_.as<float>(), d_rope_rows_.as<float>() + (size_t)r0 * half,
                    d_rope_rows_.as<float>() + (size_t)(cfg_.max_batch + r0) * half, half};
}

GreedyState Engine::greedy_rows(int r0) const {
    GreedyState g = gstate_;
    g.tokens += (size_t)r0 * (cfg_.max_new_tokens + 1);
    g.lens += r0;
    g.finished += r0;
    g.ctx_len += r0;
    return g;
}

// The one dispatch of a decode-step linear (run_decode_step and kernel_probe both go through it, so the probe times what ships).
// float checkpoint: fragment-major bf16 images; quantised checkpoint: packed 4 / 8-bit images (dec_quant.h)
void Engine::decode_gemv(DecEpi epi, const DecGemvArgs& a, const QuantImg& qi, const bf16_t* norm_w, bf16_t* h, hipStream_t s) {
    if (decw_.quant) decode_gemv_q_launch(epi, a, qi, norm_w, cfg_.rms_eps, h, s);
    else if (!norm_w && tuning().gemv_wide && decode_gemv_wide_supported(epi, a)) decode_gemv_wide_launch(epi, a, s);   // K = 6144 (1.7B)
    else decode_gemv_fused_launch(epi, a, norm_w, norm_w ? cfg_.rms_eps : 0.f, norm_w ? h : nullptr, s);
}

int Engine::step_chain(int r0, int nr) const {
    const int hd = cfg_.head_dim;
    return (!decw_.quant && r0 == 0 && !stamp_buf_ && !shared_device_ &&
            decode_chain_supported(cfg_.hidden, cfg_.heads * hd, cfg_.inter, (cfg_.heads + 2 * cfg_.kv_heads) * hd, nr)) ? tuning().chain : 0;
}
bool Engine::step_qa(int r0, int nr, int chain) const {
    const int hd = cfg_.head_dim;
    if (decw_.quant) {      // quantised checkpoints: the packed image with bf16 scales, granule hand-off
        const QuantImg& q = decw_.layers[0].qkv_q;
        if (!q.qp || !q.sb || q.sb_f32 || (q.bits != 4 && q.bits != 8) || !tuning().qa_gran) return false;
    }
    return r0 == 0 && !stamp_buf_ && !shared_device_ && chain < 3 && tuning().qa &&
           decode_qa_supported(cfg_.hidden, cfg_.heads, cfg_.kv_heads, hd, nr, max_ctx_) && (cfg_.heads + 2 * cfg_.kv_heads) * hd == 4096;
}
void Engine::decode_structure(int* fused_qa, int* chain, int* launches_per_layer) {
    if (!finalized_ || batch_ <= 0) throw std::runtime_error("decode_structure needs a prepared batch");
    const int rows = decode_group_rows();
    const bool whole = rows == batch_;
    const int c = whole ? step_chain(0, rows) : 0;
    const bool q = whole && step_qa(0, rows, c);
    *chain = c;
    *fused_qa = q ? 1 : 0;
    // five launches: q|k|v, attention, o-proj, gate|up, down
    *launches_per_layer = c == 3 ? 2 : 5 - (q ? 1 : 0) - (c == 1 ? 1 : c == 2 ? 2 : 0);
}

// One decode step for batch rows [r0, r0 + nr) on stream s.  Rows are independent, so a step can be
// issued as several row groups on parallel graph branches (see decode_loop).
void Engine::run_decode_step(bool want_logits, bool greedy, int r0, int nr, hipStream_t s, bool with_head) {
    const int H = cfg_.hidden, hd = cfg_.head_dim, nq = cfg_.heads * hd, nh = cfg_.heads + 2 * cfg_.kv_heads, I = cfg_.inter;
    bf16_t* x = d_dx_.as<bf16_t>() + (size_t)r0 * H;
    bf16_t* h = d_dh_.as<bf16_t>() + (size_t)r0 * H;
    bf16_t* qkv = d_dqkv_.as<bf16_t>() + (size_t)r0 * nh * hd;
    bf16_t* at = d_dattn_.as<bf16_t>() + (size_t)r0 * nq;
    bf16_t* act = d_dact_.as<bf16_t>() + (size_t)r0 * I;
    const GreedyState gs = greedy_rows(r0);
    struct SharedScope { bool on; SharedScope(bool s) : on(s) { if (on) tuning_thread_shared(true); } ~SharedScope() { if (on) tuning_thread_shared(false); } } shared_scope(shared_device_);
    // chain > 0: a layer's linears run as one persistent launch (dec_chain.hip) wherever it has an instantiation: the float 0.6B geometry,
    // the whole batch in one row group (the arrival counters belong to one step of one engine), up to 32 rows
    const int chain = step_chain(r0, nr);
    const bool qa = step_qa(r0, nr, chain);
    // the fused launches' arrival counters start every step at zero: a greedy step is always preceded by a finalize launch (the previous step's,
    // or the one behind the prompt pass), which zeroes them; every other caller of a step (logits for the host, forced tokens, probes) zeroes here
    if ((chain || qa) && !(greedy && with_head)) decode_chain_reset(d_chain_ctr_.as<unsigned>(), s);
    for (int l = 0; l < cfg_.dec_layers; ++l) {
        const DecLayerW& L = decw_.layers[l];
        KVLayout kv{kcache_[l]->as<bf16_t>(), nullptr, max_ctx_, cfg_.kv_heads, hd, vfcache_[l]->as<bf16_t>()};
        kv.k += kv.off(r0, 0, 0);
        kv.vf += kv.off(r0, 0, 0);
        // diagnostic (make DIAG=1): in-situ phase stamps of ONE layer's five launches inside a real step
        unsigned long long* dbg = (stamp_buf_ && l == stamp_layer_) ? stamp_buf_ : nullptr;
        const size_t dbg_stride = (size_t)512 * 16 * 8;
        DecGemvArgs a{};
        a.B = nr;
        auto gemv = [&](DecEpi epi, const QuantImg& qi, const bf16_t* norm_w) { decode_gemv(epi, a, qi, norm_w, h, s); };
        const RopeRows rr = rope_rows(r0);
        if (qa) {
            DecQaArgs q{x, L.ln1, L.wqkv_p, qkv, gs.ctx_len, L.qn, L.kn, rr.cos_rows, rr.sin_rows, kv, at, nr, cfg_.rms_eps,
                        1.0f / sqrtf((float)hd), d_chain_ctr_.as<unsigned>(), (unsigned)l, d_err_flag_,
                        (qa_dbg_ && l == cfg_.dec_layers / 2) ? qa_dbg_ : nullptr};
            q.gran = d_qa_gran_.as<unsigned long long>();
            q.part = d_qa_part_.as<unsigned long long>();
            if (decw_.quant) { q.wq_qp = L.qkv_q.qp; q.wq_sb = L.qkv_q.sb; q.wq_bits = L.qkv_q.bits; }
            decode_qa_launch(q, s);
        } else {
            if (chain < 3 || l == 0) {         // chain 3: layer l's q|k|v came out of layer l - 1's launch
                a.W = L.wqkv; a.Wp = L.wqkv_p; a.X = x; a.B = nr; a.N = nh * hd; a.K = H; a.out = qkv;
                decode_gemv_set_debug(dbg);
                gemv(DEC_EPI_BF16, L.qkv_q, L.ln1);
            }
            decode_attention_launch(qkv, gs.ctx_len, nr, cfg_.heads, cfg_.kv_heads, hd, L.qn, L.kn, cfg_.rms_eps,
                                    rr.cos_rows, rr.sin_rows, kv, at, s, dbg ? dbg + 4 * dbg_stride : nullptr);
        }
        if (chain) {
            const bool last = l + 1 == cfg_.dec_layers;
            const DecLayerW& Ln = decw_.layers[last ? l : l + 1];
            DecChainArgs c{at, L.wo_p, x, L.ln2, L.wgu_p, act, L.wdown_p, Ln.ln1, Ln.wqkv_p, qkv, nr, cfg_.rms_eps,
                           d_chain_ctr_.as<unsigned>(), (unsigned)l, d_err_flag_, (chain_dbg_ && l == cfg_.dec_layers / 2) ? chain_dbg_ : nullptr};
            int phases = CHAIN_O | CHAIN_GU;
            if (chain >= 2) phases |= CHAIN_DOWN;
            if (chain >= 3 && !last) phases |= CHAIN_QKV;
            decode_chain_launch(phases, c, s);
            if (chain == 1) {
                a.B = nr; a.W = L.wdown; a.Wp = L.wdown_p; a.X = act; a.N = H; a.K = I; a.out = x;
                gemv(DEC_EPI_RESID, L.down_q, nullptr);
            }
            continue;
        }
        a.W = L.wo; a.Wp = L.wo_p; a.X = at; a.N = H; a.K = nq; a.out = x;
        decode_gemv_set_debug(dbg ? dbg + dbg_stride : nullptr);
        gemv(DEC_EPI_RESID, L.o_q, nullptr);
        a.W = L.wgu; a.Wp = L.wgu_p; a.X = x; a.N = 2 * I; a.K = H; a.out = act;
        decode_gemv_set_debug(dbg ? dbg + 2 * dbg_stride : nullptr);
        gemv(DEC_EPI_SWIGLU, L.gu_q, L.ln2);
        a.W = L.wdown; a.Wp = L.wdown_p; a.X = act; a.N = H; a.K = I; a.out = x;
        decode_gemv_set_debug(dbg ? dbg + 3 * dbg_stride : nullptr);
        gemv(DEC_EPI_RESID, L.down_q, nullptr);
        decode_gemv_set_debug(nullptr);
    }
    if (!with_head) return;
    run_lm_head(want_logits, r0, nr, s);
    if (greedy)
        greedy_finalize_launch(d_part_val_.as<float>() + (size_t)r0 * n_parts_, d_part_idx_.as<int>() + (size_t)r0 * n_parts_,
                               n_parts_, gs, nr, 1, decw_.embed, x, H, rope_rows(r0), s, decw_.quant ? &decw_.embed_raw : nullptr);
}

// A whole step = `split` row groups.  The small-batch decode kernels are latency-bound (a 4..12 MB weight
// matrix per launch cannot fill the chip), so independent row groups issued on parallel branches overlap
// one group's weight streaming with another's attention; the second reader of a layer's weights is served
// from the Infinity Cache.
static int decode_split_env() {
    return tuning().decode_split;   // A/B: 1 won once the GEMVs were packed
}
static int decode_gran_env() {
    return tuning().decode_gran;
}

// rows of the first (largest) row group of a step
int Engine::decode_group_rows() const {
    const int B = batch_, gran = decode_gran_env();
    int split = std::min(decode_split_env(), 4);
    if (split <= 1 || B < 2 * gran) return B;
    const int tiles = (B + gran - 1) / gran;
    split = std::min(split, tiles);
    const int t = tiles / split + (tiles % split ? 1 : 0);
    return std::min(B, t * gran);
}

void Engine::issue_decode_step(int split) {
    const int B = batch_;
    const int gran = decode_gran_env();
    if (split <= 1 || B < 2 * gran) {
        run_decode_step(false, true, 0, B, stream_, true);
        return;
    }
    if (split > 4) split = 4;
    // row groups are multiples of `gran` rows (16 = one MFMA batch tile) except the last
    const int tiles = (B + gran - 1) / gran;
    if (split > tiles) split = tiles;
    QASR_HIP(hipEventRecord(fork_ev_, stream_));
    int r0 = 0;
    for (int i = 0; i < split; ++i) {
        const int t = tiles / split + (i < tiles % split ? 1 : 0);
        const int nr = std::min(B - r0, t * gran);
        hipStream_t s = i == 0 ? stream_ : side_[i - 1];
        if (i > 0) QASR_HIP(hipStreamWaitEvent(s, fork_ev_, 0));
        run_decode_step(false, true, r0, nr, s, false);           // layers only; the head runs once after the join
        if (i > 0) {
            QASR_HIP(hipEventRecord(join_ev_[i - 1], s));
            QASR_HIP(hipStreamWaitEvent(stream_, join_ev_[i - 1], 0));
        }
        r0 += nr;
    }
    // LM head streams 311 MB of tied-embedding weights: once per step for all rows, not once per row group
    run_lm_head(false, 0, B, stream_);
    greedy_finalize_launch(d_part_val_.as<float>(), d_part_idx_.as<int>(), n_parts_, gstate_, B, 1, decw_.embed,
                           d_dx_.as<bf16_t>(), cfg_.hidden, rope_rows(0), stream_, decw_.quant ? &decw_.embed_raw : nullptr);
}

// Non-default decoding options on the device: pick from the f32 logits the LM head wrote (dec_sampler.h: sampler_pick_launch), then
// the same bookkeeping + embedding gather as the greedy path.  The row partials reuse the front of the LM head's partial buffers.
void Engine::sample_and_finalize(int advance_ctx) {
    sampler_pick_launch(d_logits_.as<float>(), cfg_.vocab, gstate_, batch_, opt_rep_penalty_, opt_ngram_, opt_temperature_,
                        (unsigned long long)opt_seed_, d_part_val_.as<float>(), d_part_idx_.as<int>(), stream_);
    greedy_finalize_launch(d_part_val_.as<float>(), d_part_idx_.as<int>(), 1, gstate_, batch_, advance_ctx, decw_.embed,
                           d_dx_.as<bf16_t>(), cfg_.hidden, rope_rows(0), stream_, decw_.quant ? &decw_.embed_raw : nullptr);
}

// Greedy loop (Qwen3ASR.swift:344-389): token 0 comes from the prompt pass; every further token costs
// one decode step.  The step is captured once into a hipGraph (all per-step state -- ctx_len, tokens,
// finished -- lives in HBM, so the kernel arguments never change) and replayed.  With natural EOS the
// host polls n_active every 8 steps; with ignore_eos the loop is free of host synchronisation.
void Engine::decode_loop() {
    const int max_steps = cur_max_tokens_ - 1;
    if (max_steps <= 0) return;
    const int split = decode_split_env();
    if (split > 1 && !fork_ev_) {
        // side streams only for the row-group A/B knob (decode_split > 1), made before any capture starts: an engine that never splits
        // owns one compute stream, so that engines sharing a GPU (qasr_dp_submit) each get a hardware queue of their own
        QASR_HIP(hipEventCreateWithFlags(&fork_ev_, hipEventDisableTiming));
        for (int i = 0; i < 3; ++i) {
            QASR_HIP(hipStreamCreateWithFlags(&side_[i], hipStreamNonBlocking));
            QASR_HIP(hipEventCreateWithFlags(&join_ev_[i], hipEventDisableTiming));
        }
    }
    // sampled: the slow path's step (logits of every row, pickNextToken, bookkeeping) entirely on the device; the options are kernel
    // arguments of the captured step, so they are part of the graph key
    const bool sampled = slow_path_;
    long key = ((long)batch_ << 32) | ((long)split << 24) | ((long)cur_max_tokens_ << 1) | (cur_ignore_eos_ ? 1 : 0);
    if (sampled) {
        unsigned long long h = 0x9e3779b97f4a7c15ull;
        auto mixin = [&](unsigned long long v) { h = (h ^ v) * 0xbf58476d1ce4e5b9ull; h ^= h >> 29; };
        unsigned u;
        std::memcpy(&u, &opt_rep_penalty_, 4); mixin(u);
        std::memcpy(&u, &opt_temperature_, 4); mixin(u);
        mixin((unsigned long long)(unsigned)opt_ngram_);
        mixin((unsigned long long)opt_seed_);
        key ^= (long)(h | (1ull << 62));
    }
    const bool use_graph_ = tuning().use_graph != 0;
    auto issue_step = [&]() {
        if (sampled) {
            run_decode_step(true, false, 0, batch_, stream_, true);
            sample_and_finalize(1);
        } else issue_decode_step(split);
    };
    // steps per graph launch: the step's kernel arguments never change, so S consecutive steps are the same nodes S times; the
    // EOS poll below happens every 8 steps, so S divides 8
    const int gs = tuning().graph_steps;
    const int S = (gs == 2 || gs == 4 || gs == 8) ? gs : 1;
    auto capture = [&](int n, hipGraphExec_t* exec) {
        hipGraph_t g = nullptr;
        QASR_HIP(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
        try {
            for (int i = 0; i < n; ++i) issue_step();
        } catch (...) {
            (void)hipStreamEndCapture(stream_, &g);
            if (g) (void)hipGraphDestroy(g);
            throw;
        }
        QASR_HIP(hipStreamEndCapture(stream_, &g));
        QASR_HIP(hipGraphInstantiate(exec, g, nullptr, nullptr, 0));
        QASR_HIP(hipGraphDestroy(g));
    };
    // a knob change (qasr_set_tuning) can select other kernels: the captured step is stale then
    if (use_graph_ && (graph_exec_ == nullptr || graph_key_ != key || graph_epoch_ != tuning().epoch || graph_n_ != S)) {
        drop_graph();
        graph_epoch_ = tuning().epoch;
        capture(1, &graph_exec_);
        if (S > 1) capture(S, &graph_exec_n_);
        graph_n_ = S;
        graph_key_ = key;
    }
    int h_active = batch_;
    for (int step = 0; step < max_steps; ++step) {
        if (use_graph_ && S > 1 && step % S == 0 && step + S <= max_steps) {
            QASR_HIP(hipGraphLaunch(graph_exec_n_, stream_));
            steps_done_ += S;
            step += S - 1;
        } else {
            if (use_graph_) QASR_HIP(hipGraphLaunch(graph_exec_, stream_));
            else issue_step();
            ++steps_done_;
        }
        if (!cur_ignore_eos_ && (step % 8 == 7)) {
            QASR_HIP(hipMemcpyAsync(&h_active, gstate_.n_active, sizeof(int), hipMemcpyDeviceToHost, stream_));
            QASR_HIP(hipStreamSynchronize(stream_));
            if (h_active <= 0) break;
        }
    }
}

int32_t pick_next_token(const float* logits, int32_t vocab, const int32_t* generated, int32_t n_gen,
                        float repetition_penalty, int32_t ngram, float temperature, uint64_t* rng_state);

__global__ void set_ints_kernel(int* dst, const int* src, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// Slow path (generateSlow, Qwen3ASR.swift:396-433): the logits of every row come back to the host each step and
// pickNextToken runs on the CPU, exactly like the reference; the step itself is the same HIP decode step.
void Engine::decode_loop_slow() {
    const int B = batch_, V = cfg_.vocab, stride = cfg_.max_new_tokens + 1;
    std::vector<float> logits((size_t)B * V);
    std::vector<std::vector<int32_t>> gen(B);
    std::vector<int> done(B, 0), next(B, 0);
    std::vector<uint64_t> rng(B);
    for (int b = 0; b < B; ++b) rng[b] = opt_seed_ * 0x9e3779b97f4a7c15ull + (uint64_t)b + 1;
    HostBuf h_tok;
    h_tok.alloc((size_t)B * sizeof(int));
    DevBuf d_tok;
    d_tok.alloc((size_t)B * sizeof(int));
    for (int step = 0; step < cur_max_tokens_; ++step) {
        QASR_HIP(hipMemcpyAsync(logits.data(), d_logits_.p, logits.size() * sizeof(float), hipMemcpyDeviceToHost, stream_));
        QASR_HIP(hipStreamSynchronize(stream_));
        bool all_done = true;
        for (int b = 0; b < B; ++b) {
            if (done[b]) continue;
            const int32_t t = pick_next_token(&logits[(size_t)b * V], V, gen[b].data(), (int32_t)gen[b].size(),
                                              opt_rep_penalty_, opt_ngram_, opt_temperature_, &rng[b]);
            gen[b].push_back(t);
            next[b] = t;
            if ((t == cfg_.tok_im_end && !cur_ignore_eos_) || (int)gen[b].size() >= cur_max_tokens_) done[b] = 1;
            all_done = all_done && done[b];
        }
        if (all_done || step + 1 >= cur_max_tokens_) break;
        // feed the picked ids: x = embed[token], rope rows for the current positions, one decode step with logits
        std::memcpy(h_tok.p, next.data(), (size_t)B * sizeof(int));
        QASR_HIP(hipMemcpyAsync(d_tok.p, h_tok.p, (size_t)B * sizeof(int), hipMemcpyHostToDevice, stream_));
        embed_rows(d_tok.as<int>(), d_dx_.as<bf16_t>(), B, stream_);
        hipLaunchKernelGGL(refresh_rope_rows_kernel, dim3(B), dim3(64), 0, stream_, gstate_.ctx_len, rope_rows(0));
        run_decode_step(true, false, 0, B, stream_, true);
        hipLaunchKernelGGL(add_scalar_kernel, dim3(cdiv(B, 64)), dim3(64), 0, stream_, gstate_.ctx_len, B, 1);
        ++steps_done_;
    }
    // publish through the same device-side result block the fast path uses
    std::vector<int> flat((size_t)B * stride, -1), lens(B);
    for (int b = 0; b < B; ++b) {
        std::copy(gen[b].begin(), gen[b].end(), flat.begin() + (size_t)b * stride);
        lens[b] = (int)gen[b].size();
    }
    QASR_HIP(hipMemcpyAsync(gstate_.tokens, flat.data(), flat.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
    QASR_HIP(hipMemcpyAsync(gstate_.lens, lens.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, stream_));
    QASR_HIP(hipStreamSynchronize(stream_));
}

// ---- batch pipeline --------------------------------------------------------------------------------
void Engine::batch_begin(const float* const* pcm, const size_t* n, size_t B, const qasr_options* opt) {
    if (!finalized_) throw std::runtime_error("weights not finalized");
    require_asr("transcribe");
    if (B == 0) throw std::invalid_argument("empty batch");
    int max_tokens = opt && opt->max_tokens > 0 ? opt->max_tokens : cfg_.max_new_tokens;
    if (max_tokens > cfg_.max_new_tokens) throw std::length_error("max_tokens exceeds engine capacity");
    // Qwen3DecodingOptions: anything but the defaults selects the slow path (isGreedyFastPath, Qwen3ASR.swift:300-304)
    opt_rep_penalty_ = opt && opt->repetition_penalty != 0.0f ? opt->repetition_penalty : 1.0f;
    opt_ngram_ = opt ? opt->no_repeat_ngram_size : 0;
    opt_temperature_ = opt ? opt->temperature : 0.0f;
    opt_seed_ = opt ? opt->seed : 0;
    slow_path_ = !(opt_rep_penalty_ == 1.0f && opt_ngram_ == 0 && opt_temperature_ == 0.0f);
    // the pinned staging buffers (pcm, plans) are reused per batch: the previous batch's copies must have drained.  No wait
    // at the end: the uploads run under the host-side planning and the kernels of qasr_batch_run queue behind them.
    QASR_HIP(hipStreamSynchronize(stream_));
    upload_pcm(pcm, n, B);
    plan_batch(opt, max_tokens);
}

void Engine::plan_batch(const qasr_options* opt, int max_tokens) {
    plan_encoder();
    std::vector<int> n_audio;
    for (auto& c : clips_) n_audio.push_back(c.n_tokens);
    plan_prefill(opt, n_audio);
    reset_greedy_state(max_tokens, opt && opt->ignore_eos);
}

// qasr_batch_begin_staged: adopt the batch qasr_batch_stage uploaded ahead; only the planning is left to do here.
void Engine::batch_begin_staged(const qasr_options* opt) {
    if (!finalized_) throw std::runtime_error("weights not finalized");
    require_asr("transcribe");
    if (!staged_valid_) throw std::invalid_argument("batch_begin_staged: no staged batch (qasr_batch_stage, or it was overwritten by qasr_batch_begin)");
    int max_tokens = opt && opt->max_tokens > 0 ? opt->max_tokens : cfg_.max_new_tokens;
    if (max_tokens > cfg_.max_new_tokens) throw std::length_error("max_tokens exceeds engine capacity");
    opt_rep_penalty_ = opt && opt->repetition_penalty != 0.0f ? opt->repetition_penalty : 1.0f;
    opt_ngram_ = opt ? opt->no_repeat_ngram_size : 0;
    opt_temperature_ = opt ? opt->temperature : 0.0f;
    opt_seed_ = opt ? opt->seed : 0;
    slow_path_ = !(opt_rep_penalty_ == 1.0f && opt_ngram_ == 0 && opt_temperature_ == 0.0f);
    QASR_HIP(hipStreamSynchronize(stream_));                    // plan buffers are reused per batch
    clips_.swap(staged_clips_);
    batch_ = staged_B_;
    batch_max_frames_all_ = staged_max_frames_all_;
    staged_valid_ = false;
    run_issued_ = false;
    pcm_staged_over_ = false;
    d_pcm_off_ = d_meta_.as<long>();
    d_n_samples_ = reinterpret_cast<int*>(d_pcm_off_ + batch_);
    d_frame_off_ = d_n_samples_ + batch_;
    QASR_HIP(hipStreamWaitEvent(stream_, ev_stage_done_, 0));   // the kernels of qasr_batch_run queue behind the staged copies
    plan_batch(opt, max_tokens);
}

void Engine::batch_run() {
    require_batch("batch_run");
    if (pcm_staged_over_)
        throw std::invalid_argument("batch_run: qasr_batch_stage has replaced this batch's samples in the device PCM buffer; a batch cannot be run "
                                    "again (qasr_batch_rewind) once its successor is staged");
    hipStream_t s = stream_;
    QASR_HIP(hipEventRecord(ev_[0], s));
    run_mel();
    QASR_HIP(hipEventRecord(ev_[1], s));
    if (ev_mel_done_) QASR_HIP(hipEventRecord(ev_mel_done_, s));       // the device PCM / meta buffers are free for the next staged batch
    run_issued_ = true;
    run_encoder();
    QASR_HIP(hipEventRecord(ev_[2], s));
    run_prefill(slow_path_);
    const bool host_sampler = slow_path_ && tuning().device_sampler == 0;
    if (!slow_path_)
        greedy_finalize_launch(d_part_val_.as<float>(), d_part_idx_.as<int>(), n_parts_, gstate_, batch_, 0, decw_.embed,
                               d_dx_.as<bf16_t>(), cfg_.hidden, rope_rows(0), s, decw_.quant ? &decw_.embed_raw : nullptr);
    else if (!host_sampler) sample_and_finalize(0);
    QASR_HIP(hipEventRecord(ev_[3], s));
    if (host_sampler) decode_loop_slow();
    else decode_loop();
    QASR_HIP(hipEventRecord(ev_[4], s));
}

void Engine::batch_rewind() {
    require_batch("batch_rewind");
    QASR_HIP(hipStreamSynchronize(stream_));       // h_ginit_ is reused
    reset_greedy_state(cur_max_tokens_, cur_ignore_eos_);
}

void Engine::batch_sync() { QASR_HIP(hipStreamSynchronize(stream_)); }

void Engine::batch_tokens(int32_t* tokens, int32_t* lens) {
    const int stride = cfg_.max_new_tokens + 1;
    QASR_HIP(hipMemcpyAsync(tokens, gstate_.tokens, (size_t)batch_ * stride * sizeof(int), hipMemcpyDeviceToHost, stream_));
    QASR_HIP(hipMemcpyAsync(lens, gstate_.lens, (size_t)batch_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
    int err = 0;
    QASR_HIP(hipMemcpyAsync(&err, d_err_flag_, sizeof(int), hipMemcpyDeviceToHost, stream_));
    QASR_HIP(hipStreamSynchronize(stream_));
    if (err & CHAIN_ERR_TIMEOUT)
        throw HipError("decode chain: an in-launch hand-off wait gave up (a workgroup of the persistent grid was not resident within the budget); tokens are not valid");
    if (err) throw HipError("greedy decode saw a non-finite best logit (NaN / inf in the weights or activations); tokens are not valid");
}

void Engine::batch_timings(float ms[5], int32_t* n_steps) {
    QASR_HIP(hipStreamSynchronize(stream_));
    for (int i = 0; i < 4; ++i) QASR_HIP(hipEventElapsedTime(&ms[i], ev_[i], ev_[i + 1]));
    QASR_HIP(hipEventElapsedTime(&ms[4], ev_[0], ev_[4]));
    if (n_steps) *n_steps = steps_done_;
}

// Average duration of `reps` back-to-back launches of one decode-step kernel group on the engine
// stream (HIP events on that stream), with the current batch's state.  Algorithmic bytes:
//   0: weight-streaming GEMVs of ONE decoder layer (qkv + o + gate/up + down weights, bf16)
//   1: decode attention of ONE layer (K + V rows of every batch row at its current ctx)
//   2: LM head (vocab x hidden bf16)
//   3 / 4: prompt-pass q|k|v and gate|up GEMMs, 5: prompt attention of one layer (FLOPs instead of bytes)
void Engine::kernel_probe(int which, int reps, float* avg_ms, double* bytes_per_launch) {
    if (!finalized_ || batch_ <= 0) throw std::runtime_error("kernel_probe needs a prepared batch");
    require_asr("kernel_probe");
    const int H = cfg_.hidden, hd = cfg_.head_dim, nq = cfg_.heads * hd, nh = cfg_.heads + 2 * cfg_.kv_heads, I = cfg_.inter;
    const DecLayerW& L = decw_.layers[0];
    KVLayout kv{kcache_[0]->as<bf16_t>(), nullptr, max_ctx_, cfg_.kv_heads, hd, vfcache_[0]->as<bf16_t>()};
    hipStream_t s = stream_;
    std::vector<int> ctx(batch_);
    QASR_HIP(hipMemcpy(ctx.data(), gstate_.ctx_len, batch_ * sizeof(int), hipMemcpyDeviceToHost));
    // same launch shape as the captured step: one row group of the batch (see issue_decode_step)
    const int rows = decode_group_rows();
    const PromptW probe_w = (which == 3 || which == 4) ? prompt_weights(0, s) : PromptW{};
    long probe_layer = 0;
    const int probe_chain = rows == batch_ ? step_chain(0, rows) : 0;
    const bool probe_qa = rows == batch_ && step_qa(0, rows, probe_chain);
    hipEvent_t* probe_ev = nullptr;            // set per timed launch of the fused probe: the counter memset stays outside the event pair
    auto body = [&]() {
        DecGemvArgs a{};
        a.B = rows;
        auto gemv = [&](DecEpi epi, const QuantImg& qi, const bf16_t* norm_w) { decode_gemv(epi, a, qi, norm_w, d_dh_.as<bf16_t>(), s); };
        if (which == 0) {
            // the same launches as run_decode_step (residual epilogues write a scratch row block), one LAYER AFTER THE OTHER like
            // the step: a layer's 31 MB come back only after the other layers' 0.85 GB went through the caches, so every launch streams
            // its weights from HBM as in situ (one layer in a loop would sit in the Infinity Cache and time 15 % short)
            const DecLayerW& L = decw_.layers[(size_t)(probe_layer++ % cfg_.dec_layers)];
            if (!probe_qa) {                   // with q|k|v + attention as one launch the projection belongs to probe 1
                a.W = L.wqkv; a.Wp = L.wqkv_p; a.X = d_dx_.as<bf16_t>(); a.N = nh * hd; a.K = H; a.out = d_dqkv_.as<bf16_t>();
                gemv(DEC_EPI_BF16, L.qkv_q, L.ln1);
            }
            a.W = L.wo; a.Wp = L.wo_p; a.X = d_dattn_.as<bf16_t>(); a.N = H; a.K = nq; a.out = d_dh_.as<bf16_t>();
            gemv(DEC_EPI_RESID, L.o_q, nullptr);
            a.W = L.wgu; a.Wp = L.wgu_p; a.X = d_dx_.as<bf16_t>(); a.N = 2 * I; a.K = H; a.out = d_dact_.as<bf16_t>();
            gemv(DEC_EPI_SWIGLU, L.gu_q, L.ln2);
            a.W = L.wdown; a.Wp = L.wdown_p; a.X = d_dact_.as<bf16_t>(); a.N = H; a.K = I; a.out = d_dh_.as<bf16_t>();
            gemv(DEC_EPI_RESID, L.down_q, nullptr);
        } else if (which == 1) {
            const RopeRows rr = rope_rows(0);
            if (probe_qa) {
                // the fused launch on layer after layer: its q|k|v weights AND its K / V rows come from HBM every time, as in the step
                const int l = (int)(probe_layer++ % cfg_.dec_layers);
                const DecLayerW& Lq = decw_.layers[(size_t)l];
                KVLayout kvl{kcache_[l]->as<bf16_t>(), nullptr, max_ctx_, cfg_.kv_heads, hd, vfcache_[l]->as<bf16_t>()};
                decode_chain_reset(d_chain_ctr_.as<unsigned>(), s);
                DecQaArgs q{d_dx_.as<bf16_t>(), Lq.ln1, Lq.wqkv_p, d_dqkv_.as<bf16_t>(), gstate_.ctx_len, Lq.qn, Lq.kn, rr.cos_rows, rr.sin_rows,
                            kvl, d_dattn_.as<bf16_t>(), rows, cfg_.rms_eps, 1.0f / sqrtf((float)hd), d_chain_ctr_.as<unsigned>(), 0u, d_err_flag_};
                q.gran = d_qa_gran_.as<unsigned long long>();
                q.part = d_qa_part_.as<unsigned long long>();
                if (decw_.quant) { q.wq_qp = Lq.qkv_q.qp; q.wq_sb = Lq.qkv_q.sb; q.wq_bits = Lq.qkv_q.bits; }
                if (probe_ev) QASR_HIP(hipEventRecord(probe_ev[0], s));
                decode_qa_launch(q, s);
                if (probe_ev) QASR_HIP(hipEventRecord(probe_ev[1], s));
            } else
            decode_attention_launch(d_dqkv_.as<bf16_t>(), gstate_.ctx_len, rows, cfg_.heads, cfg_.kv_heads, hd, L.qn,
                                    L.kn, cfg_.rms_eps, rr.cos_rows, rr.sin_rows, kv, d_dattn_.as<bf16_t>(), s);
        } else if (which == 3) {      // prompt-pass QKV GEMM shape (M = packed prompt rows, N = 4096, K = 1024)
            gemm_nt(ADense{d_ph_.as<bf16_t>(), H, n_pos_, H}, probe_w.wqkv, H, n_pos_, nh * hd, H,
                    EpiStoreBf16{d_pqkv_.as<bf16_t>(), (long)nh * hd}, s);
        } else if (which == 4) {      // prompt-pass gate/up GEMM with the SwiGLU epilogue
            gemm_nt_swiglu(ADense{d_ph_.as<bf16_t>(), H, n_pos_, H}, probe_w.wgu, H, n_pos_, 2 * I, H,
                           EpiStoreBf16{d_pact_.as<bf16_t>(), I}, s);
        } else if (which == 5) {      // prompt attention of layer 0 on the prepared batch (q rows and K / V images of the last prompt pass)
            prefill_attention_launch(d_pqr_.as<bf16_t>(), kv, d_vt_.as<bf16_t>(), vt_stride_, d_p_cu_, d_p_slotclip_, batch_, max_len_,
                                     cfg_.heads, d_pattn_.as<bf16_t>(), s);
        } else {
            run_lm_head(false, 0, batch_, s);
        }
    };
    for (int i = 0; i < 3; ++i) body();
    float ms = 0;
    if (which == 1 && probe_qa) {
        // the fused launch, one event pair per launch (set inside body around the launch itself)
        std::vector<hipEvent_t> ev(2 * reps);
        for (auto& e : ev) QASR_HIP(hipEventCreate(&e));
        for (int i = 0; i < reps; ++i) { probe_ev = &ev[2 * i]; body(); }
        probe_ev = nullptr;
        QASR_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < reps; ++i) {
            float t = 0;
            QASR_HIP(hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]));
            ms += t;
        }
        for (auto& e : ev) (void)hipEventDestroy(e);
    } else if (which == 1) {
        // in-situ-like timing of the single attention kernel: every timed launch is preceded by an untimed
        // weight-streaming launch (as in the real step, which evicts the query rows / rope table from the
        // near caches), and bracketed by its own event pair on the engine stream
        std::vector<hipEvent_t> ev(2 * reps);
        for (auto& e : ev) QASR_HIP(hipEventCreate(&e));
        DecGemvArgs a{};
        a.B = rows; a.W = L.wqkv; a.Wp = L.wqkv_p; a.X = d_dx_.as<bf16_t>(); a.N = nh * hd; a.K = H; a.out = d_dqkv_.as<bf16_t>();
        for (int i = 0; i < reps; ++i) {
            decode_gemv(DEC_EPI_BF16, a, L.qkv_q, L.ln1, d_dh_.as<bf16_t>(), s);
            QASR_HIP(hipEventRecord(ev[2 * i], s));
            body();
            QASR_HIP(hipEventRecord(ev[2 * i + 1], s));
        }
        QASR_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < reps; ++i) {
            float t = 0;
            QASR_HIP(hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]));
            ms += t;
        }
        for (auto& e : ev) (void)hipEventDestroy(e);
    } else {
        QASR_HIP(hipEventRecord(ev_[5], s));
        hipEvent_t e1;
        QASR_HIP(hipEventCreate(&e1));
        for (int i = 0; i < reps; ++i) body();
        QASR_HIP(hipEventRecord(e1, s));
        QASR_HIP(hipEventSynchronize(e1));
        QASR_HIP(hipEventElapsedTime(&ms, ev_[5], e1));
        (void)hipEventDestroy(e1);
    }
    *avg_ms = ms / (float)reps;
    if (which == 1 && tuning().da_stamps) {
        // diagnostic: one launch with phase stamps (100 MHz wall clock), printed as averages over workgroups / waves
        const int nw = 16, nwg = rows * cfg_.kv_heads;
        DevBuf d;
        d.alloc((size_t)nwg * nw * 8 * sizeof(unsigned long long));
        QASR_HIP(hipMemsetAsync(d.p, 0, d.bytes, s));
        const RopeRows rr = rope_rows(0);
        decode_attention_launch(d_dqkv_.as<bf16_t>(), gstate_.ctx_len, rows, cfg_.heads, cfg_.kv_heads, hd, L.qn, L.kn,
                                cfg_.rms_eps, rr.cos_rows, rr.sin_rows, kv, d_dattn_.as<bf16_t>(), s, d.as<unsigned long long>());
        std::vector<unsigned long long> h((size_t)nwg * nw * 8);
        QASR_HIP(hipMemcpyAsync(h.data(), d.p, d.bytes, hipMemcpyDeviceToHost, s));
        QASR_HIP(hipStreamSynchronize(s));
        unsigned long long t0 = ~0ull, t6 = 0;
        for (size_t i = 0; i < h.size(); i += 8) if (h[i]) { t0 = std::min(t0, h[i]); t6 = std::max(t6, h[i + 6]); }
        double acc[7] = {0}; int cnt = 0;
        for (size_t i = 0; i < h.size(); i += 8) if (h[i]) { for (int k = 0; k < 7; ++k) acc[k] += (double)(h[i + k] - t0); ++cnt; }
        fprintf(stderr, "[qasr] attention stamps (us since first wave start, mean over %d waves): entry %.2f | phase1 done %.2f | "
                "after sync %.2f | sweeps done %.2f | slot merge done %.2f | after sync %.2f | end %.2f | span %.2f\n", cnt,
                acc[0] / cnt / 100, acc[1] / cnt / 100, acc[2] / cnt / 100, acc[3] / cnt / 100, acc[4] / cnt / 100,
                acc[5] / cnt / 100, acc[6] / cnt / 100, (double)(t6 - t0) / 100);
    }
    if (which == 5 && tuning().pa_stamps) {
        // diagnostic: one launch with per-wave phase sums (100 MHz wall clock); printed per query tile (= number of key tiles swept)
        const int qt = (max_len_ + 63) / 64, nwg = qt * cfg_.kv_heads * batch_;
        DevBuf d;
        d.alloc((size_t)nwg * 4 * 8 * sizeof(unsigned long long));
        QASR_HIP(hipMemsetAsync(d.p, 0, d.bytes, s));
        prefill_attention_launch(d_pqr_.as<bf16_t>(), kv, d_vt_.as<bf16_t>(), vt_stride_, d_p_cu_, d_p_slotclip_, batch_, max_len_,
                                 cfg_.heads, d_pattn_.as<bf16_t>(), s, d.as<unsigned long long>());
        std::vector<unsigned long long> h((size_t)nwg * 4 * 8);
        QASR_HIP(hipMemcpyAsync(h.data(), d.p, d.bytes, hipMemcpyDeviceToHost, s));
        QASR_HIP(hipStreamSynchronize(s));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (size_t i = 0; i < h.size(); i += 8) if (h[i]) { t0 = std::min(t0, h[i]); t1 = std::max(t1, h[i + 1]); }
        // heavy-first grid (kv head, clip, query tile): blockIdx.z = qt - 1 - tile
        for (int z = 0; z < qt; ++z) {
            double acc[8] = {0}; int cnt = 0;
            for (size_t w = (size_t)z * cfg_.kv_heads * batch_ * 4; w < (size_t)(z + 1) * cfg_.kv_heads * batch_ * 4; ++w) {
                const unsigned long long* e = &h[w * 8];
                if (!e[0]) continue;
                acc[0] += (double)(e[0] - t0); acc[1] += (double)(e[1] - e[0]);
                for (int k = 0; k < 6; ++k) acc[2 + k] += (double)e[2 + k];
                ++cnt;
            }
            if (!cnt) continue;
            fprintf(stderr, "[qasr] prompt attention stamps, query tile %d (%d key tiles), mean over %d waves, us: start %.2f | resident %.2f | prologue %.2f | "
                    "copy issue %.2f | S^T %.2f | softmax %.2f | P V^T %.2f | copy wait + barrier %.2f\n", qt - 1 - z, qt - z, cnt, acc[0] / cnt / 100,
                    acc[1] / cnt / 100, acc[2] / cnt / 100, acc[3] / cnt / 100, acc[4] / cnt / 100, acc[5] / cnt / 100, acc[6] / cnt / 100, acc[7] / cnt / 100);
        }
        fprintf(stderr, "[qasr] prompt attention stamps: span %.2f us\n", (double)(t1 - t0) / 100);
    }
    if (which == 6 || which == 7) {
        // diagnostic: one real (eager) decode step with the middle layer's persistent launch stamped (dec_chain.hip / dec_qa.hip, ST instantiations)
        if (which == 6 && !tuning().chain) throw std::invalid_argument("kernel_probe 6: set the chain knob first");
        if (which == 7 && !tuning().qa) throw std::invalid_argument("kernel_probe 7: set the qa knob first");
        DevBuf d;
        const size_t n = (size_t)256 * 32;
        d.alloc(n * sizeof(unsigned long long));
        std::vector<unsigned long long> hst(n);
        static const char* qnames[19] = {"entry", "rows staged", "proj summed", "signalled", "wait over", "w1 rows+K/V in", "w1 sweep done", "stored",
                                         "w0 query ready", "w1 own K/V in", "", "", "", "", "", "", "", "", ""};
        static const char* cnames[19] = {"entry", "O staged", "O summed", "O signalled", "GU wait over", "GU rows in", "GU staged", "GU summed",
                                        "GU signalled", "DOWN wait over", "DOWN rows in", "DOWN staged", "DOWN summed", "DOWN signalled",
                                        "QKV wait over", "QKV rows in", "QKV staged", "QKV summed", "QKV stored"};
        for (int rep = 0; rep < 3; ++rep) {
            QASR_HIP(hipMemsetAsync(d.p, 0, d.bytes, s));
            (which == 6 ? chain_dbg_ : qa_dbg_) = d.as<unsigned long long>();
            run_decode_step(false, false, 0, rows, s, true);
            chain_dbg_ = nullptr;
            qa_dbg_ = nullptr;
            QASR_HIP(hipMemcpyAsync(hst.data(), d.p, d.bytes, hipMemcpyDeviceToHost, s));
            QASR_HIP(hipStreamSynchronize(s));
            if (rep < 2) continue;
            unsigned long long t0 = ~0ull;
            for (size_t w = 0; w < 256; ++w) if (hst[w * 32]) t0 = std::min(t0, hst[w * 32]);
            for (int q = 0; q < 19; ++q) {
                double sum = 0, lo = 1e30, hi = 0; int cnt = 0;
                for (size_t w = 0; w < 256; ++w) if (hst[w * 32 + q]) { const double v = (double)(hst[w * 32 + q] - t0) / 100; sum += v; lo = std::min(lo, v); hi = std::max(hi, v); ++cnt; }
                const char* const* names = which == 6 ? cnames : qnames;
                if (cnt) fprintf(stderr, "[qasr] chain stamps %-16s workgroups %3d  min %6.2f  mean %6.2f  max %6.2f us\n", names[q], cnt, lo, sum / cnt, hi);
            }
        }
        *avg_ms = 0.f;
        *bytes_per_launch = 0;
        return;
    }
    if (which == 0 && tuning().stamps_insitu) {
        // one real decode step (eager, all layers, cold weights) with layer 14's five launches stamped
        DevBuf d;
        const size_t stride = (size_t)512 * 16 * 8, n = 5 * stride;
        d.alloc(n * sizeof(unsigned long long));
        std::vector<unsigned long long> hst(n);
        const char* names[5] = {"qkv (norm)", "o-proj (resid)", "gate/up (norm, swiglu)", "down (resid)", "attention"};
        for (int rep = 0; rep < 2; ++rep) {
            QASR_HIP(hipMemsetAsync(d.p, 0, d.bytes, s));
            stamp_buf_ = d.as<unsigned long long>();
            stamp_layer_ = cfg_.dec_layers / 2;
            run_decode_step(false, false, 0, rows, s, true);
            stamp_buf_ = nullptr;
            QASR_HIP(hipMemcpyAsync(hst.data(), d.p, d.bytes, hipMemcpyDeviceToHost, s));
            QASR_HIP(hipStreamSynchronize(s));
            unsigned long long first = ~0ull;
            for (size_t i = 0; i < n; i += 8) if (hst[i]) first = std::min(first, hst[i]);
            for (int k : {0, 4, 1, 2, 3}) {
                unsigned long long t0 = ~0ull, t1 = 0;
                const unsigned long long* h = hst.data() + k * stride;
                for (size_t i = 0; i < stride; i += 8) if (h[i]) { t0 = std::min(t0, h[i]); for (int q = 0; q < 7; ++q) t1 = std::max(t1, h[i + q]); }
                double acc[7] = {0}; int cnt[7] = {0};
                for (size_t i = 0; i < stride; i += 8) if (h[i]) for (int q = 0; q < 7; ++q) if (h[i + q]) { acc[q] += (double)(h[i + q] - t0); ++cnt[q]; }
                fprintf(stderr, "[qasr] in-situ %-24s starts at %.2f us, span %.2f us; mean stamps:", names[k], (double)(t0 - first) / 100, (double)(t1 - t0) / 100);
                for (int q = 0; q < 7; ++q) fprintf(stderr, " %.2f", cnt[q] ? acc[q] / cnt[q] / 100 : 0.0);
                fprintf(stderr, "\n");
            }
        }
    }
    if (which == 0 && tuning().gemv_stamps && !decw_.quant) {   // stamps live in the bf16 kernels only
        DevBuf d;
        const size_t n = (size_t)512 * 16 * 8;
        d.alloc(n * sizeof(unsigned long long));
        std::vector<unsigned long long> hst(n);
        const char* names[4] = {"qkv (norm)", "o-proj (resid)", "gate/up (norm, swiglu)", "down (resid)"};
        for (int k = 0; k < 4; ++k) {
            QASR_HIP(hipMemsetAsync(d.p, 0, d.bytes, s));
            DecGemvArgs a{};
            a.B = rows;
            decode_gemv_set_debug(d.as<unsigned long long>());
            if (k == 0) { a.W = L.wqkv; a.Wp = L.wqkv_p; a.X = d_dx_.as<bf16_t>(); a.N = nh * hd; a.K = H; a.out = d_dqkv_.as<bf16_t>();
                          decode_gemv_fused_launch(DEC_EPI_BF16, a, L.ln1, cfg_.rms_eps, d_dh_.as<bf16_t>(), s); }
            if (k == 1) { a.W = L.wo; a.Wp = L.wo_p; a.X = d_dattn_.as<bf16_t>(); a.N = H; a.K = nq; a.out = d_dh_.as<bf16_t>();
                          decode_gemv_fused_launch(DEC_EPI_RESID, a, nullptr, 0.f, nullptr, s); }
            if (k == 2) { a.W = L.wgu; a.Wp = L.wgu_p; a.X = d_dx_.as<bf16_t>(); a.N = 2 * I; a.K = H; a.out = d_dact_.as<bf16_t>();
                          decode_gemv_fused_launch(DEC_EPI_SWIGLU, a, L.ln2, cfg_.rms_eps, d_dh_.as<bf16_t>(), s); }
            if (k == 3) { a.W = L.wdown; a.Wp = L.wdown_p; a.X = d_dact_.as<bf16_t>(); a.N = H; a.K = I; a.out = d_dh_.as<bf16_t>();
                          decode_gemv_fused_launch(DEC_EPI_RESID, a, nullptr, 0.f, nullptr, s); }
            decode_gemv_set_debug(nullptr);
            QASR_HIP(hipMemcpyAsync(hst.data(), d.p, d.bytes, hipMemcpyDeviceToHost, s));
            QASR_HIP(hipStreamSynchronize(s));
            unsigned long long t0 = ~0ull, t1 = 0;
            for (size_t i = 0; i < n; i += 8) if (hst[i]) { t0 = std::min(t0, hst[i]); for (int q = 0; q < 7; ++q) t1 = std::max(t1, hst[i + q]); }
            double acc[7] = {0}; int cnt[7] = {0};
            for (size_t i = 0; i < n; i += 8) if (hst[i]) for (int q = 0; q < 7; ++q) if (hst[i + q]) { acc[q] += (double)(hst[i + q] - t0); ++cnt[q]; }
            fprintf(stderr, "[qasr] gemv stamps %-24s (us since first wave, mean over %d waves): entry %.2f | loads issued %.2f | X staged %.2f | "
                    "after sync %.2f | MFMA done %.2f | after reduce sync %.2f | end(w0) %.2f | span %.2f\n", names[k], cnt[0],
                    acc[0] / cnt[0] / 100, acc[1] / cnt[1] / 100, acc[2] / cnt[2] / 100, acc[3] / cnt[3] / 100, acc[4] / cnt[4] / 100,
                    acc[5] / cnt[5] / 100, cnt[6] ? acc[6] / cnt[6] / 100 : 0.0, (double)(t1 - t0) / 100);
        }
    }
    double bytes = 0;
    // weight bytes per element: 2 (bf16) or bits / 8 + scale and bias per 64 elements
    const double wbytes = !decw_.quant ? 2.0 : cfg_.bits / 8.0 + 2.0 * (decw_.embed_raw.sb_f32 ? 4.0 : 2.0) / 64.0;
    if (which == 0) bytes = wbytes * ((probe_qa ? 0.0 : (double)nh * hd * H) + (double)H * nq + 2.0 * I * H + (double)H * I);
    else if (which == 1) {
        for (int b = 0; b < rows; ++b) bytes += 2.0 * 2.0 * cfg_.kv_heads * hd * (double)ctx[b];
        if (probe_qa) bytes += wbytes * (double)nh * hd * H;
    }
    else if (which == 3) bytes = 2.0 * (double)n_pos_ * nh * hd * H;          // FLOPs for the GEMM probes
    else if (which == 4) bytes = 2.0 * (double)n_pos_ * 2 * I * H;
    else if (which == 5) {      // causal FLOPs: per clip and head 4 hd sum_t (t + 1) = 2 hd T (T + 1)
        std::vector<int> cu(batch_ + 1);
        QASR_HIP(hipMemcpy(cu.data(), d_p_cu_, (batch_ + 1) * sizeof(int), hipMemcpyDeviceToHost));
        for (int b = 0; b < batch_; ++b) { const double T = cu[b + 1] - cu[b]; bytes += 2.0 * hd * T * (T + 1.0) * cfg_.heads; }
    }
    else bytes = wbytes * (double)cfg_.vocab * H;
    *bytes_per_launch = bytes;
}

// ---- stage entry points ------------------------------------------------------------------------------
void Engine::prefill_logits_host(const float* audio_embeds, int n_audio, const qasr_options* opt, float* logits) {
    if (!finalized_) throw std::runtime_error("weights not finalized");
    require_asr("prefill_logits");
    if (n_audio < 0 || n_audio > max_tokens_) throw std::length_error("prefill: too many audio tokens");
    batch_ = 1;
    clips_.assign(1, ClipPlan{});
    clips_[0].n_tokens = n_audio;
    clip_tok_off_.assign(1, 0);
    const long n = (long)n_audio * cfg_.hidden;
    if (n > 0) {
        float* stage = d_encx_.as<float>();   // f32 staging (capacity max_tokens * d_model >= n? checked below)
        if ((size_t)n * sizeof(float) > d_encx_.bytes) throw std::length_error("prefill: staging too small");
        QASR_HIP(hipMemcpyAsync(stage, audio_embeds, n * sizeof(float), hipMemcpyHostToDevice, stream_));
        hipLaunchKernelGGL(narrow_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream_, stage, d_audio_.as<bf16_t>(), n);
    }
    plan_prefill(opt, std::vector<int>{n_audio});
    reset_greedy_state(cfg_.max_new_tokens, true);
    run_prefill(true);
    QASR_HIP(hipMemcpyAsync(logits, d_logits_.p, (size_t)cfg_.vocab * sizeof(float), hipMemcpyDeviceToHost, stream_));
    QASR_HIP(hipStreamSynchronize(stream_));
}

void Engine::decode_forced_host(const int32_t* tokens, int n, float* logits) {
    if (!finalized_) throw NotLoaded("decode_forced: weights not finalized");
    if (batch_ != 1 || h_ctx0_.empty()) throw std::runtime_error("decode_forced needs a preceding prefill_logits");
    require_asr("decode_forced");
    // every forced token appends one K/V row at ctx_len and reads the RoPE row of that position: stay inside the
    // cache (max_ctx_ rows; the last row is never a query position of the greedy path either)
    if (n < 0 || forced_ctx_ + n > max_ctx_ - 1)
        throw std::length_error("decode_forced: " + std::to_string(forced_ctx_) + " cached positions + " + std::to_string(n) +
                                " forced tokens exceed the cache capacity of " + std::to_string(max_ctx_ - 1));
    HostBuf idx;
    idx.alloc(sizeof(int));
    DevBuf didx;
    didx.alloc(sizeof(int));
    for (int i = 0; i < n; ++i) {
        if (tokens[i] < 0 || tokens[i] >= cfg_.vocab) throw std::invalid_argument("token id out of range");
        *idx.as<int>() = tokens[i];
        QASR_HIP(hipMemcpyAsync(didx.p, idx.p, sizeof(int), hipMemcpyHostToDevice, stream_));
        embed_rows(didx.as<int>(), d_dx_.as<bf16_t>(), 1, stream_);
        hipLaunchKernelGGL(refresh_rope_rows_kernel, dim3(1), dim3(64), 0, stream_, gstate_.ctx_len, rope_rows(0));
        run_decode_step(true, false, 0, 1, stream_, true);
        hipLaunchKernelGGL(add_scalar_kernel, dim3(1), dim3(64), 0, stream_, gstate_.ctx_len, 1, 1);
        QASR_HIP(hipMemcpyAsync(logits + (size_t)i * cfg_.vocab, d_logits_.p, (size_t)cfg_.vocab * sizeof(float),
                                hipMemcpyDeviceToHost, stream_));
        QASR_HIP(hipStreamSynchronize(stream_));
        ++forced_ctx_;
    }
}

}  // namespace qasr

"""Synthetic weights and waveforms (there is no checkpoint or dataset offline).

Recipes are the ones SURVEY.md section 8(d) fixes, so bench numbers and parity tests are
reproducible:
  * weights: `torch.manual_seed(seed)`; N(0, 0.02) for Linear/Conv/Embedding, norm weight 1,
    bias 0 (HF `initializer_range`), cast to bf16.  `init="stress"` instead draws
    N(0, 1/fan_in) weights, non-zero biases and non-unit norm gains so that attention is
    peaked and every bias/gain path is exercised by the parity tests.
  * waveform k: 0.4 sin(2 pi f1 t) + 0.2 sin(2 pi f2 t) + 0.05 N(0,1), f1 = 220 + 37k,
    f2 = 1200 + 91k, rng(20260418 + k), quantised to PCM16 and back.
Tensor names/layouts are the reference checkpoint's (Sources/Qwen3ASR/WeightLoading.swift:
235-323): conv weights [out, kH, kW, in]; Linear [out, in]; float decoder (FloatTextDecoder).
"""
import math
import numpy as np
import torch


def synth_waveform(k: int, seconds: float, sample_rate: int = 16000) -> np.ndarray:
    n = int(round(seconds * sample_rate))
    rng = np.random.default_rng(20260418 + k)
    t = np.arange(n, dtype=np.float64) / sample_rate
    f1, f2 = 220.0 + 37.0 * k, 1200.0 + 91.0 * k
    x = 0.4 * np.sin(2 * np.pi * f1 * t) + 0.2 * np.sin(2 * np.pi * f2 * t) + 0.05 * rng.standard_normal(n)
    pcm16 = np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)
    return (pcm16.astype(np.float32) / np.float32(32768.0)).astype(np.float32)


def synth_state_dict(audio_cfg, text_cfg, seed: int = 0, init: str = "hf", dtype=torch.bfloat16, classify_num: int = 0):
    """audio_cfg / text_cfg: any objects with the fields of qasr.config presets.  classify_num > 0 adds the forced
    aligner's `lm_head.{weight,bias}` Linear(hidden, classify_num) (WeightLoading.swift:228-230)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def w(name, *shape, fan_in=None):
        std = 0.02 if init == "hf" else 1.0 / math.sqrt(fan_in if fan_in else shape[-1])
        sd[name] = (torch.randn(*shape, generator=g) * std).to(dtype)

    def b(name, n):
        sd[name] = (torch.zeros(n) if init == "hf" else torch.randn(n, generator=g) * 0.05).to(dtype)

    def gain(name, n):
        sd[name] = (torch.ones(n) if init == "hf" else 1.0 + 0.1 * torch.randn(n, generator=g)).to(dtype)

    a = audio_cfg
    C = a.conv_channels
    w("audio_tower.conv2d1.weight", C, 3, 3, 1, fan_in=9)
    b("audio_tower.conv2d1.bias", C)
    for n in ("conv2d2", "conv2d3"):
        w(f"audio_tower.{n}.weight", C, 3, 3, C, fan_in=9 * C)
        b(f"audio_tower.{n}.bias", C)
    w("audio_tower.conv_out.weight", a.d_model, a.conv_out_in)
    for i in range(a.layers):
        p = f"audio_tower.layers.{i}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            w(f"{p}.self_attn.{n}.weight", a.d_model, a.d_model)
            b(f"{p}.self_attn.{n}.bias", a.d_model)
        for n in ("self_attn_layer_norm", "final_layer_norm"):
            gain(f"{p}.{n}.weight", a.d_model)
            b(f"{p}.{n}.bias", a.d_model)
        w(f"{p}.fc1.weight", a.ffn_dim, a.d_model)
        b(f"{p}.fc1.bias", a.ffn_dim)
        w(f"{p}.fc2.weight", a.d_model, a.ffn_dim)
        b(f"{p}.fc2.bias", a.d_model)
    gain("audio_tower.ln_post.weight", a.d_model)
    b("audio_tower.ln_post.bias", a.d_model)
    w("audio_tower.proj1.weight", a.d_model, a.d_model)
    b("audio_tower.proj1.bias", a.d_model)
    w("audio_tower.proj2.weight", a.output_dim, a.d_model)
    b("audio_tower.proj2.bias", a.output_dim)

    t = text_cfg
    sd["model.embed_tokens.weight"] = (torch.randn(t.vocab, t.hidden, generator=g) *
                                       (0.02 if init == "hf" else 0.05)).to(dtype)
    for i in range(t.layers):
        p = f"model.layers.{i}"
        w(f"{p}.self_attn.q_proj.weight", t.heads * t.head_dim, t.hidden)
        w(f"{p}.self_attn.k_proj.weight", t.kv_heads * t.head_dim, t.hidden)
        w(f"{p}.self_attn.v_proj.weight", t.kv_heads * t.head_dim, t.hidden)
        w(f"{p}.self_attn.o_proj.weight", t.hidden, t.heads * t.head_dim)
        gain(f"{p}.self_attn.q_norm.weight", t.head_dim)
        gain(f"{p}.self_attn.k_norm.weight", t.head_dim)
        gain(f"{p}.input_layernorm.weight", t.hidden)
        gain(f"{p}.post_attention_layernorm.weight", t.hidden)
        w(f"{p}.mlp.gate_proj.weight", t.inter, t.hidden)
        w(f"{p}.mlp.up_proj.weight", t.inter, t.hidden)
        w(f"{p}.mlp.down_proj.weight", t.hidden, t.inter)
    gain("model.norm.weight", t.hidden)
    if classify_num:
        std = 0.02 if init == "hf" else 1.0 / math.sqrt(t.hidden)
        sd["lm_head.weight"] = (torch.randn(classify_num, t.hidden, generator=g) * std).to(dtype)
        sd["lm_head.bias"] = (torch.randn(classify_num, generator=g) * 0.05).to(dtype)
    return sd


QUANT_GROUP = 64


def quantize_linear(w, bits, group=QUANT_GROUP):
    """MLX affine quantisation of one [out, in] weight (mlx `quantize`: per 64-element group of a row, scale and bias from
    the group's min / max, q in [0, 2^bits)) -> (packed int32 [out, in * bits / 32] holding the uint32 words, LSB-first
    element order, scales, biases in w's dtype).  This is how a synthetic MLX-4bit / 8bit checkpoint is made for the tests
    and the W4 / W8 bench legs; the on-disk format is the reference's (Sources/MLXCommon/WeightLoading.swift:48-96)."""
    dt = w.dtype
    x = w.to(torch.float32)
    out, n = x.shape
    assert n % group == 0 and bits in (4, 8)
    g = x.reshape(out, n // group, group)
    n_bins = float((1 << bits) - 1)
    w_max, w_min = g.max(dim=-1).values, g.min(dim=-1).values
    mask = w_min.abs() > w_max.abs()
    scales = torch.clamp((w_max - w_min) / n_bins, min=1e-7)
    scales = torch.where(mask, scales, -scales)
    edge = torch.where(mask, w_min, w_max)
    q0 = torch.round(edge / scales)
    scales = torch.where(q0 != 0, edge / q0, scales)
    biases = torch.where(q0 == 0, torch.zeros_like(edge), edge)
    scales, biases = scales.to(dt), biases.to(dt)
    q = torch.clamp(torch.round((g - biases.to(torch.float32)[..., None]) / scales.to(torch.float32)[..., None]), 0, n_bins)
    q = q.reshape(out, n).to(torch.int64)
    per = 32 // bits
    q = q.reshape(out, n // per, per)
    shifts = torch.arange(per, dtype=torch.int64) * bits
    words = (q << shifts).sum(dim=-1)                       # < 2^32
    words = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
    return words.contiguous(), scales.contiguous(), biases.contiguous()


def quantize_state_dict(sd, bits, group=QUANT_GROUP):
    """Float decoder state dict -> the quantised checkpoint layout: every `model.layers.*` Linear and the tied
    `model.embed_tokens` become {weight: packed uint32 (as int32 bits), scales, biases}; norms and the audio tower stay
    float (QuantizedTextDecoder.swift:178-199, AudioEncoder is not quantised)."""
    out = {}
    for k, v in sd.items():
        is_lin = k.startswith("model.layers.") and k.endswith("_proj.weight")
        if is_lin or k == "model.embed_tokens.weight":
            stem = k[:-len(".weight")]
            out[k], out[stem + ".scales"], out[stem + ".biases"] = quantize_linear(v, bits, group)
        else:
            out[k] = v
    return out


class _TensorSink(dict):
    """dict stand-in that hands every assigned tensor to a callback and keeps nothing (a 7B variant is 26 GB of f32 on the host otherwise)."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def __setitem__(self, name, tensor):
        self.fn(name, tensor)


def synth_omnilingual_state_dict(cfg, seed=0, bits=0, dtype=torch.float32, sink=None):
    """Seeded random weights of an Omnilingual (wav2vec2-CTC) model under the reference's fairseq2 tensor names
    (MLX/OmnilingualMLXWeightLoader.swift:40-135): PyTorch Conv1d layout [out, in, k], weight_g / weight_v for the
    positional conv, every encoder / head Linear with a bias.  cfg: object with model_dim, layers, heads, ffn_dim,
    feature_dim, pos_kernel, pos_groups, vocab.  bits 4 / 8: the encoder and head linears as MLX triplets (f32 scales, as
    the reference's loader widens them), else float weights.  sink(name, tensor): stream the tensors to a consumer instead of returning
    them (same seeded values in the same order)."""
    g = torch.Generator().manual_seed(seed)
    sd = {} if sink is None else _TensorSink(sink)
    kernels = (10, 3, 3, 3, 3, 2, 2)

    def rnd(*shape, std):
        return (torch.randn(*shape, generator=g) * std).to(dtype)

    C, D, F = cfg.feature_dim, cfg.model_dim, cfg.ffn_dim
    for i, k in enumerate(kernels):
        p = f"encoder_frontend.feature_extractor.layers.{i}"
        cin = 1 if i == 0 else C
        sd[p + ".conv.weight"] = rnd(C, cin, k, std=1.0 / math.sqrt(cin * k))
        sd[p + ".conv.bias"] = rnd(C, std=0.05)
        sd[p + ".layer_norm.weight"] = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dtype)
        sd[p + ".layer_norm.bias"] = rnd(C, std=0.05)
    sd["encoder_frontend.post_extract_layer_norm.weight"] = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dtype)
    sd["encoder_frontend.post_extract_layer_norm.bias"] = rnd(C, std=0.05)
    sd["encoder_frontend.model_dim_proj.weight"] = rnd(D, C, std=1.0 / math.sqrt(C))
    sd["encoder_frontend.model_dim_proj.bias"] = rnd(D, std=0.05)
    cpg = D // cfg.pos_groups
    sd["encoder_frontend.pos_encoder.conv.weight_v"] = rnd(D, cpg, cfg.pos_kernel, std=1.0)
    sd["encoder_frontend.pos_encoder.conv.weight_g"] = (0.5 + 0.1 * torch.randn(1, 1, cfg.pos_kernel, generator=g)).abs().to(dtype)
    sd["encoder_frontend.pos_encoder.conv.bias"] = rnd(D, std=0.05)

    def lin(stem, n, k):
        w = rnd(n, k, std=1.0 / math.sqrt(k))
        if bits in (4, 8):
            wq, s, b = quantize_linear(w.to(torch.bfloat16), bits)
            sd[stem + ".weight"], sd[stem + ".scales"], sd[stem + ".biases"] = wq, s.to(torch.float32), b.to(torch.float32)
        else:
            sd[stem + ".weight"] = w
        sd[stem + ".bias"] = rnd(n, std=0.05)

    for l in range(cfg.layers):
        p = f"encoder.layers.{l}"
        for nme in ("q_proj", "k_proj", "v_proj", "output_proj"):
            lin(f"{p}.self_attn.{nme}", D, D)
        lin(f"{p}.ffn.inner_proj", F, D)
        lin(f"{p}.ffn.output_proj", D, F)
        for nme in ("self_attn_layer_norm", "ffn_layer_norm"):
            sd[f"{p}.{nme}.weight"] = (1.0 + 0.1 * torch.randn(D, generator=g)).to(dtype)
            sd[f"{p}.{nme}.bias"] = rnd(D, std=0.05)
    sd["encoder.layer_norm.weight"] = (1.0 + 0.1 * torch.randn(D, generator=g)).to(dtype)
    sd["encoder.layer_norm.bias"] = rnd(D, std=0.05)
    lin("final_proj", cfg.vocab, D)
    return sd

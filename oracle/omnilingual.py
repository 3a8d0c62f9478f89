"""Omnilingual ASR (wav2vec2 encoder + CTC head): CPU restatement of `OmnilingualASRMLXModel.transcribeAudio`.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  BASELINE configs[3]; SURVEY.md section 8f N4.

Follows Sources/OmnilingualASR:
  * OmnilingualASR.swift:305-325            `layerNormalize`: utterance-level (x - mean) / sqrt(var + 1e-5), biased variance,
                                           single-pass f32 sums (sum, sum of squares), var = max(0, E[x^2] - mean^2)
  * MLX/Wav2Vec2Frontend.swift:12-86        feature extractor: 7 x (Conv1d k=[10,3,3,3,3,2,2] s=[5,2,2,2,2,2,2], no padding,
                                           bias) -> LayerNorm(512) -> GELU, channel-last; output length floor((L - k) / s) + 1
  * MLX/Wav2Vec2Frontend.swift:88-122       position encoder: grouped Conv1d (k 128, groups 16, padding 64), last frame
                                           trimmed (even kernel), GELU, residual
  * MLX/Wav2Vec2Frontend.swift:124-155      frontend = extractor -> post_extract LayerNorm -> Linear(512, D) -> position encoder
  * MLX/Wav2Vec2EncoderLayer.swift:13-94    pre-norm layer: x + attn(LN(x)); + ffn(LN(.)); q/k/v/output and the two FFN
                                           projections are QuantizedLinear with bias; SDPA without mask, scale 1/sqrt(64)
  * MLX/Wav2Vec2Encoder.swift:9-44          N layers, final LayerNorm; CTC head = QuantizedLinear(D, 10288, bias)
  * MLX/OmnilingualMLXModel.swift:141-210   40 s cap, empty input -> "", argmax per frame, collapse consecutive duplicates
  * MLX/OmnilingualMLXWeightLoader.swift:26-37,86-103  every non-uint32 tensor is cast to FLOAT32 at load (so the whole
                                           model computes in f32; scales / biases of the quantised linears too), Conv1d weights
                                           transposed from PyTorch [out, in, k], weight_norm(dim=2) fused:
                                           W[:, :, k] = g[k] * v[:, :, k] / max(||v[:, :, k]||, 1e-12)
  * CTCGreedyDecoder.swift:28-55            first maximum wins, consecutive duplicates collapse, blank is NOT removed here
  * SentencePieceVocabulary.swift:38-57     decode: drop ids {bos 0, pad 1, eos 2, unk 3} and control / unknown / unused / byte
                                           pieces, concatenate, U+2581 -> space, trim
Arithmetic in third-party mlx (Conv1d, LayerNorm, gelu = exact erf, quantizedMatmul, SDPA): all f32 here because the loader
widens everything; with f32 activations both mlx quantised kernels multiply by the unrounded scale * q + bias.

Policies (oracle/precision.py): REFERENCE = f32 everywhere (what MLX does).  DEVICE = the inputs of every contraction
(activations AND the dequantised / float weights) rounded to bf16, softmax probabilities rounded to bf16 -- the MFMA path.
"""
import math
import numpy as np
import torch
import torch.nn.functional as Fn
from dataclasses import dataclass
from . import precision as P
from . import quant as Q

KERNELS = (10, 3, 3, 3, 3, 2, 2)
STRIDES = (5, 2, 2, 2, 2, 2, 2)
MAX_AUDIO_SECONDS = 40.0
LN_EPS = 1e-5


@dataclass(frozen=True)
class OmniConfig:
    model_dim: int = 1024
    layers: int = 24
    heads: int = 16
    ffn_dim: int = 4096
    feature_dim: int = 512
    pos_kernel: int = 128
    pos_groups: int = 16
    vocab: int = 10288
    bits: int = 4
    group_size: int = 64
    ln_eps: float = 1e-5

    @property
    def head_dim(self):
        return self.model_dim // self.heads


# OmnilingualMLXConfig.variant (MLX/OmnilingualMLXConfig.swift:88-103)
VARIANTS = {
    "300M": OmniConfig(1024, 24, 16, 4096),
    "1B": OmniConfig(1280, 48, 20, 5120),
    "3B": OmniConfig(2048, 60, 32, 8192),
    "7B": OmniConfig(2048, 128, 32, 8192),
}
OMNI_TINY = OmniConfig(model_dim=64, layers=2, heads=2, ffn_dim=128, feature_dim=32, pos_kernel=16, pos_groups=4, vocab=40)


def output_length(n_samples):
    """Wav2Vec2FeatureExtractor.outputLength (Wav2Vec2Frontend.swift:47-54)."""
    L = n_samples
    for k, s in zip(KERNELS, STRIDES):
        L = (L - k) // s + 1
        if L <= 0:
            return 0
    return L


def layer_normalize(samples, eps=LN_EPS):
    """OmnilingualASR.swift:305-325, float32 single-pass sums in sample order."""
    x = np.asarray(samples, dtype=np.float32)
    n = x.shape[0]
    if n == 0:
        return x
    s = np.float32(0)
    ss = np.float32(0)
    # sequential f32 accumulation like the Swift loop (np.cumsum keeps the order and the dtype)
    s = np.cumsum(x, dtype=np.float32)[-1]
    ss = np.cumsum(x * x, dtype=np.float32)[-1]
    mean = np.float32(s / np.float32(n))
    var = np.float32(max(np.float32(0), np.float32(ss / np.float32(n)) - mean * mean))
    inv = np.float32(1) / np.sqrt(np.float32(var + np.float32(eps)))
    return ((x - mean) * inv).astype(np.float32)


def fuse_weight_norm(g, v):
    """OmnilingualMLXWeightLoader.swift:92-103: g [1, 1, K], v [out, in/groups, K] -> dense [out, in/groups, K]."""
    v = v.to(torch.float32)
    g = g.to(torch.float32)
    norm = torch.sqrt((v * v).sum(dim=(0, 1), keepdim=True))
    return g * v / torch.clamp(norm, min=1e-12)


class OmniWeights:
    """State dict wrapper (reference tensor names).  Linears are float `X.weight` or MLX triplets; everything is widened to
    f32 like the reference's loader does."""

    def __init__(self, sd):
        self.sd = sd
        self._c = {}

    def f32(self, key):
        t = self._c.get(key)
        if t is None:
            t = torch.as_tensor(self.sd[key]).to(torch.float32)
            self._c[key] = t
        return t

    def linear_weight(self, stem):
        t = self._c.get(stem)
        if t is None:
            if (stem + ".scales") in self.sd:
                wq = np.asarray(self.sd[stem + ".weight"]).view(np.uint32)
                s = np.asarray(torch.as_tensor(self.sd[stem + ".scales"]).to(torch.float32))
                b = np.asarray(torch.as_tensor(self.sd[stem + ".biases"]).to(torch.float32))
                bits = 32 * wq.shape[1] // (s.shape[1] * Q.GROUP)
                t = Q.dequantize_f32(wq, s, b, bits)
            else:
                t = self.f32(stem + ".weight")
            self._c[stem] = t
        return t


def _linear(x, W: OmniWeights, stem, pol: P.Policy):
    return pol.enc(x) @ pol.enc(W.linear_weight(stem)).T + W.f32(stem + ".bias")


def _ln(x, W: OmniWeights, stem, eps):
    return Fn.layer_norm(x, (x.shape[-1],), W.f32(stem + ".weight"), W.f32(stem + ".bias"), eps)


def feature_extractor(wave, W: OmniWeights, pol: P.Policy):
    """wave [T] f32 (already utterance-normalised) -> [T', feature_dim] f32."""
    x = torch.as_tensor(wave, dtype=torch.float32)[None, None, :]             # [1, C=1, T]
    for i, (k, s) in enumerate(zip(KERNELS, STRIDES)):
        p = f"encoder_frontend.feature_extractor.layers.{i}"
        w = W.f32(p + ".conv.weight")                                          # PyTorch layout [out, in, k]
        xin = x if i == 0 else pol.enc(x)                                      # the first conv consumes f32 samples on the device too
        wn = w if i == 0 else pol.enc(w)
        x = Fn.conv1d(xin, wn, W.f32(p + ".conv.bias"), stride=s)
        x = Fn.layer_norm(x.transpose(1, 2), (x.shape[1],), W.f32(p + ".layer_norm.weight"), W.f32(p + ".layer_norm.bias"), LN_EPS)
        x = Fn.gelu(x).transpose(1, 2)
    return x[0].T


def position_encoder(x, W: OmniWeights, cfg: OmniConfig, pol: P.Policy):
    """x [T, D] -> gelu(conv(x))[:T] + x  (Wav2Vec2Frontend.swift:113-121)."""
    p = "encoder_frontend.pos_encoder.conv"
    w = fuse_weight_norm(W.f32(p + ".weight_g"), W.f32(p + ".weight_v"))     # [out, in/groups, K]
    T = x.shape[0]
    h = Fn.conv1d(pol.enc(x).T[None], pol.enc(w), W.f32(p + ".bias"), padding=cfg.pos_kernel // 2, groups=cfg.pos_groups)[0].T
    if cfg.pos_kernel % 2 == 0:
        h = h[:T]
    return Fn.gelu(h) + x


def encoder_layer(x, W: OmniWeights, i, cfg: OmniConfig, pol: P.Policy):
    p = f"encoder.layers.{i}"
    T, D = x.shape
    h = _ln(x, W, p + ".self_attn_layer_norm", cfg.ln_eps)
    q = pol.enc(_linear(h, W, p + ".self_attn.q_proj", pol))
    k = pol.enc(_linear(h, W, p + ".self_attn.k_proj", pol))
    v = pol.enc(_linear(h, W, p + ".self_attn.v_proj", pol))
    hd = cfg.head_dim
    qh = q.reshape(T, cfg.heads, hd).transpose(0, 1)
    kh = k.reshape(T, cfg.heads, hd).transpose(0, 1)
    vh = v.reshape(T, cfg.heads, hd).transpose(0, 1)
    pr = torch.softmax((qh @ kh.transpose(1, 2)) * (1.0 / math.sqrt(hd)), dim=-1)
    a = (pol.enc(pr) @ vh).transpose(0, 1).reshape(T, D)
    x = x + _linear(a, W, p + ".self_attn.output_proj", pol)
    h = _ln(x, W, p + ".ffn_layer_norm", cfg.ln_eps)
    h = Fn.gelu(_linear(h, W, p + ".ffn.inner_proj", pol))
    return x + _linear(h, W, p + ".ffn.output_proj", pol)


def forward(samples, sd, cfg: OmniConfig, pol: P.Policy = P.REFERENCE, return_stages=False):
    """raw 16 kHz samples -> CTC logits [T', vocab] f32 (OmnilingualMLXModel.swift:165-177)."""
    W = sd if isinstance(sd, OmniWeights) else OmniWeights(sd)
    wave = layer_normalize(samples)
    feats = feature_extractor(wave, W, pol)
    h = _ln(feats, W, "encoder_frontend.post_extract_layer_norm", LN_EPS)
    h = _linear(h, W, "encoder_frontend.model_dim_proj", pol)
    x = position_encoder(h, W, cfg, pol)
    front = x
    for i in range(cfg.layers):
        x = encoder_layer(x, W, i, cfg, pol)
    x = _ln(x, W, "encoder.layer_norm", cfg.ln_eps)
    logits = _linear(x, W, "final_proj", pol)
    if return_stages:
        return logits, dict(features=feats, frontend=front, encoded=x)
    return logits


def ctc_greedy(logits, valid_frames=None):
    """CTCGreedyDecoder.decode: first maximum per frame, consecutive duplicates collapsed, blank kept."""
    lg = np.asarray(logits)
    T = lg.shape[0] if lg.ndim == 2 else 0
    frames = min(valid_frames if valid_frames is not None else T, T)
    out, prev = [], -1
    for t in range(frames):
        b = int(np.argmax(lg[t]))           # numpy argmax = first maximum
        if b != prev:
            out.append(b)
            prev = b
    return out


def collapse(ids):
    """OmnilingualMLXModel.collapseConsecutiveDuplicates (:195-209)."""
    out, prev = [], -1
    for i in ids:
        if i != prev:
            out.append(int(i))
            prev = i
    return out


# SentencePiece piece types (sentencepiece_model.proto): NORMAL 1, UNKNOWN 2, CONTROL 3, USER_DEFINED 4, UNUSED 5, BYTE 6
SPM_DROPPED_TYPES = {2, 3, 5, 6}


def vocab_decode(ids, pieces, special_ids=(0, 1, 2, 3)):
    """OmnilingualVocabulary.decode; pieces = list of (text, type).  Out-of-range ids are skipped."""
    out = ""
    for i in ids:
        if i < 0 or i >= len(pieces):
            continue
        text, typ = pieces[i]
        if i in special_ids or typ in SPM_DROPPED_TYPES:
            continue
        out += text
    return out.replace("▁", " ").strip(" \t\n\r")


def transcribe_ids(samples, sd, cfg: OmniConfig, pol: P.Policy = P.REFERENCE):
    """-> collapsed token ids (empty input -> []; > 40 s raises like the reference)."""
    n = len(samples)
    if n / 16000.0 > MAX_AUDIO_SECONDS:
        raise ValueError("input exceeds the Omnilingual cap of 40 s")
    if n == 0:
        return []
    with torch.no_grad():
        logits = forward(samples, sd, cfg, pol)
    return collapse(torch.argmax(logits, dim=-1).tolist())

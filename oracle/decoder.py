"""Text decoder + greedy loop: CPU restatement of `QuantizedTextModel` / `FloatTextModel`
and `Qwen3ASRModel.generateText`.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows:
  * Sources/Qwen3ASR/QuantizedTextDecoder.swift:56-105  attention: q/k/v proj, per-head q/k
    RMSNorm, split-half RoPE (offset = cache length), cache append, GQA SDPA, o_proj
  * :132-137 SwiGLU MLP; :156-174 pre-norm residual layer; :202-251 model forward,
    causal mask where(col > row + cacheLen, -1e9, 0) for T_q > 1, no mask for T_q == 1
  * Sources/Qwen3ASR/FloatTextDecoder.swift:35-226 (bf16 twin, identical structure, Linear)
  * Sources/MLXCommon/PreQuantizedEmbedding.swift:35-49 (embedding lookup; tied LM head)
  * Sources/Qwen3ASR/Qwen3ASR.swift:236-256 (splice audio embeddings, cast to embed dtype,
    last-position logits), :317-390 greedy loop (append then break on EOS), :449-520 sampler
Third-party arithmetic not in the reference tree (mlx-swift >= 0.30.0, unpinned):
  * RMSNorm: y = w * T(x * rsqrt(mean(x^2) + eps)), internal float32            (mlx fast::rms_norm)
  * RoPE(traditional=false): theta_i = base^(-i/(d/2)); out = [x1*cos - x2*sin, x1*sin + x2*cos]
  * outputs of every op are rounded to the tensor dtype (bf16 in the decoder)
  * argMax returns the lowest index among equal maxima; logits are bf16 (x dtype)
"""
import math
import numpy as np
import torch
from .config import TextDecoderConfig, TokenIds, TOKENS
from . import precision as P
from . import quant as Q


class Weights:
    """State dict wrapper with cached float32 views (weights are stored bf16/f32 tensors).

    A quantised checkpoint (QuantizedTextModel, QuantizedTextDecoder.swift:178-199) holds MLX triplets for every decoder
    Linear and the tied embedding: `X.weight` uint32, `X.scales`, `X.biases`; `linear` / `embed_rows` then follow
    oracle/quant.py (qmv for one row of x, qmm_t for the prompt pass, `dequantized` for the gather)."""

    def __init__(self, sd):
        self.sd = sd
        self._f32 = {}

    def __call__(self, key):
        t = self._f32.get(key)
        if t is None:
            t = self.sd[key].to(torch.float32)
            self._f32[key] = t
        return t

    def quantized(self, stem):
        return (stem + ".scales") in self.sd

    def _triplet(self, stem):
        wq = np.asarray(self.sd[stem + ".weight"]).view(np.uint32)
        s = np.asarray(torch.as_tensor(self.sd[stem + ".scales"]).to(torch.float32))
        b = np.asarray(torch.as_tensor(self.sd[stem + ".biases"]).to(torch.float32))
        bits = 32 * wq.shape[1] // (s.shape[1] * Q.GROUP)
        return wq, s, b, bits

    def _dequant(self, stem, rounded):
        key = (stem, rounded)
        t = self._f32.get(key)
        if t is None:
            wq, s, b, bits = self._triplet(stem)
            t = Q.dequantized(wq, s, b, bits) if rounded else Q.dequantize_f32(wq, s, b, bits)
            self._f32[key] = t
        return t

    def linear(self, x, stem):
        """x [rows, in] @ W^T (no bias, not yet rounded to the activation dtype)."""
        if not self.quantized(stem):
            return x @ self(stem + ".weight").T
        return x @ self._dequant(stem, rounded=x.shape[0] > Q.QMV_MAX_ROWS).T

    def embed_rows(self, ids):
        idx = torch.as_tensor(ids, dtype=torch.long)
        stem = "model.embed_tokens"
        if not self.quantized(stem):
            return self(stem + ".weight")[idx]
        wq, s, b, bits = self._triplet(stem)
        i = idx.numpy()
        return Q.dequantized(wq[i], s[i], b[i], bits)


def rms_norm(x, w, eps, pol: P.Policy):
    inv = torch.rsqrt((x * x).mean(dim=-1, keepdim=True) + eps)
    return pol.dec(w * pol.dec(x * inv))


def rope(x, positions, theta):
    """x [T, heads, hd]; split-half rotation (MLXNN.RoPE traditional=false)."""
    hd = x.shape[-1]
    half = hd // 2
    inv = torch.exp(torch.arange(half, dtype=torch.float32) * (-math.log(theta) / half))
    ang = positions.to(torch.float32)[:, None] * inv[None, :]
    cos, sin = torch.cos(ang)[:, None, :], torch.sin(ang)[:, None, :]
    x1, x2 = x[..., :half], x[..., half:]
    return torch.cat([x1 * cos - x2 * sin, x1 * sin + x2 * cos], dim=-1)


class DecoderState:
    """KV cache: per layer (K [kv_heads, ctx, hd], V [kv_heads, ctx, hd]) grown by concat."""

    def __init__(self, layers):
        self.k = [None] * layers
        self.v = [None] * layers

    @property
    def length(self):
        return 0 if self.k[0] is None else self.k[0].shape[1]


def flash_prefill_attention(qh, kh, vh, scale, off, tile=64):
    """Restatement of the HIP prompt-pass attention (csrc/dec_prefill.hip: prefill_attention_kernel):
    online softmax over key tiles of `tile`, un-normalised P rounded to bf16 before P.V, running
    sum taken over the rounded P, one division at the end.  qh [H, T, hd], kh/vh [H, ctx, hd].
    The reference's own SDPA internals (MLXFast, not in its tree) are unobservable; this is the
    DEVICE policy, compared against the plain-softmax REFERENCE policy with a stated tolerance."""
    H, T, hd = qh.shape
    ctx = kh.shape[1]
    row = (torch.arange(T) + off)[:, None]
    m = torch.full((H, T), float("-inf"))
    l = torch.zeros(H, T)
    o = torch.zeros(H, T, hd)
    for k0 in range(0, ctx, tile):
        k1 = min(k0 + tile, ctx)
        s = (qh @ kh[:, k0:k1].transpose(1, 2)) * scale
        col = torch.arange(k0, k1)[None, :]
        s = torch.where((col <= row)[None], s, torch.tensor(float("-inf")))
        m_new = torch.maximum(m, s.max(dim=-1).values)
        m_ref = torch.where(torch.isinf(m_new), torch.zeros_like(m_new), m_new)
        alpha = torch.exp(m - m_ref)
        p = P.bf16_round(torch.exp(s - m_ref[..., None]))
        l = l * alpha + p.sum(dim=-1)
        o = o * alpha[..., None] + p @ vh[:, k0:k1]
        m = m_new
    return o / l[..., None]


def forward(embeds, W: Weights, cfg: TextDecoderConfig, state: DecoderState, pol: P.Policy,
            p_bf16=None):
    """embeds [T, hidden] (already in decoder dtype) -> final-normed hidden [T, hidden]."""
    T = embeds.shape[0]
    off = state.length
    pos = torch.arange(off, off + T)
    flash = pol.name == "device" and T > 1
    if p_bf16 is None:
        p_bf16 = False                        # plain softmax keeps f32 probabilities (MLX SDPA)
    scale = 1.0 / math.sqrt(cfg.head_dim)
    rep = cfg.heads // cfg.kv_heads
    x = embeds
    for i in range(cfg.layers):
        p = f"model.layers.{i}"
        h = rms_norm(x, W(p + ".input_layernorm.weight"), cfg.rms_eps, pol)
        q = pol.dec(W.linear(h, p + ".self_attn.q_proj")).reshape(T, cfg.heads, cfg.head_dim)
        k = pol.dec(W.linear(h, p + ".self_attn.k_proj")).reshape(T, cfg.kv_heads, cfg.head_dim)
        v = pol.dec(W.linear(h, p + ".self_attn.v_proj")).reshape(T, cfg.kv_heads, cfg.head_dim)
        q = rms_norm(q, W(p + ".self_attn.q_norm.weight"), cfg.rms_eps, pol)
        k = rms_norm(k, W(p + ".self_attn.k_norm.weight"), cfg.rms_eps, pol)
        q = pol.dec(rope(q, pos, cfg.rope_theta))
        k = pol.dec(rope(k, pos, cfg.rope_theta))
        kT, vT = k.transpose(0, 1), v.transpose(0, 1)              # [kv, T, hd]
        if state.k[i] is not None:
            kT = torch.cat([state.k[i], kT], dim=1)
            vT = torch.cat([state.v[i], vT], dim=1)
        state.k[i], state.v[i] = kT, vT
        ctx = kT.shape[1]
        qh = q.transpose(0, 1)                                     # [heads, T, hd]
        kh = kT.repeat_interleave(rep, dim=0)
        vh = vT.repeat_interleave(rep, dim=0)
        if flash:
            av = flash_prefill_attention(qh, kh, vh, scale, off)
        else:
            sc = (qh @ kh.transpose(1, 2)) * scale                 # [heads, T, ctx]
            if T > 1:
                col = torch.arange(ctx)[None, :]
                row = (torch.arange(T) + off)[:, None]
                sc = sc + torch.where(col > row, torch.tensor(-1e9), torch.tensor(0.0))[None]
            pr = torch.softmax(sc, dim=-1)
            if p_bf16:
                pr = P.bf16_round(pr)
            av = pr @ vh
        a = pol.dec(av).transpose(0, 1).reshape(T, cfg.heads * cfg.head_dim)
        x = pol.dec(x + pol.dec(W.linear(a, p + ".self_attn.o_proj")))
        h = rms_norm(x, W(p + ".post_attention_layernorm.weight"), cfg.rms_eps, pol)
        g = pol.dec(W.linear(h, p + ".mlp.gate_proj"))
        u = pol.dec(W.linear(h, p + ".mlp.up_proj"))
        act = pol.dec(pol.dec(g * torch.sigmoid(g)) * u)
        x = pol.dec(x + pol.dec(W.linear(act, p + ".mlp.down_proj")))
    return rms_norm(x, W("model.norm.weight"), cfg.rms_eps, pol)


def embed(ids, W: Weights):
    return W.embed_rows(ids)


def lm_head(h_last, W: Weights, pol: P.Policy):
    """Tied head on one position: [hidden] -> [vocab] logits in the decoder dtype (`asLinear`, one row of x: qmv)."""
    return pol.dec(W.linear(h_last[None, :], "model.embed_tokens")[0])


def argmax_lowest(logits):
    """MLX argMax: first index of the maximum."""
    return int(torch.argmax(logits).item()) if logits.numel() else 0


def build_prompt(n_audio, tok: TokenIds = TOKENS, context_ids=None, language_ids=None):
    """Qwen3ASR.swift:199-233 -> (ids, audio_start_index)."""
    ids = [tok.im_start, tok.system, tok.newline]
    if context_ids:
        ids += list(context_ids)
    ids += [tok.im_end, tok.newline]
    ids += [tok.im_start, tok.user, tok.newline, tok.audio_start]
    a0 = len(ids)
    ids += [tok.audio_pad] * n_audio
    ids += [tok.audio_end, tok.im_end, tok.newline]
    ids += [tok.im_start, tok.assistant, tok.newline]
    if language_ids:
        ids += list(language_ids)
    ids.append(tok.asr_text)
    return ids, a0


def prefill(audio_embeds, W: Weights, cfg: TextDecoderConfig, pol: P.Policy,
            tok: TokenIds = TOKENS, context_ids=None, language_ids=None):
    """Qwen3ASR.swift:236-256 -> (logits [vocab], state, prompt ids)."""
    n_audio = audio_embeds.shape[0]
    ids, a0 = build_prompt(n_audio, tok, context_ids, language_ids)
    x = embed(ids, W).clone()
    x[a0:a0 + n_audio] = pol.dec(torch.as_tensor(audio_embeds, dtype=torch.float32))
    state = DecoderState(cfg.layers)
    h = forward(x, W, cfg, state, pol)
    return lm_head(h[-1], W, pol), state, ids


def decode_step(token, W: Weights, cfg: TextDecoderConfig, state: DecoderState, pol: P.Policy):
    h = forward(embed([token], W), W, cfg, state, pol)
    return lm_head(h[-1], W, pol)


def greedy(audio_embeds, W: Weights, cfg: TextDecoderConfig, pol: P.Policy = P.REFERENCE,
           tok: TokenIds = TOKENS, max_tokens=448, context_ids=None, language_ids=None,
           ignore_eos=False, return_logits=False):
    """`generateGreedyAsyncEval` Qwen3ASR.swift:317-390: EOS is appended, then the loop stops.

    `ignore_eos` forces exactly `max_tokens` tokens (bench workload: fixed decode length).
    """
    out, all_logits = [], []
    if max_tokens <= 0:
        return (out, all_logits) if return_logits else out
    logits, state, _ = prefill(audio_embeds, W, cfg, pol, tok, context_ids, language_ids)
    nxt = argmax_lowest(logits)
    for step in range(max_tokens):
        out.append(nxt)
        if return_logits:
            all_logits.append(logits)
        if nxt == tok.eos and not ignore_eos:
            break
        if step + 1 >= max_tokens:
            break
        logits = decode_step(nxt, W, cfg, state, pol)
        nxt = argmax_lowest(logits)
    return (out, all_logits) if return_logits else out


def pick_next_token(logits, generated, repetition_penalty=1.0, no_repeat_ngram=0, temperature=0.0,
                    rng=None):
    """`pickNextToken` Qwen3ASR.swift:449-520 (CPU sampler of the slow path)."""
    scores = np.asarray(logits, dtype=np.float32).reshape(-1).copy()
    V = scores.shape[0]
    if repetition_penalty == 1.0 and no_repeat_ngram == 0 and temperature == 0:
        return int(np.argmax(scores))
    if repetition_penalty > 1.0 and len(generated):
        for t in set(int(g) for g in generated):
            if 0 <= t < V:
                v = scores[t]
                scores[t] = v / np.float32(repetition_penalty) if v > 0 else v * np.float32(repetition_penalty)
    n = no_repeat_ngram
    if n > 0 and len(generated) >= n - 1:
        g = [int(x) for x in generated]
        last = g[len(g) - (n - 1):] if n > 1 else []
        if len(g) >= n:
            for i in range(0, len(g) - n + 1):
                if g[i:i + n - 1] == last:
                    f = g[i + n - 1]
                    if 0 <= f < V:
                        scores[f] = -np.inf
    if temperature > 0:
        rng = rng or np.random.default_rng()
        u = rng.uniform(1e-6, 1.0, size=V).astype(np.float32)
        scores = scores / np.float32(temperature) - np.log(-np.log(u))
    best, best_s = 0, -np.inf
    for i in range(V):                      # strict '>' : first maximum wins (:515-518)
        if scores[i] > best_s:
            best_s, best = scores[i], i
    return int(best)

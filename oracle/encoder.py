"""Audio encoder: CPU restatement of `Qwen3AudioEncoder.callAsFunction`.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows Sources/Qwen3ASR/AudioEncoder.swift:
  * :362-406  chunking into 100-frame pieces, last chunk zero-padded to maxChunkLen
  * :409-414  conv2d1/2/3 (3x3, stride 2, pad 1) + exact GELU, NHWC [chunk, mel, time, c]
  * :423-427  flatten [chunk, time, c*16 + f] and conv_out (7680 -> d_model, no bias)
  * :171-199,431-439  sinusoid PE, position restarts at 0 in every chunk
  * :442-460  keep valid tokens per chunk, concatenate
  * :464-489  attention windows of maxLenAfterCnn * (n_window_infer / chunk) tokens
  * :130-164  pre-LN block: x += out_proj(SDPA(q,k,v)); x += fc2(gelu(fc1(LN(x))))
  * :503-508  ln_post -> proj1 -> gelu -> proj2
  * Sources/MLXCommon/SDPA.swift:18-37  heads split, scale 1/sqrt(hd), additive -1e9 mask
Weights use the reference checkpoint key names / layouts (WeightLoading.swift:235-323):
conv weights are MLX layout [out, kH, kW, in].
"""
import math
import torch
import torch.nn.functional as Fn
from .config import AudioEncoderConfig
from . import precision as P


def conv_len(n):
    """One 3x3 / stride 2 / pad 1 conv on a length-n axis."""
    return (n - 1) // 2 + 1


def tokens_for_chunk(clen):
    """AudioEncoder.swift:443-449."""
    return conv_len(conv_len(conv_len(clen)))


def get_output_length(n_frames, chunk=100):
    """`getOutputLength` AudioEncoder.swift:287-303."""
    rem = n_frames % chunk
    feat = tokens_for_chunk(rem)
    full = (n_frames // chunk) * 13 if chunk == 100 else (n_frames // chunk) * tokens_for_chunk(chunk)
    return full + (max(feat, 1) if rem > 0 else 0)


def chunk_lengths(n_frames, chunk):
    """AudioEncoder.swift:367-378."""
    n_chunks = (n_frames + chunk - 1) // chunk
    out = []
    for i in range(n_chunks):
        if i == n_chunks - 1:
            rem = n_frames % chunk
            out.append(chunk if rem == 0 else rem)
        else:
            out.append(chunk)
    return out


def window_lengths(n_frames, cfg: AudioEncoderConfig):
    """AudioEncoder.swift:464-478: attention window lengths for one clip."""
    clens = chunk_lengths(n_frames, cfg.chunk)
    feat = [tokens_for_chunk(c) for c in clens]
    max_after = max(feat) if feat else 13
    window = max_after * (cfg.n_window_infer // cfg.chunk)
    total = get_output_length(n_frames, cfg.chunk)
    wins = [window] * (total // window)
    if total % window:
        wins.append(total % window)
    return wins


def sinusoid_pe(seq_len, d_model):
    """AudioEncoder.swift:171-199 ([sin | cos], float32)."""
    half = d_model // 2
    inc = torch.tensor(math.log(10000.0), dtype=torch.float32) / float(half - 1)
    inv = torch.exp(torch.arange(half, dtype=torch.float32) * (-inc))
    t = torch.arange(seq_len, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([torch.sin(t), torch.cos(t)], dim=1)


def _w(sd, key):
    """`sd` is either a plain state dict or a `decoder.Weights` (cached float32 views)."""
    return sd(key) if callable(sd) else sd[key].to(torch.float32)


# Summation-order probe (tests only): with FLIP_K set, every contraction runs over the reversed K axis -- the same
# mathematical sum associated differently in f32.  Two CPU evaluations that differ only in this flag measure how far
# bf16 rounding flips spread through the layers for reasons that have nothing to do with the device
# (tests/test_gpu_encoder.py: distributional bound).
FLIP_K = False


def _mm(x, w):
    """x [.., K] @ w[N, K]^T"""
    if FLIP_K:
        return x.flip(-1) @ w.flip(-1).T
    return x @ w.T


def _linear(x, sd, prefix, bias=True):
    y = _mm(x, _w(sd, prefix + ".weight"))
    if bias:
        y = y + _w(sd, prefix + ".bias")
    return y


def _gelu(x):
    return Fn.gelu(x)  # exact erf form, like MLXNN.gelu


def conv_stem(mel, sd, cfg: AudioEncoderConfig, pol: P.Policy):
    """mel [n_mels, T] f32 -> (tokens [sum valid, d_model] f32, chunk lengths)."""
    T = mel.shape[1]
    clens = chunk_lengths(T, cfg.chunk)
    max_len = max(clens)
    chunks = torch.zeros(len(clens), 1, cfg.n_mels, max_len, dtype=torch.float32)
    pos = 0
    for i, c in enumerate(clens):
        chunks[i, 0, :, :c] = mel[:, pos:pos + c]          # zero pad (:392-398)
        pos += c
    x = chunks
    for name in ("conv2d1", "conv2d2", "conv2d3"):
        w = _w(sd, f"audio_tower.{name}.weight").permute(0, 3, 1, 2)   # [o,kh,kw,i]->[o,i,kh,kw]
        b = _w(sd, f"audio_tower.{name}.bias")
        xin = x if name == "conv2d1" else pol.enc(x)       # conv1 consumes f32 mel on device too
        if FLIP_K and name != "conv2d1":
            xin, w = xin.flip(1), w.flip(1)
        x = _gelu(Fn.conv2d(xin, w, b, stride=2, padding=1))
    x = pol.enc(x)
    n, c, f, t = x.shape                                    # [chunks, 480, 16, t']
    x = x.permute(0, 3, 1, 2).reshape(n, t, c * f)          # idx = c*16 + f   (:423-424)
    x = _mm(x, _w(sd, "audio_tower.conv_out.weight"))       # no bias (:261)
    x = x + sinusoid_pe(t, cfg.d_model)[None]               # PE restarts per chunk (:439)
    valid = [tokens_for_chunk(c_) for c_ in clens]
    return torch.cat([x[i, :v] for i, v in enumerate(valid)], dim=0), clens


def _layer_norm(x, sd, prefix, eps):
    return Fn.layer_norm(x, (x.shape[-1],), _w(sd, prefix + ".weight"), _w(sd, prefix + ".bias"), eps)


def _window_attention(q, k, v, wins, heads, pol: P.Policy):
    """Block-diagonal attention == additive -1e9 mask of AudioEncoder.swift:337-357."""
    T, D = q.shape
    hd = D // heads
    scale = 1.0 / math.sqrt(hd)
    out = torch.empty_like(q)
    s = 0
    for wlen in wins:
        e = s + wlen
        qh = q[s:e].reshape(wlen, heads, hd).transpose(0, 1)
        kh = k[s:e].reshape(wlen, heads, hd).transpose(0, 1)
        vh = v[s:e].reshape(wlen, heads, hd).transpose(0, 1)
        sc = (qh @ kh.transpose(1, 2)) * scale
        pr = torch.softmax(sc, dim=-1)
        o = pol.enc(pr) @ vh
        out[s:e] = o.transpose(0, 1).reshape(wlen, D)
        s = e
    assert s == T, (s, T)
    return out


def encode(mel, sd, cfg: AudioEncoderConfig, pol: P.Policy = P.REFERENCE, return_hidden=False):
    """[n_mels, T] float32 log-mel -> [tokens, output_dim] float32 audio embeddings."""
    mel = torch.as_tensor(mel, dtype=torch.float32)
    x, _ = conv_stem(mel, sd, cfg, pol)
    wins = window_lengths(mel.shape[1], cfg)
    assert sum(wins) == x.shape[0], (wins, x.shape)
    for i in range(cfg.layers):
        p = f"audio_tower.layers.{i}"
        h = pol.enc(_layer_norm(x, sd, p + ".self_attn_layer_norm", cfg.ln_eps))
        q = pol.enc(_linear(h, sd, p + ".self_attn.q_proj"))
        k = pol.enc(_linear(h, sd, p + ".self_attn.k_proj"))
        v = pol.enc(_linear(h, sd, p + ".self_attn.v_proj"))
        a = pol.enc(_window_attention(q, k, v, wins, cfg.heads, pol))
        x = x + _linear(a, sd, p + ".self_attn.out_proj")
        h = pol.enc(_layer_norm(x, sd, p + ".final_layer_norm", cfg.ln_eps))
        h = pol.enc(_gelu(_linear(h, sd, p + ".fc1")))
        x = x + _linear(h, sd, p + ".fc2")
    hidden = x
    h = pol.enc(_layer_norm(x, sd, "audio_tower.ln_post", cfg.ln_eps))
    h = pol.enc(_gelu(_linear(h, sd, "audio_tower.proj1")))
    out = _linear(h, sd, "audio_tower.proj2")
    if return_hidden:
        return out, hidden
    return out

"""CPU restatement of `Qwen3ForcedAligner` (word-level timestamps for an audio + text pair).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows, in the reference (ivan-digital/qwen3-asr-swift):
  Sources/Qwen3ASR/TextPreprocessing.swift:48-93   prepareForAlignment (timestamp slots around words)
  Sources/Qwen3ASR/TextPreprocessing.swift:165-335 default path: whitespace split, per-Han break, cleanToken
  Sources/Qwen3ASR/TimestampCorrection.swift:15-145 LIS + interpolation monotonicity fix-up (integer)
  Sources/Qwen3ASR/ForcedAligner.swift:226-331     align: mel -> encoder -> one decoder pass -> classify head
  Sources/Qwen3ASR/ForcedAligner.swift:97-215      alignLong: trailing-plateau detection and re-alignment
  Sources/Qwen3ASR/ForcedAligner.swift:337-378     buildInputIds (chat template without <asr_text>)

Not restatable: the Japanese / Korean / Thai / Lao / Khmer / Burmese / Tibetan word splitters call Apple's
NaturalLanguage `NLTokenizer` (closed source, TextPreprocessing.swift:129-163).  For those languages the caller
passes pre-split words (`words=`); everything after the split is identical.

Pinned by: the reference's own unit tests for this logic (tests/golden/kat_aligner.json, copied cases from
Tests/Qwen3ASRTests/ForcedAlignerTests.swift:14-47,140-260,441-510) and, for the single-pass decoder + linear
head arithmetic, by transformers' independent `Qwen3ASRForTokenClassification`-style forward
(tests/golden/hf_tiny_aligner.npz).
"""
import unicodedata
import numpy as np
import torch
from . import decoder as dec_mod
from . import precision as P
from .config import TokenIds, TOKENS

TIMESTAMP_TOKEN_ID = 151705          # <|timestamp|>  Qwen3ASR.swift:62
CLASSIFY_NUM = 5000                  # Configuration.swift:132
SEGMENT_TIME = 0.08                  # seconds per class, Configuration.swift:133

NL_LANGUAGES = ("japanese", "korean", "thai", "lao", "khmer", "burmese", "myanmar", "tibetan")
NL_CODES = ("ja", "ko", "th", "lo", "km", "my", "bo")

# Unicode White_Space (what Swift's Character.isWhitespace tests, scalar level)
_WS = {0x09, 0x0A, 0x0B, 0x0C, 0x0D, 0x20, 0x85, 0xA0, 0x1680, 0x2028, 0x2029, 0x202F, 0x205F, 0x3000} | set(range(0x2000, 0x200B))


def is_whitespace(ch):
    return ord(ch) in _WS


def is_kept_scalar(ch):
    """TextPreprocessing.swift:300-316: Letters, Numbers, Marks and the ASCII apostrophe."""
    return ch == "'" or unicodedata.category(ch)[0] in "LNM"


def clean_token(token):
    """TextPreprocessing.swift:290-298."""
    return "".join(c for c in token if is_kept_scalar(c))


def is_han(ch):
    """TextPreprocessing.swift:322-332."""
    v = ord(ch)
    return (0x4E00 <= v <= 0x9FFF or 0x3400 <= v <= 0x4DBF or 0x20000 <= v <= 0x2A6DF or 0x2A700 <= v <= 0x2B73F
            or 0x2B740 <= v <= 0x2B81F or 0x2B820 <= v <= 0x2CEAF or 0xF900 <= v <= 0xFAFF)


def needs_nl_tokenizer(language):
    """TextPreprocessing.swift:103-129: the languages dispatched to NLTokenizer."""
    lang = language.lower()
    return any(n in lang for n in NL_LANGUAGES) or lang in NL_CODES


def _pairs_for_segment(seg):
    """TextPreprocessing.swift:207-263 -> list of [surface, cleaned]."""
    if not any(is_han(c) for c in seg):
        cleaned = clean_token(seg)
        return [[seg, cleaned]] if cleaned else []
    pairs, buf = [], ""

    def flush(before_han):
        nonlocal buf
        if not buf:
            return
        cleaned = clean_token(buf)
        if not cleaned:
            if pairs:
                pairs[-1][0] += buf
                buf = ""
            elif not before_han:
                buf = ""
            return
        pairs.append([buf, cleaned])
        buf = ""

    for ch in seg:
        if is_han(ch):
            flush(True)
            if buf:                      # leading pure punctuation waiting for a Han anchor
                pairs.append([buf + ch, ch])
                buf = ""
            else:
                pairs.append([ch, ch])
        else:
            buf += ch
    flush(False)
    return pairs


def split_word_pairs(text, language="English"):
    """TextPreprocessing.swift:103-127,174-199 (default path) -> list of (surface, cleaned)."""
    if needs_nl_tokenizer(language):
        raise NotImplementedError("the reference splits this language with Apple's NLTokenizer; pass words=[...]")
    pairs, seg = [], ""
    segments = []
    for ch in text:
        if is_whitespace(ch):
            if seg:
                segments.append(seg)
            seg = ""
        else:
            seg += ch
    if seg:
        segments.append(seg)
    for s in segments:
        sp = _pairs_for_segment(s)
        if not sp:
            if pairs:
                pairs[-1][0] += s
            continue
        pairs.extend(sp)
    return [(a, b) for a, b in pairs]


def split_words(text, language="English"):
    return [c for _, c in split_word_pairs(text, language)]


def prepare_for_alignment(pairs, encode, ts_id=TIMESTAMP_TOKEN_ID):
    """TextPreprocessing.swift:48-93.  `pairs` = [(surface, cleaned)], `encode` = tokenizer.encode.
    -> (token ids, timestamp positions, surface words)."""
    ids, ts_pos, words = [], [], []
    for surface, cleaned in pairs:
        toks = list(encode(cleaned))
        if not toks:
            if words:
                words[-1] += surface
            continue
        ts_pos.append(len(ids))
        ids.append(ts_id)
        ids.extend(toks)
        ts_pos.append(len(ids))
        ids.append(ts_id)
        words.append(surface)
    return ids, ts_pos, words


def lis_positions(arr):
    """TimestampCorrection.swift:102-144: positions of one longest strictly increasing subsequence."""
    n = len(arr)
    if n == 0:
        return []
    tails, tail_idx, parent = [], [], [-1] * n
    for i in range(n):
        lo, hi = 0, len(tails)
        while lo < hi:
            mid = (lo + hi) // 2
            if tails[mid] < arr[i]:
                lo = mid + 1
            else:
                hi = mid
        if lo == len(tails):
            tails.append(arr[i])
            tail_idx.append(i)
        else:
            tails[lo] = arr[i]
            tail_idx[lo] = i
        parent[i] = tail_idx[lo - 1] if lo > 0 else -1
    pos, idx = [], tail_idx[-1]
    while idx != -1:
        pos.append(idx)
        idx = parent[idx]
    return pos[::-1]


def enforce_monotonicity(raw):
    """TimestampCorrection.swift:15-99 (float32 interpolation weight, truncation toward zero)."""
    raw = [int(v) for v in raw]
    if len(raw) <= 1:
        return raw
    lis = lis_positions(raw)
    lis_set = set(lis)
    anchors = [(p, raw[p]) for p in lis]
    if len(anchors) == len(raw):
        return raw
    out = list(raw)
    a_idx = 0
    for i in range(len(out)):
        if i in lis_set:
            for k, (p, _) in enumerate(anchors):
                if p == i:
                    a_idx = k
                    break
            continue
        if a_idx < len(anchors) and anchors[a_idx][0] < i:
            prev = anchors[a_idx]
        elif a_idx > 0:
            prev = anchors[a_idx - 1]
        else:
            prev = None
        nxt_i = a_idx
        while nxt_i < len(anchors) and anchors[nxt_i][0] <= i:
            nxt_i += 1
        nxt = anchors[nxt_i] if nxt_i < len(anchors) else None
        if prev is not None and nxt is not None:
            if nxt[0] - prev[0] <= 3:
                out[i] = prev[1] if (i - prev[0]) <= (nxt[0] - i) else nxt[1]
            else:
                t = np.float32(i - prev[0]) / np.float32(nxt[0] - prev[0])
                out[i] = prev[1] + int(np.float32(t * np.float32(nxt[1] - prev[1])))
        elif prev is not None:
            out[i] = prev[1]
        elif nxt is not None:
            out[i] = nxt[1]
    for i in range(1, len(out)):
        if out[i] < out[i - 1]:
            out[i] = out[i - 1]
    return out


def find_trailing_plateau_start(starts, tolerance, min_size):
    """ForcedAligner.swift:196-215 on the words' start times (float32)."""
    n = len(starts)
    if n <= min_size:
        return n
    s = np.asarray(starts, dtype=np.float32)
    plateau = n
    for i in range(n - 1, 0, -1):
        if abs(np.float32(s[i] - s[i - 1])) < np.float32(tolerance):
            plateau = i - 1
        else:
            break
    return plateau if (n - plateau) >= min_size else n


def build_input_ids(slotted_ids, n_audio, tok: TokenIds = TOKENS):
    """ForcedAligner.swift:337-378 -> (ids, index of the first audio pad)."""
    ids = [tok.im_start, tok.system, tok.newline, tok.im_end, tok.newline,
           tok.im_start, tok.user, tok.newline, tok.audio_start]
    a0 = len(ids)
    ids += [tok.audio_pad] * n_audio
    ids += [tok.audio_end, tok.im_end, tok.newline, tok.im_start, tok.assistant, tok.newline]
    ids += list(slotted_ids)
    return ids, a0


def classify_logits(audio_embeds, slotted_ids, ts_positions, W: dec_mod.Weights, cfg, pol: P.Policy,
                    tok: TokenIds = TOKENS):
    """ForcedAligner.swift:258-299: one decoder pass (no cache), classify head at the timestamp slots.
    -> logits [n_ts, classify_num] in the decoder dtype."""
    n_audio = audio_embeds.shape[0]
    ids, a0 = build_input_ids(slotted_ids, n_audio, tok)
    x = dec_mod.embed(ids, W).clone()
    x[a0:a0 + n_audio] = pol.dec(torch.as_tensor(audio_embeds, dtype=torch.float32))
    h = dec_mod.forward(x, W, cfg, dec_mod.DecoderState(cfg.layers), pol)
    start = len(ids) - len(slotted_ids)
    rows = h[[start + p for p in ts_positions]]
    return pol.dec(rows @ W("lm_head.weight").T + W("lm_head.bias"))


def words_from_indices(corrected, words, segment_time=SEGMENT_TIME):
    """ForcedAligner.swift:311-330 -> [(word, start, end)] with float32 times."""
    out = []
    for w, word in enumerate(words):
        if 2 * w + 1 >= len(corrected):
            break
        s = np.float32(corrected[2 * w]) * np.float32(segment_time)
        e = np.float32(corrected[2 * w + 1]) * np.float32(segment_time)
        out.append((word, float(s), float(max(e, s))))
    return out


def align(model, pcm, pairs, encode, ts_id=TIMESTAMP_TOKEN_ID):
    """ForcedAligner.swift:226-331 with `model` an oracle.pipeline.OracleModel holding aligner weights.
    -> (aligned words, raw indices)."""
    with torch.no_grad():
        emb = model.encode(model.mel(np.asarray(pcm, dtype=np.float32)))
        ids, ts_pos, words = prepare_for_alignment(pairs, encode, ts_id)
        if not words:
            return [], []
        logits = classify_logits(emb, ids, ts_pos, model.W, model.text_cfg, model.policy, model.tok)
        raw = [dec_mod.argmax_lowest(r) for r in logits]
    return words_from_indices(enforce_monotonicity(raw), words), raw


def align_long(align_fn, pcm, text, sample_rate=16000, bypass_s=240.0, min_chunk_s=5.0, plateau_tol=0.1, plateau_min=5):
    """ForcedAligner.swift:97-180 around a single-pass `align_fn(audio, text) -> [(word, start, end)]`.
    -> (words, passes)."""
    f32 = np.float32
    out, audio, rem_text, offset, passes = [], np.asarray(pcm, dtype=np.float32), text, f32(0), 1
    while len(audio) and rem_text:
        duration = f32(len(audio)) / f32(sample_rate)
        aligned = align_fn(audio, rem_text)
        if not aligned:
            break

        def shifted(ws):
            return [(w, float(f32(s) + offset), float(f32(e) + offset)) for w, s, e in ws] if offset != 0 else list(ws)
        if duration <= bypass_s or len(aligned) < plateau_min * 2:
            out += shifted(aligned)
            break
        p = find_trailing_plateau_start([s for _, s, _ in aligned], plateau_tol, plateau_min)
        if p == len(aligned):
            out += shifted(aligned)
            break
        if p == 0:
            break                                   # (the reference would trap on `reliable.last!`)
        split_time = f32(aligned[p - 1][2])
        out += shifted(aligned[:p])
        split_sample = int(split_time * f32(sample_rate))
        if split_sample >= len(audio):
            break
        nxt = audio[split_sample:]
        if f32(len(nxt)) / f32(sample_rate) < min_chunk_s:
            break
        words_all = [w for w in rem_text.split(" ") if w]
        if p >= len(words_all):
            break
        audio, rem_text, offset = nxt, " ".join(words_all[p:]), f32(offset + split_time)
        passes += 1
        if passes > 10:
            break
    return out, passes

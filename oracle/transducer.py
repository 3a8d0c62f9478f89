"""Transducer greedy loops, vocabulary decode and the 160 ms streaming session of the Parakeet / Nemotron models: CPU restatement
(BASELINE configs[4], restatable slice -- integer / host logic only).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The networks themselves (FastConformer encoder, LSTM prediction network, joint) are opaque CoreML bundles in the reference; here they
are callables supplied by the caller:
    decoder(token)            -> None      advance the prediction network with `token` (its state lives with the caller)
    joint(t)                  -> logits    token logits [vocab + 1] (TDT: (token_logits, duration_logits)) for encoder frame t and the
                                           current prediction-network output, float32 (the CoreML outputs are float16, widened)
    encoder(mel [128, frames]) -> n_frames  run one streaming chunk, return the number of valid output frames

Follows:
  * `Sources/ParakeetASR/TDTGreedyDecoder.swift:45-143` (loop), `:149-172` (logSoftmax), `:183-205` (argmax)   -> tdt_greedy
  * `Sources/NemotronStreamingASR/RNNTGreedyDecoder.swift:38-90`                                               -> rnnt_greedy
  * `Sources/ParakeetStreamingASR/RNNTGreedyDecoder.swift:58-126` (EOU token ends the utterance)               -> rnnt_greedy(eou_id=...)
  * `Sources/ParakeetASR/Vocabulary.swift:42-96`                                                               -> ParakeetVocabulary
  * `Sources/NemotronStreamingASR/Vocabulary.swift:31-77` (= ParakeetStreamingASR/Vocabulary.swift)            -> StreamVocabulary
  * `Sources/NemotronStreamingASR/StreamingSession.swift:110-212` (pushAudio / finalize / processChunk)        -> NemotronSession
  * configs: `ParakeetASR/Configuration.swift:36-71`, `NemotronStreamingASR/Configuration.swift:35-61`,
    `ParakeetStreamingASR/Configuration.swift:29-55`
"""
import math
from dataclasses import dataclass, field

import numpy as np

F32 = np.float32


@dataclass(frozen=True)
class ParakeetConfig:                     # ParakeetConfig.default
    num_mel_bins: int = 128
    sample_rate: int = 16000
    n_fft: int = 512
    hop_length: int = 160
    win_length: int = 400
    pre_emphasis: float = 0.97
    encoder_hidden: int = 1024
    encoder_layers: int = 24
    subsampling_factor: int = 8
    decoder_hidden: int = 640
    decoder_layers: int = 2
    vocab_size: int = 8192
    blank_token_id: int = 8192
    num_duration_bins: int = 5
    duration_bins: tuple = (0, 1, 2, 3, 4)
    first_text_token_id: int = 274        # TDTGreedyDecoder.swift:91-94


@dataclass(frozen=True)
class NemotronStreamingConfig:            # NemotronStreamingConfig.default (the 160 ms bundle)
    num_mel_bins: int = 128
    sample_rate: int = 16000
    n_fft: int = 512
    hop_length: int = 160
    win_length: int = 400
    pre_emphasis: float = 0.97
    encoder_hidden: int = 1024
    encoder_layers: int = 24
    subsampling_factor: int = 8
    attention_context: int = 70
    conv_cache_size: int = 8
    decoder_hidden: int = 640
    decoder_layers: int = 2
    vocab_size: int = 1024
    blank_token_id: int = 1024
    chunk_ms: int = 160
    chunk_size: int = 2
    right_context: int = 1
    mel_frames: int = 17
    pre_cache_size: int = 16
    output_frames: int = 2


@dataclass(frozen=True)
class ParakeetEOUConfig:                  # ParakeetEOUConfig.default (320 ms chunks)
    num_mel_bins: int = 128
    sample_rate: int = 16000
    n_fft: int = 512
    hop_length: int = 160
    win_length: int = 400
    pre_emphasis: float = 0.97
    encoder_hidden: int = 512
    encoder_layers: int = 17
    subsampling_factor: int = 8
    attention_context: int = 70
    conv_cache_size: int = 8
    decoder_hidden: int = 640
    decoder_layers: int = 1
    vocab_size: int = 1026
    blank_token_id: int = 1026
    eou_token_id: int = 1024
    eob_token_id: int = 1025
    chunk_ms: int = 320
    mel_frames: int = 33
    pre_cache_size: int = 9
    output_frames: int = 4


MAX_SYMBOLS_PER_STEP = 10                 # RNNTGreedyDecoder.swift:35


def argmax_first(v):
    """vDSP_maxvi / the scalar `>` scan: index of the FIRST maximum."""
    return int(np.argmax(np.asarray(v, dtype=F32)))


def log_softmax_at(logits, token_id):
    """logit[id] - (log(sum(exp(logits - max))) + max), Float32 throughout (TDTGreedyDecoder.swift:149-172)."""
    x = np.asarray(logits, dtype=F32)
    mx = x.max()
    s = np.exp((x - mx).astype(F32), dtype=F32).sum(dtype=F32)
    return F32(x[token_id] - (np.log(s, dtype=F32) + mx))


def confidence(log_probs):
    """min(1, exp(mean log-prob)), 0 when nothing was emitted (TDTGreedyDecoder.swift:135-141)."""
    if len(log_probs) == 0:
        return F32(0.0)
    acc = F32(0.0)
    for lp in log_probs:                                  # reduce(0, +) in Float32, left to right
        acc = F32(acc + F32(lp))
    return F32(min(F32(1.0), np.exp(acc / F32(len(log_probs)), dtype=F32)))


def tdt_greedy(encoded_length, decoder, joint, cfg=ParakeetConfig()):
    """TDTGreedyDecoder.decode: the prediction network is primed with the blank id; blank advances one frame; a non-blank token
    advances max(duration, 1) frames and feeds the network; ids below 274 (language / control pieces) are fed but not reported."""
    tokens, log_probs = [], []
    decoder(cfg.blank_token_id)
    t = 0
    while t < encoded_length:
        token_logits, duration_logits = joint(t)
        tok = argmax_first(token_logits[:cfg.vocab_size + 1])
        if tok == cfg.blank_token_id:
            t += 1
            continue
        if tok >= cfg.first_text_token_id:
            tokens.append(tok)
            log_probs.append(log_softmax_at(token_logits[:cfg.vocab_size + 1], tok))
        dur = cfg.duration_bins[argmax_first(duration_logits[:cfg.num_duration_bins])]
        t += max(dur, 1)
        decoder(tok)
    return tokens, log_probs, confidence(log_probs)


def rnnt_greedy(encoded_length, decoder, joint, vocab_size, blank_id, eou_id=None, frame_offset=0):
    """RNNTGreedyDecoder.decode on frames [frame_offset, frame_offset + encoded_length): at most 10 symbols per frame, blank moves on;
    the EOU variant stops the whole decode at the EOU id (not reported, not fed).  The prediction-network state persists across calls
    (the caller primes it once per session)."""
    tokens, log_probs, eou = [], [], False
    total = vocab_size + 1
    for i in range(encoded_length):
        t = i + frame_offset
        for _ in range(MAX_SYMBOLS_PER_STEP):
            logits = joint(t)[:total]
            tok = argmax_first(logits)
            if tok == blank_id:
                break
            if eou_id is not None and tok == eou_id:
                eou = True
                break
            tokens.append(tok)
            log_probs.append(log_softmax_at(logits, tok))
            decoder(tok)
        if eou:
            break
    return tokens, log_probs, eou


def _trim_spaces(s):
    return s.strip(" \t")                                  # CharacterSet.whitespaces = Unicode Zs + tab; pieces only ever hold U+0020


@dataclass
class WordConfidence:
    word: str
    confidence: float


def _word_conf(lps):
    acc = F32(0.0)
    for lp in lps:
        acc = F32(acc + F32(lp))
    return float(min(F32(1.0), np.exp(acc / F32(len(lps)), dtype=F32)))


class ParakeetVocabulary:
    """ParakeetASR/Vocabulary.swift: unknown ids are skipped, U+2581 -> space, the joined text trimmed."""

    def __init__(self, id_to_token):
        self.t = dict(id_to_token)

    def decode(self, ids):
        return _trim_spaces("".join(self.t[i].replace("▁", " ") for i in ids if i in self.t))

    def decode_words(self, ids, log_probs):
        if len(ids) != len(log_probs):
            return [WordConfidence(self.decode(ids), 0.0)]
        words, cur, lps = [], "", []
        for i, tid in enumerate(ids):
            if tid not in self.t:
                continue
            tok = self.t[tid]
            if tok.startswith("▁") and cur != "":
                words.append(WordConfidence(cur, _word_conf(lps)))
                cur, lps = "", []
            cur += tok.replace("▁", "")
            lps.append(log_probs[i])
        if cur != "":
            words.append(WordConfidence(cur, _word_conf(lps)))
        return words


class StreamVocabulary:
    """NemotronStreamingASR/Vocabulary.swift (= ParakeetStreamingASR): pieces concatenated first, then U+2581 -> space, trimmed;
    decodeWords returns [] on a length mismatch and drops words that trim to nothing."""

    def __init__(self, id_to_token):
        self.t = dict(id_to_token)

    def decode(self, ids):
        return _trim_spaces("".join(self.t[i] for i in ids if i in self.t).replace("▁", " "))

    def decode_words(self, ids, log_probs):
        if len(ids) != len(log_probs):
            return []
        words, cur, lps = [], "", []

        def flush():
            w = _trim_spaces(cur.replace("▁", " "))
            if w != "":
                words.append(WordConfidence(w, _word_conf(lps)))

        for i, tid in enumerate(ids):
            if tid not in self.t:
                continue
            tok = self.t[tid]
            if tok.startswith("▁") and cur != "":
                flush()
                cur, lps = tok, [log_probs[i]]
            else:
                cur += tok
                lps.append(log_probs[i])
        if cur != "":
            flush()
        return words


@dataclass
class Partial:
    text: str
    is_final: bool
    confidence: float
    segment_index: int = 0


class NemotronSession:
    """StreamingSession of the 160 ms Nemotron streamer: samples accumulate; every time 17 mel frames' worth (2720 samples) is buffered
    one chunk is cut and the buffer advances by outputFrames x subsampling x hop = 2560 samples (the 160-sample overlap is the encoder's
    right context); each chunk -> extractRaw mel fitted to 17 frames -> encoder -> RNNT greedy over min(outputFrames, valid) frames.
    `mel_fn(chunk) -> ([128, frames], melLength)`, `encoder(mel17) -> valid frames`; decoder / joint as above, primed once at creation."""

    def __init__(self, mel_fn, encoder, decoder, joint, vocab, cfg=NemotronStreamingConfig()):
        self.cfg, self.mel_fn, self.encoder, self.decoder, self.joint, self.vocab = cfg, mel_fn, encoder, decoder, joint, vocab
        self.buf = np.zeros(0, dtype=F32)
        self.tokens, self.log_probs = [], []
        self.chunks = []                                       # the chunks handed to the mel (test visibility)
        decoder(cfg.blank_token_id)                            # StreamingSession.swift:93-99

    @property
    def samples_per_chunk(self):
        return self.cfg.mel_frames * self.cfg.hop_length

    @property
    def shift_samples(self):
        return self.cfg.output_frames * self.cfg.subsampling_factor * self.cfg.hop_length

    def push_audio(self, samples):                             # :110-131
        self.buf = np.concatenate([self.buf, np.asarray(samples, dtype=F32)])
        out = []
        while self.buf.shape[0] >= self.samples_per_chunk:
            chunk = self.buf[:self.samples_per_chunk].copy()
            self.buf = self.buf[min(self.shift_samples, self.buf.shape[0]):]
            p = self._process(chunk)
            if p is not None:
                out.append(p)
        return out

    def finalize(self):                                        # :133-158
        if self.buf.shape[0] > 0:
            pad = max(0, self.samples_per_chunk - self.buf.shape[0])
            chunk = np.concatenate([self.buf, np.zeros(pad, dtype=F32)])[:self.samples_per_chunk]
            self.buf = np.zeros(0, dtype=F32)
            self._process(chunk)
        if not self.tokens:
            return []
        return [Partial(self.vocab.decode(self.tokens), True, float(confidence(self.log_probs)))]

    def _process(self, chunk):                                 # :160-231
        self.chunks.append(chunk)
        mel, mel_len = self.mel_fn(chunk)
        if mel_len <= 0:
            return None
        from oracle.nemo_mel import fit_frames
        valid = self.encoder(fit_frames(mel, self.cfg.mel_frames))
        n = min(self.cfg.output_frames, valid)
        if n <= 0:
            return None
        toks, lps, _ = rnnt_greedy(n, self.decoder, self.joint, self.cfg.vocab_size, self.cfg.blank_token_id)
        self.tokens += toks
        self.log_probs += lps
        text = self.vocab.decode(self.tokens)
        if text == "":
            return None
        return Partial(text, False, float(confidence(self.log_probs)))

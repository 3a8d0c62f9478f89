"""End-to-end `Qwen3ASRModel.transcribe` on CPU (B = 1, sequential, like the reference).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows Sources/Qwen3ASR/Qwen3ASR.swift:131-164 (mel -> encoder -> generateText).
"""
import numpy as np
import torch
from . import mel as mel_mod
from . import encoder as enc_mod
from . import decoder as dec_mod
from . import precision as P
from .config import AudioEncoderConfig, TextDecoderConfig, TokenIds, TOKENS, AUDIO_SMALL, TEXT_SMALL


class OracleModel:
    def __init__(self, sd, audio_cfg: AudioEncoderConfig = AUDIO_SMALL,
                 text_cfg: TextDecoderConfig = TEXT_SMALL, tok: TokenIds = TOKENS,
                 policy: P.Policy = P.REFERENCE, fft_scale=2.0):
        self.sd = sd
        self.W = dec_mod.Weights(sd)
        self.audio_cfg, self.text_cfg, self.tok = audio_cfg, text_cfg, tok
        self.policy = policy
        self.fft_scale = fft_scale
        self._fb = mel_mod.mel_filterbank()
        self._win = mel_mod.hann_window()

    def mel(self, pcm):
        return mel_mod.log_mel(pcm, fft_scale=self.fft_scale, _fb=self._fb, _win=self._win)

    def encode(self, mel):
        return enc_mod.encode(mel, self.W, self.audio_cfg, self.policy)

    def transcribe_tokens(self, pcm, max_tokens=448, ignore_eos=False, context_ids=None,
                          language_ids=None, return_logits=False):
        with torch.no_grad():
            emb = self.encode(self.mel(np.asarray(pcm, dtype=np.float32)))
            return dec_mod.greedy(emb, self.W, self.text_cfg, self.policy, self.tok, max_tokens,
                                  context_ids, language_ids, ignore_eos, return_logits)

"""Rounding policy: where the compared implementations round activations to bfloat16.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference executes on MLX: tensors carry a dtype and every op rounds its result to that
dtype while computing internally in float32.  In the reference the mel features are float32
(AudioPreprocessing.swift:315), so the whole audio encoder runs in float32 (MLX promotes
f32 x bf16 -> f32), and the cast to the decoder dtype happens once, at the splice
(Qwen3ASR.swift:240).  The text decoder's tensors are bfloat16 (dequantised embeddings are
bf16, PreQuantizedEmbedding.swift:28-29,38-40), so each decoder op output is rounded to bf16.

Policies
  REFERENCE : encoder f32 everywhere; decoder bf16 at every op boundary.   (what MLX does)
  DEVICE    : decoder identical to REFERENCE; encoder additionally rounds the *inputs of
              matrix products* (and the stored conv activations) to bf16, because the HIP
              path feeds MFMA bf16 operands.  This is the only deliberate numerical
              deviation of the MI355X path and its effect is measured by the tests
              (encoder output tolerance vs REFERENCE is stated there).
  F32       : no rounding anywhere (structure checks against the HF implementation).
"""
from dataclasses import dataclass
import torch


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


@dataclass(frozen=True)
class Policy:
    name: str
    enc_gemm_in_bf16: bool      # round encoder matmul/conv inputs + stored activations
    dec_bf16: bool              # round decoder op outputs

    def enc(self, x):
        return bf16_round(x) if self.enc_gemm_in_bf16 else x

    def dec(self, x):
        return bf16_round(x) if self.dec_bf16 else x


REFERENCE = Policy("reference", enc_gemm_in_bf16=False, dec_bf16=True)
DEVICE = Policy("device", enc_gemm_in_bf16=True, dec_bf16=True)
F32 = Policy("f32", enc_gemm_in_bf16=False, dec_bf16=False)

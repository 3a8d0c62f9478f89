#!/usr/bin/env python3
"""Structural golden for the Omnilingual (wav2vec2-CTC) oracle: outputs of the independently written
`transformers.models.wav2vec2` (v5.x, installed in the build container only; never shipped to the GPU box) on seeded random
weights at a tiny geometry, with the state dict renamed to the reference's fairseq2 tensor names
(MLX/OmnilingualMLXWeightLoader.swift:40-135).  Configuration = fairseq2's wav2vec2 as the reference builds it: layer-norm
feature extractor with conv bias, stable-layer-norm (pre-norm) encoder, weight-normed grouped conv positional encoder.

Run from the repo root:  python tests/golden/make_hf_goldens_w2v.py   -> tests/golden/hf_tiny_w2v.npz
"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import omnilingual as O  # noqa: E402


def rename(k):
    k = k.replace("wav2vec2.feature_extractor.conv_layers.", "encoder_frontend.feature_extractor.layers.")
    k = k.replace("wav2vec2.feature_projection.layer_norm", "encoder_frontend.post_extract_layer_norm")
    k = k.replace("wav2vec2.feature_projection.projection", "encoder_frontend.model_dim_proj")
    k = k.replace("wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0", "encoder_frontend.pos_encoder.conv.weight_g")
    k = k.replace("wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1", "encoder_frontend.pos_encoder.conv.weight_v")
    k = k.replace("wav2vec2.encoder.pos_conv_embed.conv.bias", "encoder_frontend.pos_encoder.conv.bias")
    k = k.replace("wav2vec2.encoder.layers.", "encoder.layers.")
    k = k.replace(".attention.out_proj", ".self_attn.output_proj").replace(".attention.", ".self_attn.")
    k = k.replace(".feed_forward.intermediate_dense", ".ffn.inner_proj").replace(".feed_forward.output_dense", ".ffn.output_proj")
    k = k.replace(".final_layer_norm", ".ffn_layer_norm")
    if k.startswith("encoder.layers.") and k.split(".")[3] == "layer_norm":
        k = k.replace(".layer_norm.", ".self_attn_layer_norm.")
    k = k.replace("wav2vec2.encoder.layer_norm", "encoder.layer_norm")
    k = k.replace("lm_head.", "final_proj.")
    return k


def main():
    from transformers import Wav2Vec2Config, Wav2Vec2ForCTC
    c = O.OMNI_TINY
    cfg = Wav2Vec2Config(vocab_size=c.vocab, hidden_size=c.model_dim, num_hidden_layers=c.layers, num_attention_heads=c.heads,
                         intermediate_size=c.ffn_dim, feat_extract_norm="layer", feat_extract_activation="gelu",
                         conv_dim=(c.feature_dim,) * 7, conv_stride=O.STRIDES, conv_kernel=O.KERNELS, conv_bias=True,
                         num_conv_pos_embeddings=c.pos_kernel, num_conv_pos_embedding_groups=c.pos_groups,
                         do_stable_layer_norm=True, hidden_act="gelu", layer_norm_eps=c.ln_eps, attention_dropout=0.0,
                         hidden_dropout=0.0, feat_proj_dropout=0.0, final_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0)
    torch.manual_seed(0)
    m = Wav2Vec2ForCTC(cfg).eval()
    with torch.no_grad():                                    # non-trivial norm gains / biases so every affine path is exercised
        for n, p in m.named_parameters():
            if "layer_norm" in n:
                p.add_(torch.randn_like(p) * 0.1)
            elif n.endswith(".bias"):
                p.add_(torch.randn_like(p) * 0.05)
    sd = {rename(k): v.detach().clone() for k, v in m.state_dict().items() if "masked_spec_embed" not in k}
    out = {"sd/" + k: v.numpy() for k, v in sd.items()}
    rng = np.random.default_rng(7)
    for name, n in (("a", 4000), ("b", 7321), ("c", 401)):
        wave = (0.3 * np.sin(2 * np.pi * 220 * np.arange(n) / 16000) + 0.1 * rng.standard_normal(n) + 0.05).astype(np.float32)
        norm = O.layer_normalize(wave)
        with torch.no_grad():
            r = m(torch.from_numpy(norm)[None], output_hidden_states=True)
        out[f"wave/{name}"] = wave
        out[f"logits/{name}"] = r.logits[0].numpy()
        out[f"frontend/{name}"] = r.hidden_states[0][0].numpy()          # after the positional encoder
        assert r.logits.shape[1] == O.output_length(n), (r.logits.shape, O.output_length(n))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hf_tiny_w2v.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()

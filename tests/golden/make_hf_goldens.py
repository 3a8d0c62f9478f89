"""Generate encoder/decoder golden vectors from an independent implementation.

Run in the BUILD container only (needs `transformers.models.qwen3_asr`, v5.15.0 here):
    python tests/golden/make_hf_goldens.py
Writes tests/golden/hf_tiny.npz (inputs + expected outputs, tiny geometry, float32).

Why: the reference (ivan-digital/qwen3-asr-swift) holds no tensor-level golden for this path
and cannot run on Linux.  Its encoder/decoder are stated to match the
HuggingFace Qwen3-ASR architecture (AudioEncoder.swift:202 "Matches HuggingFace weight
structure exactly"); the locally installed `transformers` carries an independently written
implementation of that architecture.  We instantiate it with seeded random weights, run it in
float32, and store inputs/outputs.  tests/test_oracle_hf.py then feeds the SAME tensors to
oracle/ (policy F32) and asserts agreement => the oracle's structure is pinned by something
other than itself.  Nothing from `transformers` ships to the GPU box; only this data does.
"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))

from transformers.models.qwen3_asr import modeling_qwen3_asr as M           # noqa: E402
from transformers.models.qwen3_asr.configuration_qwen3_asr import Qwen3ASRConfig  # noqa: E402
from oracle import config as C                                              # noqa: E402
from qasr import synth                                                      # noqa: E402


def hf_model(a: C.AudioEncoderConfig, t: C.TextDecoderConfig, tok: C.TokenIds):
    cfg = Qwen3ASRConfig(
        audio_config=dict(num_mel_bins=a.n_mels, encoder_layers=a.layers,
                          encoder_attention_heads=a.heads, encoder_ffn_dim=a.ffn_dim,
                          d_model=a.d_model, n_window=a.n_window, output_dim=a.output_dim,
                          n_window_infer=a.n_window_infer,
                          downsample_hidden_size=a.conv_channels),
        text_config=dict(model_type="qwen3", vocab_size=t.vocab, hidden_size=t.hidden,
                         intermediate_size=t.inter, num_hidden_layers=t.layers,
                         num_attention_heads=t.heads, num_key_value_heads=t.kv_heads,
                         head_dim=t.head_dim, max_position_embeddings=65536,
                         rms_norm_eps=t.rms_eps, rope_theta=t.rope_theta,
                         tie_word_embeddings=True),
        audio_token_id=tok.audio_pad,
    )
    cfg._attn_implementation = "eager"
    m = M.Qwen3ASRForConditionalGeneration(cfg).eval().to(torch.float32)
    return m


def load_reference_names(m, sd):
    """Copy a reference-named state dict (qasr.synth) into the HF module tree."""
    hf = {}
    for k, v in sd.items():
        v = v.to(torch.float32)
        if k.startswith("audio_tower.proj1."):
            hf["model.multi_modal_projector.linear_1." + k.split(".")[-1]] = v
        elif k.startswith("audio_tower.proj2."):
            hf["model.multi_modal_projector.linear_2." + k.split(".")[-1]] = v
        elif k.startswith("audio_tower.conv2d") and k.endswith(".weight"):
            hf["model." + k] = v.permute(0, 3, 1, 2).contiguous()     # [o,kh,kw,i] -> [o,i,kh,kw]
        elif k.startswith("audio_tower."):
            hf["model." + k] = v
        elif k.startswith("model."):
            hf["model.language_model." + k[len("model."):]] = v
        else:
            raise KeyError(k)
    missing, unexpected = m.load_state_dict(hf, strict=False)
    missing = [x for x in missing if "lm_head" not in x and "positional_embedding" not in x]
    assert not missing and not unexpected, (missing, unexpected)
    m.tie_weights()


def main():
    a, t, tok = C.AUDIO_TINY, C.TEXT_TINY, C.TOKENS_TINY
    sd = synth.synth_state_dict(a, t, seed=1234, init="stress", dtype=torch.float32)
    m = hf_model(a, t, tok)
    load_reference_names(m, sd)
    out = {}
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for T in (100, 250, 300, 530):
            mel = torch.randn(a.n_mels, T, generator=g) * 0.5
            Tp = ((T + 99) // 100) * 100
            feats = torch.zeros(1, a.n_mels, Tp)
            feats[0, :, :T] = mel
            mask = torch.zeros(1, Tp, dtype=torch.long)
            mask[0, :T] = 1
            res = m.get_audio_features(feats, mask)
            out[f"mel_{T}"] = mel.numpy()
            out[f"enc_hidden_{T}"] = res.last_hidden_state.numpy()
            out[f"enc_out_{T}"] = res.pooler_output.numpy()
        # decoder: prompt with spliced audio features -> prefill logits + greedy ids
        emb = torch.from_numpy(out["enc_out_250"])
        n_audio = emb.shape[0]
        from oracle.decoder import build_prompt
        ids, a0 = build_prompt(n_audio, tok)
        ids_t = torch.tensor([ids])
        lm = m.model.language_model
        x = lm.embed_tokens(ids_t).clone()
        x[0, a0:a0 + n_audio] = emb
        o = lm(inputs_embeds=x, use_cache=True)
        logits = o.last_hidden_state[0, -1] @ lm.embed_tokens.weight.T
        out["dec_prompt_ids"] = np.array(ids, dtype=np.int32)
        out["dec_prefill_logits"] = logits.numpy()
        past = o.past_key_values
        toks, step_logits = [], []
        nxt = int(torch.argmax(logits))
        for _ in range(16):
            toks.append(nxt)
            o = lm(input_ids=torch.tensor([[nxt]]), past_key_values=past, use_cache=True)
            past = o.past_key_values
            lg = o.last_hidden_state[0, -1] @ lm.embed_tokens.weight.T
            step_logits.append(lg.numpy())
            nxt = int(torch.argmax(lg))
        out["dec_greedy_ids"] = np.array(toks, dtype=np.int32)
        out["dec_step_logits"] = np.stack(step_logits)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})
    aligner_golden(m, out)


def aligner_golden(m, enc):
    """Forced-aligner forward (ForcedAligner.swift:258-299): ONE pass over [template + audio + slotted text] with no
    cache, final-normed hidden state of every position, then a Linear(hidden, classes) head -- the computation of
    transformers' Qwen3ASRForTokenClassification (GenericForTokenClassification: `score(hidden_states)`).  Stored:
    ids, audio features, head weights, hidden states and logits at the timestamp slots."""
    from oracle.aligner import build_input_ids
    tok = C.TOKENS_TINY
    ts_id, n_cls = 506, 40
    g = torch.Generator().manual_seed(11)
    slotted, ts_pos = [], []
    for w in range(9):                       # 9 "words" of 1-3 tokens between timestamp slots
        ts_pos.append(len(slotted)); slotted.append(ts_id)
        slotted += [int(v) for v in torch.randint(10, 290, (1 + w % 3,), generator=g)]
        ts_pos.append(len(slotted)); slotted.append(ts_id)
    emb = torch.from_numpy(enc["enc_out_300"])
    n_audio = emb.shape[0]
    ids, a0 = build_input_ids(slotted, n_audio, tok)
    lm = m.model.language_model
    hidden = lm.embed_tokens.weight.shape[1]
    wc = torch.randn(n_cls, hidden, generator=g) * 0.3
    bc = torch.randn(n_cls, generator=g) * 0.1
    with torch.no_grad():
        x = lm.embed_tokens(torch.tensor([ids])).clone()
        x[0, a0:a0 + n_audio] = emb
        h = lm(inputs_embeds=x, use_cache=False).last_hidden_state[0]
        start = len(ids) - len(slotted)
        logits = h[[start + p for p in ts_pos]] @ wc.T + bc
    out = {"ids": np.array(ids, dtype=np.int32), "slotted": np.array(slotted, dtype=np.int32),
           "ts_pos": np.array(ts_pos, dtype=np.int32), "audio": emb.numpy(), "head_w": wc.numpy(), "head_b": bc.numpy(),
           "hidden": h.numpy(), "logits": logits.numpy(), "ts_id": np.int32(ts_id)}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_tiny_aligner.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: getattr(v, "shape", ()) for k, v in out.items()})


if __name__ == "__main__":
    main()

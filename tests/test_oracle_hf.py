"""Oracle encoder/decoder vs goldens generated from the independent HF implementation
(tests/golden/make_hf_goldens.py).  Tolerance: float32 re-association only (1e-4 abs on
O(1) values); greedy ids must be identical."""
import os
import numpy as np
import torch
from conftest import GOLDEN
from oracle import config as C, decoder, encoder, precision as P
from qasr import synth

G = np.load(os.path.join(GOLDEN, "hf_tiny.npz"))
A, T, TOK = C.AUDIO_TINY, C.TEXT_TINY, C.TOKENS_TINY
SD = synth.synth_state_dict(A, T, seed=1234, init="stress", dtype=torch.float32)
W = decoder.Weights(SD)


def test_encoder_matches_hf():
    with torch.no_grad():
        for n in (100, 250, 300, 530):
            out = encoder.encode(G[f"mel_{n}"], W, A, P.F32).numpy()
            assert out.shape == G[f"enc_out_{n}"].shape
            assert np.abs(out - G[f"enc_out_{n}"]).max() < 1e-4


def test_encoder_dense_mask_equals_windows():
    """Block-diagonal windows == the reference's dense additive -1e9 mask (AudioEncoder.swift:337-357)."""
    torch.manual_seed(0)
    wins, heads, D = [26, 26, 17], 2, 64
    n = sum(wins)
    q, k, v = (torch.randn(n, D) for _ in range(3))
    got = encoder._window_attention(q, k, v, wins, heads, P.F32)
    ids = torch.repeat_interleave(torch.arange(len(wins)), torch.tensor(wins))
    mask = torch.where(ids[:, None] == ids[None, :], 0.0, -1e9)
    hd = D // heads
    qh, kh, vh = (x.reshape(n, heads, hd).transpose(0, 1) for x in (q, k, v))
    ref = torch.softmax(qh @ kh.transpose(1, 2) / hd ** 0.5 + mask, -1) @ vh
    assert torch.allclose(got, ref.transpose(0, 1).reshape(n, D), atol=1e-5)


def test_decoder_matches_hf():
    with torch.no_grad():
        emb = torch.from_numpy(G["enc_out_250"])
        ids, _ = decoder.build_prompt(emb.shape[0], TOK)
        assert ids == G["dec_prompt_ids"].tolist()
        toks, logits = decoder.greedy(emb, W, T, P.F32, TOK, max_tokens=17, ignore_eos=True,
                                      return_logits=True)
        assert toks[:16] == G["dec_greedy_ids"].tolist()
        assert np.abs(logits[0].numpy() - G["dec_prefill_logits"]).max() < 1e-4
        assert np.abs(torch.stack(logits[1:]).numpy() - G["dec_step_logits"]).max() < 1e-4


def test_greedy_eos_semantics():
    """EOS is appended, then the loop stops; at most max_tokens tokens (Qwen3ASR.swift:344-389)."""
    with torch.no_grad():
        emb = torch.from_numpy(G["enc_out_100"])
        free = decoder.greedy(emb, W, T, P.F32, TOK, max_tokens=6, ignore_eos=True)
        assert len(free) == 6
        import dataclasses
        tok2 = dataclasses.replace(TOK, im_end=free[2])       # make the 3rd token the EOS id
        cut = decoder.greedy(emb, W, T, P.F32, tok2, max_tokens=6)
        first = free.index(free[2])
        assert cut == free[:first + 1] and cut[-1] == tok2.eos
        assert decoder.greedy(emb, W, T, P.F32, TOK, max_tokens=0) == []


def test_policies_close():
    """bf16 op-boundary rounding (REFERENCE/DEVICE) stays near the f32 structure."""
    sd = synth.synth_state_dict(A, T, seed=5, init="stress")
    Wb = decoder.Weights(sd)
    with torch.no_grad():
        mel = G["mel_300"]
        f = encoder.encode(mel, Wb, A, P.F32)
        r = encoder.encode(mel, Wb, A, P.REFERENCE)
        d = encoder.encode(mel, Wb, A, P.DEVICE)
        assert torch.equal(f, r)                                 # reference encoder is f32
        rel = (d - f).norm() / f.norm()
        assert rel < 2e-2, rel

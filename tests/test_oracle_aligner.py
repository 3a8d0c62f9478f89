"""oracle/aligner.py against (a) the reference's own unit-test cases for the forced-aligner host logic
(tests/golden/kat_aligner.json, from Tests/Qwen3ASRTests/ForcedAlignerTests.swift) and (b) the HF golden of the
single-pass decoder + linear head (tests/golden/hf_tiny_aligner.npz, made by make_hf_goldens.py)."""
import json
import os
import numpy as np
import pytest
import torch
from conftest import GOLDEN
from oracle import aligner as AL, config as C, decoder, precision as P
from qasr import synth

KAT = json.load(open(os.path.join(GOLDEN, "kat_aligner.json"), encoding="utf-8"))


@pytest.mark.parametrize("case", KAT["split_words"], ids=lambda c: c["ref"])
def test_split_words_kat(case):
    pairs = AL.split_word_pairs(case["text"], case["language"])
    assert [c for _, c in pairs] == case["cleaned"]
    if "surface" in case:
        assert [s for s, _ in pairs] == case["surface"]


def test_nl_tokenizer_languages_are_refused():
    for lang in KAT["nl_tokenizer_languages"]["languages"]:
        with pytest.raises(NotImplementedError):
            AL.split_word_pairs("x", lang)


@pytest.mark.parametrize("case", KAT["monotonicity"], ids=lambda c: c["ref"])
def test_monotonicity_kat(case):
    out = AL.enforce_monotonicity(case["input"])
    if "expected" in case:
        assert out == case["expected"]
    assert all(out[i] >= out[i - 1] for i in range(1, len(out)))


def test_lis_kat():
    for case in KAT["lis"]:
        arr = case["input"]
        pos = AL.lis_positions(arr)
        assert len(pos) >= case["min_length"]
        assert all(arr[pos[i - 1]] < arr[pos[i]] for i in range(1, len(pos)))


def test_lis_is_longest_on_random_inputs():
    rng = np.random.default_rng(0)
    for _ in range(200):
        arr = rng.integers(0, 30, size=rng.integers(1, 40)).tolist()
        pos = AL.lis_positions(arr)
        best = [1] * len(arr)                      # O(n^2) reference length
        for i in range(len(arr)):
            for j in range(i):
                if arr[j] < arr[i]:
                    best[i] = max(best[i], best[j] + 1)
        assert len(pos) == max(best)
        assert all(pos[i - 1] < pos[i] and arr[pos[i - 1]] < arr[pos[i]] for i in range(1, len(pos)))
        out = AL.enforce_monotonicity(arr)
        assert len(out) == len(arr) and all(out[i] >= out[i - 1] for i in range(1, len(out)))


def _plateau_starts(c):
    f = np.float32
    return [f(i) * f(0.5) for i in range(c["healthy"])] + [f(c["stuck_start"]) + f(c["drift"]) * f(j) for j in range(c["stuck"])]


@pytest.mark.parametrize("case", KAT["plateau"], ids=lambda c: c["ref"])
def test_plateau_kat(case):
    assert AL.find_trailing_plateau_start(_plateau_starts(case), case["tolerance"], case["min_size"]) == case["expected"]


def test_input_ids_template():
    """ForcedAligner.swift:337-378: same chat template as the ASR prompt, no context, no <asr_text>."""
    ids, a0 = AL.build_input_ids([7, 8, 9], 4)
    assert ids == [151644, 8948, 198, 151645, 198, 151644, 872, 198, 151669] + [151676] * 4 + \
        [151670, 151645, 198, 151644, 77091, 198, 7, 8, 9]
    assert a0 == 9
    asr, _ = decoder.build_prompt(4)
    assert ids[:-3] == asr[:-1]                   # the ASR prompt is this template + <asr_text>


def test_prepare_for_alignment_slots():
    enc = {"Hello": [11, 12], "world": [13], "": []}
    ids, ts, words = AL.prepare_for_alignment([("Hello,", "Hello"), ("??", ""), ("world!", "world")], lambda s: enc[s], ts_id=99)
    assert ids == [99, 11, 12, 99, 99, 13, 99] and ts == [0, 3, 4, 6]
    assert words == ["Hello,??", "world!"]       # an unencodable word's surface rides with the previous word


def test_classify_forward_matches_hf():
    G = np.load(os.path.join(GOLDEN, "hf_tiny_aligner.npz"))
    A, T, TOK = C.AUDIO_TINY, C.TEXT_TINY, C.TOKENS_TINY
    sd = synth.synth_state_dict(A, T, seed=1234, init="stress", dtype=torch.float32)
    sd["lm_head.weight"] = torch.from_numpy(G["head_w"])
    sd["lm_head.bias"] = torch.from_numpy(G["head_b"])
    W = decoder.Weights(sd)
    slotted, ts_pos = G["slotted"].tolist(), G["ts_pos"].tolist()
    ids, _ = AL.build_input_ids(slotted, G["audio"].shape[0], TOK)
    assert ids == G["ids"].tolist()
    with torch.no_grad():
        logits = AL.classify_logits(torch.from_numpy(G["audio"]), slotted, ts_pos, W, T, P.F32, TOK).numpy()
    assert logits.shape == G["logits"].shape
    assert np.abs(logits - G["logits"]).max() < 1e-4
    assert (logits.argmax(1) == G["logits"].argmax(1)).all()


def test_words_from_indices():
    out = AL.words_from_indices([10, 12, 12, 11], ["a", "b"])
    assert out[0] == ("a", pytest.approx(0.8), pytest.approx(0.96))
    assert out[1][1] == out[1][2] == pytest.approx(0.96)      # end clamped up to start (ForcedAligner.swift:326)

"""Forced-aligner host logic through the C ABI (qasr_split_words / qasr_lis_positions / qasr_enforce_monotonicity /
qasr_find_trailing_plateau: pure CPU, no GPU call) against the reference's unit-test cases and the oracle."""
import json
import os
import numpy as np
import pytest
from conftest import GOLDEN
from oracle import aligner as OA
from qasr import aligner as QA, _lib
import ctypes as C

KAT = json.load(open(os.path.join(GOLDEN, "kat_aligner.json"), encoding="utf-8"))


@pytest.mark.parametrize("case", KAT["split_words"], ids=lambda c: c["ref"])
def test_split_words_kat(case):
    pairs = QA.split_word_pairs(case["text"], case["language"])
    assert [c for _, c in pairs] == case["cleaned"]
    if "surface" in case:
        assert [s for s, _ in pairs] == case["surface"]


def test_nl_tokenizer_languages_are_refused():
    for lang in KAT["nl_tokenizer_languages"]["languages"]:
        with pytest.raises(QA.UnsupportedLanguage):
            QA.split_word_pairs("x", lang)
    assert QA.split_words("nothing special", "English") == ["nothing", "special"]


def test_split_words_matches_oracle_on_mixed_text():
    texts = ["", "   ", "...", "— leading dash", "a  b\tc\nd", "it's 3.14, isn't it?", "你好，world。再见!", "«Guillemets» and “quotes”",
             "naïve café déjà-vu", "x y　z", "١٢٣ ٤٥٦", "\"(nested [brackets])\" fin.", "๑๒๓ abc", "Ⅻ ½ ②",
             "!!! 你 ???", "中a文b", "ab，，cd", "égalité (combining marks)", "🙂 emoji word 🙂", "𠀀𠀁 ext-b"]
    for t in texts:
        assert QA.split_word_pairs(t, "English") == [tuple(p) for p in OA.split_word_pairs(t, "English")], t


@pytest.mark.parametrize("case", KAT["monotonicity"], ids=lambda c: c["ref"])
def test_monotonicity_kat(case):
    out = QA.enforce_monotonicity(case["input"])
    if "expected" in case:
        assert out == case["expected"]
    assert all(out[i] >= out[i - 1] for i in range(1, len(out)))


def test_lis_and_monotonicity_match_oracle():
    for case in KAT["lis"]:
        assert QA.lis_positions(case["input"]) == OA.lis_positions(case["input"])
    rng = np.random.default_rng(1)
    for _ in range(300):
        n = int(rng.integers(0, 60))
        kind = rng.integers(0, 3)
        if kind == 0:
            arr = rng.integers(0, 5000, size=n)
        elif kind == 1:                                # mostly increasing with outliers (what the model emits)
            arr = np.sort(rng.integers(0, 5000, size=n))
            for _ in range(n // 6):
                arr[rng.integers(0, n)] = rng.integers(0, 5000)
        else:                                          # healthy prefix + collapsed tail
            arr = np.concatenate([np.sort(rng.integers(0, 3000, size=n)), rng.integers(0, 40, size=n // 2)])
        arr = arr.astype(np.int32).tolist()
        assert QA.lis_positions(arr) == OA.lis_positions(arr)
        assert QA.enforce_monotonicity(arr) == OA.enforce_monotonicity(arr)
    assert QA.enforce_monotonicity([]) == [] and QA.enforce_monotonicity([7]) == [7] and QA.lis_positions([]) == []


def _plateau_starts(c):
    f = np.float32
    return [f(i) * f(0.5) for i in range(c["healthy"])] + [f(c["stuck_start"]) + f(c["drift"]) * f(j) for j in range(c["stuck"])]


@pytest.mark.parametrize("case", KAT["plateau"], ids=lambda c: c["ref"])
def test_plateau_kat(case):
    assert QA.find_trailing_plateau_start(_plateau_starts(case), case["tolerance"], case["min_size"]) == case["expected"]


def test_plateau_matches_oracle():
    rng = np.random.default_rng(2)
    for _ in range(200):
        n = int(rng.integers(0, 40))
        s = np.cumsum(rng.choice([0.0, 0.04, 0.08, 0.5], size=n)).astype(np.float32)
        for tol, m in ((0.1, 5), (0.05, 2), (0.2, 1)):
            assert QA.find_trailing_plateau_start(s, tol, m) == OA.find_trailing_plateau_start(s, tol, m)


def test_plateau_degenerate_arguments():
    """n == 0 or min_size <= 0 over the public ABI: "no plateau" (= n), never a read before the array (the scan starts at n - 1)."""
    assert QA.find_trailing_plateau_start(np.zeros(0, np.float32), 0.1, -1) == 0
    assert QA.find_trailing_plateau_start(np.zeros(0, np.float32), 0.1, 0) == 0
    assert QA.find_trailing_plateau_start(np.zeros(0, np.float32), 0.1, 5) == 0
    s = np.asarray([0.0, 0.01, 0.02, 0.03], np.float32)
    assert QA.find_trailing_plateau_start(s, 0.1, 0) == 4 and QA.find_trailing_plateau_start(s, 0.1, -3) == 4
    assert QA.find_trailing_plateau_start(s, 0.1, 2) == 0
    lib = _lib.load(strict=True)
    assert lib.qasr_find_trailing_plateau(None, 3, C.c_float(0.1), 5) < 0


def test_aligner_presets():
    lib = _lib.load(strict=True)
    cfg = _lib.QasrConfig()
    k = KAT["constants"]
    for preset, bits in (("aufklarer/Qwen3-ForcedAligner-0.6B-4bit", 4), ("aufklarer/Qwen3-ForcedAligner-0.6B-8bit", 8),
                         ("aufklarer/Qwen3-ForcedAligner-0.6B-bf16", 16), ("aligner-0.6B", 4)):
        assert lib.qasr_default_config(preset.encode(), C.byref(cfg)) == 0
        e = k["aligner_encoder"]
        assert (cfg.enc_d_model, cfg.enc_heads, cfg.enc_ffn, cfg.enc_layers, cfg.enc_out_dim, cfg.conv_channels, cfg.n_window,
                cfg.n_window_infer) == (e["d_model"], e["heads"], e["ffn"], e["layers"], e["output_dim"], e["conv_channels"],
                                        e["n_window"], e["n_window_infer"])
        assert (cfg.hidden, cfg.dec_layers, cfg.heads, cfg.kv_heads, cfg.inter) == (1024, 28, 16, 8, 3072)     # TextDecoderConfig.small
        assert cfg.classify_num == k["classify_num"] and cfg.tok_timestamp == k["timestamp_token_id"]
        assert cfg.timestamp_segment_time == pytest.approx(k["segment_time"])
        assert cfg.bits == bits
    assert lib.qasr_default_config(b"0.6B", C.byref(cfg)) == 0 and cfg.classify_num == 0

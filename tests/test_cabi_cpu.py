"""CPU-side checks of the C ABI: libqasr.so loads, exports every symbol include/qasr.h declares, the
ctypes structs match the header, and the GPU-free entry points (presets, frame counts) agree with the
reference's known answers.  No compute call is made here (there is no GPU and no CPU fallback)."""
import ctypes as C
import json
import os
import re
import pytest
from conftest import GOLDEN, ROOT
from qasr import _lib

HEADER = os.path.join(ROOT, "include", "qasr.h")
KAT = json.load(open(os.path.join(GOLDEN, "kat_reference_tests.json"), encoding="utf-8"))


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qasr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load(strict=True)
    declared = _declared_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in qasr.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} lacks a ctypes signature"
    assert sorted(_lib.SIGNATURES) == declared


def test_config_struct_matches_header():
    text = open(HEADER).read()
    body = text[text.index("typedef struct qasr_config {"):text.index("} qasr_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for m in re.finditer(r"\b(int32_t|float)\s+([^;]+);", body):
        fields += [f.strip() for f in m.group(2).split(",")]
    assert [f for f, _ in _lib.QasrConfig._fields_] == fields


def _cfg(preset):
    lib = _lib.load()
    c = _lib.QasrConfig()
    assert lib.qasr_default_config(preset.encode(), C.byref(c)) == 0
    return c


def test_presets_match_reference_constants():
    k = KAT["config"]
    s = _cfg("0.6B")
    a, t = k["audio_small"], k["text_small"]
    assert (s.enc_d_model, s.enc_layers, s.enc_heads, s.enc_ffn, s.n_mels, s.enc_out_dim, s.conv_channels) == \
        (a["d_model"], a["layers"], a["heads"], a["ffn_dim"], a["n_mels"], a["output_dim"], a["conv_channels"])
    assert (s.hidden, s.dec_layers, s.heads, s.kv_heads, s.head_dim, s.inter, s.vocab, s.bits, s.group_size) == \
        (t["hidden"], t["layers"], t["heads"], t["kv_heads"], t["head_dim"], t["inter"], t["vocab"], t["bits"], t["group_size"])
    big = _cfg("1.7B")
    al, tl = k["audio_large"], k["text_large"]
    assert (big.enc_d_model, big.enc_layers, big.enc_heads, big.enc_ffn, big.enc_out_dim) == \
        (al["d_model"], al["layers"], al["heads"], al["ffn_dim"], al["output_dim"])
    assert (big.hidden, big.inter, big.dec_layers, big.heads, big.kv_heads, big.head_dim) == \
        (tl["hidden"], tl["inter"], tl["layers"], tl["heads"], tl["kv_heads"], tl["head_dim"])
    for model_id, size, bits in k["detect"]:
        c = _cfg(model_id)
        assert c.bits == bits
        assert (c.hidden == 2048) == (size == "large")
    p = KAT["prompt"]["full_no_context_A3"]
    assert [s.tok_im_start, s.tok_system, s.tok_newline, s.tok_im_end] == p[:4]
    assert (s.tok_user, s.tok_audio_start, s.tok_audio_pad, s.tok_audio_end, s.tok_assistant, s.tok_asr_text) == \
        (p[6], p[8], p[9], p[12], p[16], p[18])
    assert s.max_new_tokens == 448 and s.fft_scale == 2.0


@pytest.mark.parametrize("n,frames", [(80000, 500), (480000, 3000), (16000, 100), (400, 2), (1, 0), (160, 1)])
def test_frame_counts(n, frames):
    assert _lib.load().qasr_num_mel_frames(n) == frames


def test_error_paths_without_gpu():
    lib = _lib.load()
    assert lib.qasr_default_config(b"0.6B", None) != 0
    assert lib.qasr_input_sample_rate(None) == 16000
    assert lib.qasr_is_loaded(None) == 0
    assert lib.qasr_memory_footprint(None) == 0
    assert lib.qasr_transcribe(None, None, 0, 16000, None, None) != 0


def test_tuning_knobs_roundtrip():
    """qasr_set_tuning / qasr_get_tuning: the knob table of csrc/tuning.h (pure host state, no GPU)."""
    lib = _lib.load()
    v = C.c_int(-1)
    assert lib.qasr_get_tuning(b"gemm_nbuf", C.byref(v)) == 0 and v.value == 0
    assert lib.qasr_set_tuning(b"gemm_nbuf", 1) == 0
    assert lib.qasr_get_tuning(b"gemm_nbuf", C.byref(v)) == 0 and v.value == 1
    assert lib.qasr_set_tuning(b"gemm_nbuf", 0) == 0
    assert lib.qasr_get_tuning(b"use_graph", C.byref(v)) == 0 and v.value == 1
    assert lib.qasr_set_tuning(b"no_such_knob", 1) != 0
    assert lib.qasr_get_tuning(b"no_such_knob", C.byref(v)) != 0
    assert lib.qasr_set_tuning(None, 1) != 0
    # values outside a knob's enumerated set / range are refused and leave the table unchanged (decode_gran = 0 would divide by zero in
    # the step issue, lmh_grid <= 0 would size empty partial buffers); timing-only diagnostics exist in `make DIAG=1` builds only
    for key, bad in ((b"decode_gran", 0), (b"decode_gran", 17), (b"lmh_grid", 0), (b"lmh_grid", -4), (b"graph_steps", 3), (b"gemm_nbuf", 7),
                     (b"decode_split", 0), (b"lmh_diag", 1)):
        before = C.c_int(-1)
        assert lib.qasr_get_tuning(key, C.byref(before)) == 0
        assert lib.qasr_set_tuning(key, bad) != 0, (key, bad)
        assert lib.qasr_get_tuning(key, C.byref(v)) == 0 and v.value == before.value


# ---- pickNextToken through the C ABI (pure CPU function; reference KATs: Qwen3DecodingOptionsTests.swift:51-235)
import numpy as np  # noqa: E402


def _pick(logits, generated, repetition_penalty=1.0, no_repeat_ngram=0, temperature=0.0, rng=None):
    lib = _lib.load()
    lg = np.ascontiguousarray(logits, dtype=np.float32)
    gen = np.ascontiguousarray(generated, dtype=np.int32)
    st = C.c_uint64(rng if rng is not None else 1)
    tok = lib.qasr_pick_next_token(lg.ctypes.data_as(C.POINTER(C.c_float)), lg.shape[0],
                                   gen.ctypes.data_as(C.POINTER(C.c_int32)), gen.shape[0],
                                   repetition_penalty, no_repeat_ngram, temperature, C.byref(st))
    return tok, st.value


@pytest.mark.parametrize("case", KAT["sampler"], ids=lambda c: c["name"])
def test_sampler_kats(case):
    logits = np.full(case["n"], case["fill"], dtype=np.float32)
    for k, v in case["set"].items():
        logits[int(k)] = v
    for _ in range(3):
        assert _pick(logits, case["generated"], **case["opts"])[0] == case["expect"]


def test_sampler_matches_oracle_on_random_cases():
    from oracle import decoder
    rng = np.random.default_rng(0)
    for _ in range(200):
        n = int(rng.integers(4, 40))
        logits = rng.normal(size=n).astype(np.float32) * 3
        gen = rng.integers(0, n, size=int(rng.integers(0, 12))).tolist()
        rp = float(rng.choice([1.0, 1.1, 2.0]))
        ng = int(rng.choice([0, 2, 3]))
        assert _pick(logits, gen, rp, ng)[0] == decoder.pick_next_token(logits, gen, rp, ng)


def test_sampler_temperature_statistics():
    st = 7
    seen = set()
    for _ in range(50):
        t, st = _pick(np.zeros(16, np.float32), [], temperature=1.0, rng=st)
        seen.add(t)
    assert len(seen) >= 3
    peak = np.zeros(16, np.float32)
    peak[9] = 10.0
    hits, st = 0, 11
    for _ in range(50):
        t, st = _pick(peak, [], temperature=0.1, rng=st)
        hits += t == 9
    assert hits > 25
    assert _pick(peak, [], temperature=0.7, rng=5) == _pick(peak, [], temperature=0.7, rng=5)   # seeded => reproducible


# ---- WAV reader hardening (data of Tests/Qwen3ASRTests/SecurityHardeningTests.swift:83-196) -----------------
import struct  # noqa: E402


def _build_wav(fmt=1, channels=1, rate=16000, bits=16, samples=(0, 1000, -1000, 2000, -2000, 3000, -3000, 0),
               data_size=None, extra=b"", corrupt=False):
    block = channels * (bits // 8)
    fmt_chunk = struct.pack("<HHIIHH", fmt, channels, rate, rate * block, block, bits)
    body = b"" if corrupt else b"".join(struct.pack("<h", s) for s in samples)
    data = b"data" + struct.pack("<I", len(samples) * 2 if data_size is None else data_size) + body
    total = 4 + 8 + len(fmt_chunk) + len(extra) + len(data)
    return b"RIFF" + struct.pack("<I", total) + b"WAVE" + b"fmt " + struct.pack("<I", len(fmt_chunk)) + fmt_chunk + extra + data


def _load(tmp_path, blob):
    from qasr.model import load_wav
    p = tmp_path / "t.wav"
    p.write_bytes(blob)
    return load_wav(p)


def test_wav_valid_mono_and_stereo(tmp_path):
    s, r = _load(tmp_path, _build_wav())
    assert r == 16000 and len(s) == 8 and s[1] == np.float32(1000) / np.float32(32768)
    s, r = _load(tmp_path, _build_wav(channels=2, samples=(100, -100, 200, -200, 300, -300, 400, -400)))
    assert len(s) == 4 and np.allclose(s[:2], [100 / 32768.0, 200 / 32768.0])          # first channel only
    s, _ = _load(tmp_path, _build_wav(data_size=4))                                      # claims 2 samples
    assert len(s) == 2


@pytest.mark.parametrize("blob", [
    b"\x00" * 20,                                                  # too small
    b"NOPE" + _build_wav()[4:],                                    # missing RIFF
    _build_wav()[:8] + b"NOPE" + _build_wav()[12:],                # missing WAVE
    _build_wav(channels=0),                                        # zero channels
    _build_wav(data_size=99999),                                   # data chunk larger than the file
    _build_wav().replace(b"data", b"xxxx"),                        # no data chunk
    _build_wav(extra=b"LIST" + struct.pack("<I", 0xFFFFFFFF) + b"\x00" * 4),   # chunk with a huge size field
    _build_wav(fmt=3), _build_wav(bits=8),                         # not PCM / not 16-bit
], ids=["small", "riff", "wave", "zero_ch", "oversized", "nodata", "huge_chunk", "float", "8bit"])
def test_wav_malformed_rejected(tmp_path, blob):
    from qasr.model import QasrError
    with pytest.raises(QasrError):
        _load(tmp_path, blob)


def test_wav_reference_fixture():
    """The reference's own test clip (data fixture, Tests/Qwen3ASRTests/Resources/test_audio.wav): PCM16 mono
    24 kHz, 20.0 s, speech in 5.19-8.34 s."""
    from qasr.model import load_wav
    s, r = load_wav(os.path.join(GOLDEN, "test_audio.wav"))
    assert r == 24000 and len(s) == 480000
    assert abs(float(np.abs(s).max()) - 0.715) < 0.01
    active = np.nonzero(np.abs(s) > 0.02)[0]
    assert 5.0 < active[0] / r < 5.4 and 8.2 < active[-1] / r < 8.5


def test_dp_argument_checks_without_a_gpu():
    """qasr_dp_*: argument validation is host code (no device is touched before it passes)."""
    lib = _lib.load()
    cfg = _lib.QasrConfig()
    assert lib.qasr_default_config(b"tiny", C.byref(cfg)) == 0
    h = C.c_void_p()
    dev = (C.c_int32 * 2)(0, 0)
    assert lib.qasr_dp_create(None, None, dev, 2, C.byref(h)) != 0
    assert lib.qasr_dp_create(None, C.byref(cfg), None, 2, C.byref(h)) != 0
    assert lib.qasr_dp_create(None, C.byref(cfg), dev, 0, C.byref(h)) != 0
    assert lib.qasr_dp_create(None, C.byref(cfg), dev, 65, C.byref(h)) != 0
    assert lib.qasr_dp_n_devices(None) == 0 and lib.qasr_dp_engine(None, 0) is None
    assert lib.qasr_dp_transcribe_batch(None, None, None, 0, 16000, None, None, None) != 0
    assert lib.qasr_dp_finalize(None) != 0 and lib.qasr_dp_set_tensor(None, b"x", None, 0, None, 0) != 0
    t = C.c_int64(-1)
    assert lib.qasr_dp_submit(None, None, None, 0, 16000, None, C.byref(t)) != 0 and lib.qasr_dp_collect(None, 0, None, None) != 0
    lib.qasr_dp_destroy(None)

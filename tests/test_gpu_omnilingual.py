"""Omnilingual ASR (wav2vec2 encoder + CTC head, BASELINE configs[3]) on the device vs the CPU oracle, via the C ABI.

Bars (stated): the reference computes everything in f32 (its loader widens every tensor); the device rounds the inputs of
every contraction to bf16 (MFMA operands) and keeps LayerNorm / residual / softmax statistics / bias in f32 -- oracle policy
DEVICE.  Logits: relative L2 < 1e-2 vs DEVICE and < 2e-2 vs REFERENCE (the cost of the bf16 operands, same bars as the
Qwen3 audio encoder).  Tokens: the device's per-frame argmax must be the oracle's argmax wherever the oracle's top-2 margin
exceeds the logit tolerance (max |d| of that clip), and the collapsed id sequences obey the reference's integer semantics
exactly (duplicate collapse, batch invariance, 40 s cap, empty input)."""
import dataclasses
import numpy as np
import pytest
import torch
from conftest import GOLDEN
from oracle import omnilingual as O, precision as P
from qasr import synth
from qasr.omnilingual import OmnilingualASRMLXModel
from qasr.model import QasrError
import os

pytestmark = pytest.mark.gpu


def _wave(k, seconds):
    return synth.synth_waveform(k, seconds)


def _check_logits(m, sd, cfg, pcm, what, bar_dev=1e-2, bar_ref=2e-2, sure_frac=0.5):
    got = m.logits(pcm)
    W = O.OmniWeights(sd)
    with torch.no_grad():
        dev = O.forward(pcm, W, cfg, P.DEVICE).numpy()
        ref = O.forward(pcm, W, cfg, P.REFERENCE).numpy()
    assert got.shape == dev.shape == (O.output_length(len(pcm)), cfg.vocab)
    rd = np.linalg.norm(got - dev) / np.linalg.norm(dev)
    rr = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    tol = float(np.abs(got - dev).max())
    print(f"{what}: rel-L2 vs DEVICE {rd:.2e} vs REFERENCE {rr:.2e}, max|d| {tol:.3e}")
    assert rd < bar_dev and rr < bar_ref
    top2 = np.sort(dev, axis=1)[:, -2:]
    sure = (top2[:, 1] - top2[:, 0]) > 2 * tol
    assert (got.argmax(1)[sure] == dev.argmax(1)[sure]).all() and sure.mean() > sure_frac
    return got


@pytest.fixture(scope="module")
def tiny():
    sd = synth.synth_omnilingual_state_dict(O.OMNI_TINY, seed=1)
    m = OmnilingualASRMLXModel.from_state_dict(sd, variant="tiny", max_batch=6, max_audio_seconds=12)
    yield m, sd
    m.close()


@pytest.mark.parametrize("n", [400, 401, 721, 3000, 16000, 52345, 160000])
def test_tiny_logits_vs_oracle(tiny, n):
    """1 frame (400 samples = the receptive field), odd lengths, 10 s; every conv layer's floor((L - k) / s) + 1 edge."""
    m, sd = tiny
    pcm = _wave(n % 7, 11.0)[:n]
    _check_logits(m, sd, O.OMNI_TINY, pcm, f"tiny n={n}")


def test_hf_golden_weights_on_device(tiny):
    """The weights of the transformers-generated golden through the engine: device logits vs the golden's own logits."""
    G = np.load(os.path.join(GOLDEN, "hf_tiny_w2v.npz"))
    sd = {k[3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("sd/")}
    m = OmnilingualASRMLXModel.from_state_dict(sd, variant="tiny", max_batch=2, max_audio_seconds=2)
    try:
        for name in "abc":
            got = m.logits(G["wave/" + name])
            ref = G["logits/" + name]
            rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
            print(name, "rel-L2 vs transformers", rel)
            assert rel < 2e-2
    finally:
        m.close()


def test_batch_semantics_and_text(tiny):
    m, sd = tiny
    clips = [_wave(0, 2.5), _wave(1, 0.9), _wave(2, 0.03), _wave(3, 7.3)]
    out = m.transcribe_batch(clips)
    assert m.transcribe_batch(clips) == out                                   # deterministic
    for c, ids in zip(clips, out):
        assert m.transcribe_batch([c])[0] == ids                              # batch invariance, bit for bit
        assert all(a != b for a, b in zip(ids, ids[1:]))                       # consecutive duplicates collapsed
        assert len(ids) <= O.output_length(len(c))
    assert m.transcribe_batch([clips[3], clips[0]]) == [out[3], out[0]]
    # teacher check against the oracle: collapsed ids equal wherever every frame's argmax is unambiguous
    with torch.no_grad():
        for c, ids in zip(clips[:2], out[:2]):
            lg = O.forward(c, sd, O.OMNI_TINY, P.DEVICE).numpy()
            got = m.logits(c)
            tol = float(np.abs(got - lg).max())
            top2 = np.sort(lg, axis=1)[:, -2:]
            if ((top2[:, 1] - top2[:, 0]) > 2 * tol).all():
                assert ids == O.collapse(lg.argmax(1).tolist())
    # vocabulary: OmnilingualVocabulary.decode rules through the C ABI
    pieces = [("<s>", 3), ("<pad>", 3), ("</s>", 3), ("<unk>", 2)] + [("▁w%d" % i if i % 3 == 0 else "x%d" % i, 1) for i in range(4, 40)]
    pieces[7] = ("<0x41>", 6)
    m.set_pieces(pieces)
    for ids in out:
        assert m.detokenize(ids) == O.vocab_decode(ids, pieces)
    assert m.transcribe_audio(clips[0]) == O.vocab_decode(out[0], pieces)
    assert m.detokenize([0, 1, 2, 3, 7, 99]) == ""


def test_reference_error_behaviour(tiny):
    """40 s cap is an error, not a truncation (OmnilingualMLXModel.swift:154-159); empty input -> "" (:160-162); the
    protocol surface never raises (OmnilingualASRMLXModel+Protocols.swift:8-14); unloaded model refuses."""
    m, sd = tiny
    long_clip = np.zeros(16000 * 41, np.float32)
    with pytest.raises(QasrError, match="qasr error 5"):
        m.transcribe_audio(long_clip)
    assert m.transcribe(long_clip) == ""
    assert m.transcribe_audio(np.zeros(0, np.float32)) == ""
    with pytest.raises(QasrError, match="qasr error 5"):
        m.transcribe_audio(np.zeros(16000 * 13, np.float32))                  # above this engine's own capacity (12 s)
    assert m.transcribe(np.ones(3000, np.float32), sample_rate=8000) == ""
    m2 = OmnilingualASRMLXModel.from_state_dict(sd, variant="tiny", max_batch=1, max_audio_seconds=2)
    try:
        assert m2.is_loaded and m2.memory_footprint > 0
        m2.unload()
        assert not m2.is_loaded and m2.memory_footprint == 0
        assert m2.transcribe(np.ones(3000, np.float32)) == ""
        with pytest.raises(QasrError, match="qasr error 3"):
            m2.transcribe_audio(np.ones(3000, np.float32))
    finally:
        m2.close()


def test_7b_width_one_layer():
    """BASELINE configs[3]'s geometry (Omnilingual-ASR-CTC-7B: D 2048, 32 heads x 64, FFN 8192, positional conv groups of 128
    channels) with one transformer layer: the 256-wide GEMM tiles at K = 2048 / 8192, the grouped positional conv at N = 128,
    32-head attention, against the oracle; a ragged batch equals the single-clip calls."""
    cfg = dataclasses.replace(O.VARIANTS["7B"], layers=1)
    sd = synth.synth_omnilingual_state_dict(cfg, seed=17)
    m = OmnilingualASRMLXModel.from_state_dict(sd, variant="7B", layers=1, max_batch=3, max_audio_seconds=8)
    try:
        _check_logits(m, sd, cfg, _wave(8, 6.4), "7B width, 1 layer")
        clips = [_wave(k, 1.1 + 1.9 * k) for k in range(3)]
        out = m.transcribe_batch(clips)
        assert [m.transcribe_batch([c])[0] for c in clips] == out
    finally:
        m.close()


def test_7b_width_sixteen_layers_deep():
    """Depth: the 7B widths with SIXTEEN transformer layers on a 10 s clip against the oracle under both rounding policies -- what 16
    consecutive bf16-operand residual updates (the stated deviation of this path: f32 residual stream, bf16 MFMA operands) cost at the
    logits.  The per-layer error adds roughly like a random walk (1 layer: 3-4e-3 vs DEVICE), so the bars are the one-layer bars x 2.5;
    per-frame argmax must still agree wherever the oracle's own top-2 margin exceeds the logit tolerance."""
    cfg = dataclasses.replace(O.VARIANTS["7B"], layers=16)
    sd = synth.synth_omnilingual_state_dict(cfg, seed=23)
    m = OmnilingualASRMLXModel.from_state_dict(sd, variant="7B", layers=16, max_batch=2, max_audio_seconds=10)
    try:
        pcm = _wave(4, 10.0)
        _check_logits(m, sd, cfg, pcm, "7B width, 16 layers, 10 s", bar_dev=2.5e-2, bar_ref=5e-2, sure_frac=0.3)
        a = m.transcribe_batch([pcm, _wave(5, 4.2)])
        assert a == m.transcribe_batch([pcm, _wave(5, 4.2)]) and a[0] == m.transcribe_batch([pcm])[0]
    finally:
        m.close()


def test_7b_full_geometry_32_clips_30s_properties():
    """BASELINE configs[3] at its full size: Omnilingual-ASR-CTC-7B (128 layers, D 2048), 32 clips x 30 s in one pass.  No oracle can
    afford 128 layers x 48 k frames; what the domain offers at this size: every clip's id count is bounded by its frame count (1499),
    the pass is deterministic, and clips do not interact -- two clips alone give the ids they gave inside the batch of 32."""
    m = OmnilingualASRMLXModel.from_synthetic(variant="7B", max_batch=32, max_audio_seconds=30)
    try:
        clips = [_wave(k, 30.0 if k % 5 else 27.3) for k in range(32)]
        a = m.transcribe_batch(clips)
        ms = m.timings()
        assert len(a) == 32 and all(0 < len(t) <= m.num_frames(len(c)) for t, c in zip(a, clips))
        assert m.transcribe_batch(clips) == a
        for k in (0, 17):
            assert m.transcribe_batch([clips[k]])[0] == a[k], k
        print(f"7B, 32 x 30 s: device {ms[3]:.0f} ms per pass, ids per clip {[len(t) for t in a[:4]]}")
    finally:
        m.close()


def test_1b_width_one_layer():
    """Omnilingual-ASR-CTC-1B (D 1280, 20 heads x 64, FFN 5120; OmnilingualMLXConfig.swift:88-103): the one published width that is
    not a power of two -- positional conv groups of 80 channels, 5 / 15 / 20 column tiles of 256, the generic LayerNorm form -- with
    one transformer layer against the oracle; a ragged batch equals the single-clip calls.  (3B has the 7B widths.)"""
    cfg = dataclasses.replace(O.VARIANTS["1B"], layers=1)
    sd = synth.synth_omnilingual_state_dict(cfg, seed=19)
    m = OmnilingualASRMLXModel.from_state_dict(sd, variant="1B", layers=1, max_batch=3, max_audio_seconds=8)
    try:
        _check_logits(m, sd, cfg, _wave(9, 5.3), "1B width, 1 layer")
        clips = [_wave(k, 0.9 + 2.3 * k) for k in range(3)]
        out = m.transcribe_batch(clips)
        assert [m.transcribe_batch([c])[0] for c in clips] == out
    finally:
        m.close()


def test_maximum_clip_length_40s():
    """The reference's cap (40 s = 640 000 samples -> 1999 frames, OmnilingualMLXModel.swift:154-159) at the 300M widths, one
    layer: the longest attention sweep (32 key tiles, 16 query blocks per head), the largest conv row tables, and one sample
    more is refused."""
    cfg = dataclasses.replace(O.VARIANTS["300M"], layers=1)
    sd = synth.synth_omnilingual_state_dict(cfg, seed=13)
    m = OmnilingualASRMLXModel.from_state_dict(sd, variant="300M", layers=1, max_batch=2, max_audio_seconds=40)
    try:
        pcm = _wave(6, 40.0)
        assert len(pcm) == 640000 and O.output_length(len(pcm)) == 1999
        _check_logits(m, sd, cfg, pcm, "40 s clip, 300M width")
        ids = m.transcribe_batch([pcm, _wave(7, 3.3)])
        assert ids[0] == m.transcribe_batch([pcm])[0] and len(ids[0]) <= 1999
        with pytest.raises(QasrError, match="qasr error 5"):
            m.transcribe_audio(np.concatenate([pcm, np.zeros(1, np.float32)]))
    finally:
        m.close()


def test_non_finite_logits_are_an_error_status(tiny):
    """A NaN anywhere in the weights reaches the logits; the engine reports it (QASR_ERR_HIP, 'non-finite') instead of
    returning whatever token ids the comparisons leave -- same policy as the Qwen3 path's greedy loop."""
    _, sd = tiny
    bad = dict(sd)
    name = "encoder.layers.1.ffn.output_proj.bias"
    t = bad[name].clone().to(torch.float32)
    t[3] = float("nan")
    bad[name] = t
    m = OmnilingualASRMLXModel.from_state_dict(bad, variant="tiny", max_batch=2, max_audio_seconds=4)
    try:
        with pytest.raises(QasrError, match="non-finite"):
            m.transcribe_audio(_wave(1, 1.5))
        assert m.transcribe(_wave(1, 1.5)) == ""                  # the never-raising protocol surface
    finally:
        m.close()


@pytest.mark.parametrize("bits", [0, 4, 8], ids=["float", "mlx-4bit", "mlx-8bit"])
def test_300m_width_two_layers(bits):
    """Omnilingual-300M widths (D 1024, 16 heads x 64, FFN 4096, 512-channel extractor, k = 128 / 16-group positional conv,
    vocab 10288) with 2 transformer layers so the CPU oracle stays affordable; float and MLX-quantised linears."""
    cfg = dataclasses.replace(O.VARIANTS["300M"], layers=2)
    sd = synth.synth_omnilingual_state_dict(cfg, seed=2, bits=bits)
    m = OmnilingualASRMLXModel.from_state_dict(sd, variant="300M", layers=2, bits=bits if bits else 4, max_batch=4, max_audio_seconds=6)
    try:
        pcm = _wave(1, 5.0)
        _check_logits(m, sd, cfg, pcm, f"300M-width bits={bits}")
        clips = [_wave(k, 1.0 + 0.7 * k) for k in range(4)]
        out = m.transcribe_batch(clips)
        assert m.transcribe_batch([clips[2]])[0] == out[2]
    finally:
        m.close()


def test_attention_forms_agree_at_head_dim_64():
    """The head_dim-64 attention kernels (transposed scores on 32x32x16 MFMAs, P in registers, V by transposed LDS reads, 128 or
    256 queries per workgroup | 16x16x32 with P through LDS) against the oracle on clips whose frame counts hit: one partial tile, an exact multiple of
    the 64-key tile, several 128-query workgroups with a ragged tail; and against each other."""
    cfg = dataclasses.replace(O.VARIANTS["300M"], layers=1)
    sd = synth.synth_omnilingual_state_dict(cfg, seed=5)
    m = OmnilingualASRMLXModel.from_state_dict(sd, variant="300M", layers=1, max_batch=4, max_audio_seconds=12)
    try:
        n128 = next(n for n in range(40000, 42000) if O.output_length(n) == 128)
        for n in (int(0.4 * 16000), n128, int(11.3 * 16000)):
            pcm = _wave(3, n / 16000.0)[:n]
            m.lib.qasr_set_tuning(b"mha_form", 0)
            b = _check_logits(m, sd, cfg, pcm, f"mha_form=0 frames={O.output_length(n)}")
            for form in (1, 2):
                m.lib.qasr_set_tuning(b"mha_form", form)
                a = _check_logits(m, sd, cfg, pcm, f"mha_form={form} frames={O.output_length(n)}")
                rel = np.linalg.norm(a - b) / np.linalg.norm(b)
                print(f"form {form} vs form 0: rel-L2 {rel:.2e}")
                assert rel < 5e-3
        clips = [_wave(k, 0.9 + 2.3 * k) for k in range(4)]
        for form in (1, 2):
            m.lib.qasr_set_tuning(b"mha_form", form)
            out = m.transcribe_batch(clips)
            assert [m.transcribe_batch([c])[0] for c in clips] == out
    finally:
        m.lib.qasr_set_tuning(b"mha_form", 1)
        m.close()


def test_gemm_forms_agree_bit_for_bit():
    """Every GEMM of the path (conv feature extractor through row tables, projection, grouped positional conv, q|k|v, o, FFN,
    CTC head with its ragged 10288 columns) on the 128 x 128 forms and on the 256 x 256 ping-pong form (gemm_p8.h): each
    output is summed in the same k order, so the logits must be the same bits."""
    cfg = dataclasses.replace(O.VARIANTS["300M"], layers=2)
    sd = synth.synth_omnilingual_state_dict(cfg, seed=7)
    m = OmnilingualASRMLXModel.from_state_dict(sd, variant="300M", layers=2, max_batch=4, max_audio_seconds=12)
    try:
        pcm = _wave(2, 11.0)
        out = []
        for p8 in (0, 2, 1):
            m.lib.qasr_set_tuning(b"gemm_p8", p8)
            out.append(m.logits(pcm))
        assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])
        clips = [_wave(k, 0.9 + 2.3 * k) for k in range(4)]
        m.lib.qasr_set_tuning(b"gemm_p8", 0)
        a = m.transcribe_batch(clips)
        m.lib.qasr_set_tuning(b"gemm_p8", 2)
        assert m.transcribe_batch(clips) == a
        _check_logits(m, sd, cfg, pcm, "300M-width, gemm_p8=2")
    finally:
        m.lib.qasr_set_tuning(b"gemm_p8", 1)
        m.close()


def test_safetensors_directory_with_sentencepiece(tmp_path):
    """model.safetensors (f16 floats + uint32 triplets, as published) + tokenizer.model (SentencePiece protobuf) through
    qasr_ctc_create == the same tensors through qasr_ctc_set_tensor."""
    from safetensors.torch import save_file
    import json
    sd = synth.synth_omnilingual_state_dict(O.OMNI_TINY, seed=4, bits=8)
    disk = {k: (v if v.dtype == torch.int32 else v.to(torch.float16)) for k, v in sd.items()}
    path = tmp_path / "model.safetensors"
    save_file({k: v.contiguous() for k, v in disk.items()}, str(path))
    raw = path.read_bytes()
    hlen = int.from_bytes(raw[:8], "little")
    header = json.loads(raw[8:8 + hlen])
    for k, v in header.items():
        if k != "__metadata__" and v["dtype"] == "I32":
            v["dtype"] = "U32"
    hb = json.dumps(header, separators=(",", ":")).encode()
    hb += b" " * (hlen - len(hb))
    path.write_bytes(raw[:8] + hb + raw[8 + hlen:])
    # minimal SentencePiece ModelProto: repeated field 1 { 1: piece, 2: score (fixed32), 3: type }
    pieces = [("<s>", 3), ("<pad>", 3), ("</s>", 3), ("<unk>", 2)] + [("▁t%d" % i, 1) for i in range(4, 40)]
    blob = b""
    for text, typ in pieces:
        t = text.encode("utf-8")
        inner = b"\x0a" + bytes([len(t)]) + t + b"\x15" + np.float32(-1.5).tobytes() + (b"" if typ == 1 else b"\x18" + bytes([typ]))
        blob += b"\x0a" + bytes([len(inner)]) + inner
    blob += b"\x12\x02\x08\x01"                                   # an unrelated field (trainer_spec) the reader must skip
    (tmp_path / "tokenizer.model").write_bytes(blob)
    expect = {k: (v if v.dtype == torch.int32 else v.to(torch.float16).to(torch.float32)) for k, v in sd.items()}
    m = OmnilingualASRMLXModel(variant="tiny", model_dir=str(tmp_path), max_batch=2, max_audio_seconds=4, bits=8)
    ref = OmnilingualASRMLXModel.from_state_dict(expect, variant="tiny", max_batch=2, max_audio_seconds=4, bits=8)
    try:
        pcm = _wave(2, 3.0)
        ids = m.transcribe_batch([pcm])[0]
        assert ids == ref.transcribe_batch([pcm])[0]
        assert np.array_equal(m.logits(pcm), ref.logits(pcm))
        assert m.transcribe_audio(pcm) == O.vocab_decode(ids, pieces)
    finally:
        m.close()
        ref.close()

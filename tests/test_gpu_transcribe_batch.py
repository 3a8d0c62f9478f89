"""The reference's own measurement harness for this path -- `speech transcribe-batch` (Sources/AudioCLILib/TranscribeBatchCommand.swift:45-139:
model loaded once, one warm-up, per-file time / RTF, Aggregate RTF = sum(inference) / sum(audio)) -- mirrored over the C ABI
(qasr/transcribe_batch.py) on a tiny engine with natural EOS: same texts whether the files go one by one (the reference's loop) or in
device batches, line formats of the reference, per-file errors reported without stopping the run, transcripts written per file."""
import io
import json
import os
import re
import wave

import numpy as np
import pytest
from qasr import config as QC, synth, transcribe_batch as TB
from qasr.model import Qwen3ASRModel

pytestmark = pytest.mark.gpu


def _write_wav(path, pcm, rate=16000):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes((np.clip(pcm, -1, 1) * 32767.0).astype(np.int16).tobytes())


@pytest.fixture(scope="module")
def model():
    sd = synth.synth_state_dict(QC.AUDIO_TINY, QC.TEXT_TINY, seed=11, init="stress")
    m = Qwen3ASRModel.from_state_dict(sd, preset="tiny", max_batch=4, max_audio_seconds=4, max_new_tokens=24)
    m.set_vocab({i: ("Ġw%d" % i if i % 3 else "x%d" % i) for i in range(m.cfg.vocab)})
    yield m
    m.close()


def test_batched_run_equals_the_sequential_loop(model, tmp_path):
    d = tmp_path / "audio"
    d.mkdir()
    for k in range(7):
        _write_wav(d / f"clip{k:02d}.wav", synth.synth_waveform(k, 0.6 + 0.35 * (k % 4)))
    _write_wav(d / "zz_24k.wav", synth.synth_waveform(9, 0.5), rate=24000)           # the reference resamples; here: a per-file error
    (d / "notes.txt").write_text("not audio")
    files = TB.find_audio_files(str(d))
    assert [os.path.basename(f) for f in files] == [f"clip{k:02d}.wav" for k in range(7)] + ["zz_24k.wav"]
    seq_out, bat_out = io.StringIO(), io.StringIO()
    seq = TB.run(model, files, batch=1, out=seq_out, max_tokens=20)
    bat = TB.run(model, files, batch=4, out=bat_out, output_dir=str(tmp_path / "txt"), max_tokens=20)
    assert seq["texts"] == bat["texts"] and len(seq["texts"]) == 7
    for name, text in bat["texts"].items():
        assert (tmp_path / "txt" / f"{name}.txt").read_text(encoding="utf-8") == text
    lines = bat_out.getvalue().splitlines()
    assert lines[0] == "Found 8 audio files" and re.fullmatch(r"  Warmup: \d+\.\d\ds", lines[1])
    per_file = [l for l in lines if re.match(r"  \[\d+/8\] \(\d+%\) clip\d\d: .*  \(\d+\.\d\ds, RTF=\d+\.\d{3}\)$", l)]
    assert len(per_file) == 7
    assert any(l.startswith("  [8/8] zz_24k: ERROR - ") for l in lines)
    total_audio = sum(0.6 + 0.35 * (k % 4) for k in range(7))
    assert f"Batch complete: 8 files, {total_audio:.1f}s audio" in bat_out.getvalue()
    m = re.search(r"Total inference: (\d+\.\d\d)s, Aggregate RTF: (\d+\.\d{4})", bat_out.getvalue())
    assert m and abs(float(m.group(2)) - bat["aggregate_rtf"]) < 1e-4 and abs(bat["total_audio"] - total_audio) < 1e-3
    assert bat["aggregate_rtf"] == pytest.approx(bat["total_inference"] / bat["total_audio"])
    # JSON lines, like --jsonl
    js = io.StringIO()
    TB.run(model, files[:3], batch=2, jsonl=True, out=js, max_tokens=20)
    recs = [json.loads(l) for l in js.getvalue().splitlines() if l.startswith("{")]
    assert [r["file"] for r in recs] == ["clip00", "clip01", "clip02"] and all(set(r) == {"file", "text", "time", "rtf", "duration"} for r in recs)
    assert [r["text"] for r in recs] == [seq["texts"][f"clip{k:02d}"] for k in range(3)]


def test_groups_in_flight_give_the_same_texts(model, tmp_path):
    """--lanes: groups go whole to engines sharing the GPU (qasr_dp_submit), two in flight; same texts, charges add up to the wall time."""
    from qasr.dp import Qwen3ASRDataParallel
    d = tmp_path / "audio"
    d.mkdir()
    for k in range(9):
        _write_wav(d / f"clip{k:02d}.wav", synth.synth_waveform(k, 0.6 + 0.35 * (k % 4)))
    _write_wav(d / "clip04b_24k.wav", synth.synth_waveform(9, 0.5), rate=24000)
    files = TB.find_audio_files(str(d))
    seq = TB.run(model, files, batch=1, out=io.StringIO(), max_tokens=20)
    sd = synth.synth_state_dict(QC.AUDIO_TINY, QC.TEXT_TINY, seed=11, init="stress")
    dp = Qwen3ASRDataParallel.from_state_dict(sd, [0, 0], preset="tiny", max_batch=4, max_audio_seconds=4, max_new_tokens=24)
    try:
        view = dp.engine(0)
        view.set_vocab({i: ("Ġw%d" % i if i % 3 else "x%d" % i) for i in range(view.cfg.vocab)})
        out = io.StringIO()
        r = TB.run(view, files, batch=2, out=out, max_tokens=20, lanes=dp)
        assert r["texts"] == seq["texts"] and len(r["texts"]) == 9
        order = [int(m.group(1)) for m in re.finditer(r"^  \[(\d+)/10\]", out.getvalue(), re.M)]
        assert order == list(range(1, 11))                                       # reported in file order, the error line in its place
        assert r["total_inference"] <= r["wall"] + 1e-6 and r["total_inference"] > 0.5 * r["wall"]
        view.close()                                                             # a view: the engine stays with dp
        assert dp.transcribe_batch([synth.synth_waveform(0, 0.6)], max_tokens=5)
    finally:
        dp.close()


def test_empty_directory(model, tmp_path):
    out = io.StringIO()
    r = TB.run(model, [], out=out)
    assert r["total_audio"] == 0.0 and "No audio files found" in out.getvalue()

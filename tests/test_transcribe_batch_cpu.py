"""Host logic of the `speech transcribe-batch` mirror (qasr/transcribe_batch.py; reference: Sources/AudioCLILib/TranscribeBatchCommand.swift:45-139)
without a GPU: file discovery, grouping, per-file errors, the report lines and -- with groups in flight over lanes (qasr_dp_submit / qasr_dp_collect)
-- submission order, collection order and the time accounting.  The engine is a stand-in that "transcribes" a clip to its sample count."""
import io
import json
import re
import wave

import numpy as np
from qasr import transcribe_batch as TB


class FakeModel:
    def __init__(self):
        self.calls = []

    def transcribe(self, pcm, **kw):
        self.calls.append(("warmup", len(pcm)))
        return "w"

    def transcribe_batch(self, clips, **kw):
        self.calls.append(("batch", [len(c) for c in clips]))
        return [[len(c)] for c in clips]

    def detokenize(self, toks):
        return "n%d" % toks[0]

    def encode_text(self, text):
        return [1, 2]


class FakeLanes:
    """Tickets complete in submission order; at most n_devices may be in flight (the C ABI refuses more)."""
    def __init__(self, n):
        self.n_devices, self.in_flight, self.log, self.next = n, {}, [], 0

    def submit(self, clips, **kw):
        assert len(self.in_flight) < self.n_devices, "a submit while every lane is busy"
        t = self.next
        self.next += 1
        self.in_flight[t] = [[len(c)] for c in clips]
        self.log.append(("submit", t, len(clips)))
        return t

    def collect(self, ticket):
        self.log.append(("collect", ticket))
        return self.in_flight.pop(ticket)


def _write(path, n, rate=16000):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes((np.sin(np.arange(n) * 0.01) * 8000).astype(np.int16).tobytes())


def _dir(tmp_path, n=7):
    d = tmp_path / "a"
    d.mkdir()
    for k in range(n):
        _write(d / f"c{k:02d}.wav", 1600 * (k + 2))
    _write(d / "c03b.wav", 800, rate=24000)            # the reference resamples; here a per-file error, in its place in the order
    (d / "readme.txt").write_text("x")
    return d


def test_sequential_and_grouped_runs_report_the_same_files(tmp_path):
    files = TB.find_audio_files(str(_dir(tmp_path)))
    assert [f.rsplit("/", 1)[1] for f in files] == ["c00.wav", "c01.wav", "c02.wav", "c03.wav", "c03b.wav", "c04.wav", "c05.wav", "c06.wav"]
    m1, m3 = FakeModel(), FakeModel()
    o1, o3 = io.StringIO(), io.StringIO()
    r1 = TB.run(m1, files, batch=1, out=o1)
    r3 = TB.run(m3, files, batch=3, out=o3, language="en")
    assert r1["texts"] == r3["texts"] == {f"c{k:02d}": "n%d" % (1600 * (k + 2)) for k in range(7)}
    assert m3.calls[0][0] == "warmup" and [c[1] for c in m3.calls[1:]] == [[3200, 4800, 6400], [8000, 9600], [11200, 12800]]   # the 24 kHz file drops out of its group
    for out in (o1, o3):
        lines = out.getvalue().splitlines()
        assert lines[0] == "Found 8 audio files" and re.fullmatch(r"  Warmup: \d+\.\d\ds", lines[1])
        assert [int(m.group(1)) for m in re.finditer(r"^  \[(\d+)/8\]", out.getvalue(), re.M)] == list(range(1, 9))
        assert "  [5/8] c03b: ERROR - 24000 Hz input (16 kHz only)" in lines
        assert "Batch complete: 8 files, %.1fs audio" % (sum(1600 * (k + 2) for k in range(7)) / 16000) in out.getvalue()
    assert abs(r3["aggregate_rtf"] - r3["total_inference"] / r3["total_audio"]) < 1e-12


def test_groups_in_flight_keep_order_and_account_for_the_wall_time(tmp_path):
    files = TB.find_audio_files(str(_dir(tmp_path)))
    model, lanes, out = FakeModel(), FakeLanes(2), io.StringIO()
    r = TB.run(model, files, batch=2, out=out, lanes=lanes, jsonl=True)
    assert r["texts"] == {f"c{k:02d}": "n%d" % (1600 * (k + 2)) for k in range(7)}
    assert [c[0] for c in model.calls] == ["warmup"]                          # every group went through the lanes
    # two groups in flight: the oldest is collected right before the third submit, and so on; the rest drains in order
    assert lanes.log == [("submit", 0, 2), ("submit", 1, 2), ("collect", 0), ("submit", 2, 1), ("collect", 1), ("submit", 3, 2), ("collect", 2), ("collect", 3)]
    recs = [json.loads(l) for l in out.getvalue().splitlines() if l.startswith("{")]
    assert [x["file"] for x in recs] == ["c00", "c01", "c02", "c03", "c03b", "c04", "c05", "c06"] and "error" in recs[4]
    assert 0 < r["total_inference"] <= r["wall"] + 1e-9                        # a group is charged the time since the previous one finished


def test_a_group_of_unreadable_files_takes_no_lane(tmp_path):
    d = tmp_path / "b"
    d.mkdir()
    _write(d / "a0.wav", 1600, rate=8000)
    _write(d / "a1.wav", 1600, rate=8000)
    _write(d / "b0.wav", 3200)
    files = TB.find_audio_files(str(d))
    model, lanes, out = FakeModel(), FakeLanes(1), io.StringIO()
    model.transcribe = lambda pcm, **kw: "w"
    try:
        TB.run(model, files, batch=2, out=out, lanes=lanes)
        raised = False
    except ValueError:
        raised = True                                                         # the warm-up file itself is 8 kHz: the command refuses like any loader error
    assert raised
    files = files[::-1]                                                       # b0 first: the warm-up passes, the 8 kHz pair forms an all-error group
    r = TB.run(model, files, batch=1, out=io.StringIO(), lanes=lanes)
    assert list(r["texts"]) == ["b0"] and [e for e in lanes.log if e[0] == "submit"] == [("submit", 0, 1)]

#!/usr/bin/env python3
"""`speech transcribe-batch` of the reference (Sources/AudioCLILib/TranscribeBatchCommand.swift:45-139) over the C ABI: a directory of
audio files, model loaded once, ONE warm-up transcription of the first file, then every file with per-file time / RTF and
`Aggregate RTF = sum(inference) / sum(audio)` -- the in-tree way the reference measures this path.  Same line
formats.  Differences, all stated: the reference transcribes the files one after the other (`for (idx, fileURL) in files.enumerated()`,
:82-93); here `--batch N` files go through one `qasr_transcribe_batch` call (natural EOS, ragged lengths; N = 1 is the reference's
loop) and a file's time is its batch's elapsed time apportioned by audio duration; input must be 16 kHz PCM16 WAV (the reference
loads at 24 kHz and resamples with AVAudioConverter -- closed source, out of scope); there is no downloader: `model_dir` is a local
directory in the reference's cache layout.

With `--lanes N` (N > 1) up to N groups are in flight on the GPU (qasr_dp_submit / qasr_dp_collect over N engines sharing the device): a
group is charged the time since the previous group finished, so `Total inference` is the wall time of the overlapped passes.

usage: python -m qasr.transcribe_batch INPUT_DIR --model-dir DIR [--model 0.6B] [--batch 32] [--lanes 1] [--language en] [--output-dir D] [--jsonl]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

from .model import Qwen3ASRModel, load_wav


def find_audio_files(input_dir, extensions=("wav",)):
    """findAudioFiles (:141-152): files of the directory with one of the extensions, sorted by name."""
    names = sorted(n for n in os.listdir(input_dir) if n.rsplit(".", 1)[-1].lower() in extensions and os.path.isfile(os.path.join(input_dir, n)))
    return [os.path.join(input_dir, n) for n in names]


def run(model, files, batch=1, language=None, output_dir=None, jsonl=False, out=sys.stdout, max_tokens=448, lanes=None):
    """The command's body after model loading.  -> dict(total_inference, total_audio, aggregate_rtf, wall, warmup, texts).
    lanes: a Qwen3ASRDataParallel whose engines share the GPU(s) -- group k goes whole to engine k % n (qasr_dp_submit) and up to n groups
    are in flight; `model` is then one of its engines (warm-up, tokenizer).  Texts are the same either way."""
    def emit(line):
        print(line, file=out, flush=True)

    if not files:
        emit("No audio files found")
        return {"total_inference": 0.0, "total_audio": 0.0, "aggregate_rtf": 0.0, "wall": 0.0, "warmup": 0.0, "texts": {}}
    emit(f"Found {len(files)} audio files")
    if output_dir:
        os.makedirs(output_dir, exist_ok=True)
    t0 = time.perf_counter()
    pcm, rate = load_wav(files[0])
    if rate != 16000:
        raise ValueError(f"{files[0]}: {rate} Hz (16 kHz input only: the reference's resampler is closed source)")
    model.transcribe(pcm, sample_rate=16000, language=language, max_tokens=max_tokens)        # warm-up (:69-75)
    warmup = time.perf_counter() - t0
    emit("  Warmup: %.2fs" % warmup)
    total_inference = total_audio = 0.0
    texts = {}
    batch_start = time.perf_counter()
    lang_ids = model.encode_text("language " + language) if language else None
    state = {"tick": batch_start, "io": 0.0}
    pending = []                       # lanes: (ticket, group, clips, names, errors) of the passes in flight, oldest first

    def report(group, clips, names, errors, toks, elapsed):
        nonlocal total_inference, total_audio
        results = [model.detokenize(t) for t in toks]
        durations = [c.shape[0] / 16000.0 for c in clips]
        dsum = max(sum(durations), 1e-3)
        k = 0
        for path in group:
            name = os.path.splitext(os.path.basename(path))[0]
            idx = files.index(path)
            pct = (idx + 1) / len(files) * 100.0
            if name in errors:
                emit(json.dumps({"file": name, "error": errors[name]}) if jsonl else f"  [{idx + 1}/{len(files)}] {name}: ERROR - {errors[name]}")
                continue
            text, duration = results[k], durations[k]
            share = elapsed * duration / dsum
            rtf = share / max(duration, 1e-3)
            k += 1
            total_inference += share
            total_audio += duration
            texts[name] = text
            if jsonl:
                emit(json.dumps({"file": name, "text": text, "time": round(share, 3), "rtf": round(rtf, 4), "duration": round(duration, 2)}, ensure_ascii=False))
            else:
                emit("  [%d/%d] (%.0f%%) %s: %s  (%.2fs, RTF=%.3f)" % (idx + 1, len(files), pct, name, text, share, rtf))
            if output_dir:
                with open(os.path.join(output_dir, name + ".txt"), "w", encoding="utf-8") as f:
                    f.write(text)

    def collect_oldest():
        ticket, group, clips, names, errors = pending.pop(0)
        toks = lanes.collect(ticket) if ticket is not None else []
        now = time.perf_counter()
        # passes overlap on the GPU: a group is charged the time since the previous group finished MINUS the file reading done in that
        # interval (the reference's per-file timer starts after the load, TranscribeBatchCommand.swift:91-94), so the charges add up to the
        # inference wall time of the whole job and Aggregate RTF stays sum(charged) / sum(audio)
        report(group, clips, names, errors, toks, max(now - state["tick"] - state["io"], 0.0))
        state["tick"] = now
        state["io"] = 0.0

    for b0 in range(0, len(files), batch):
        group = files[b0:b0 + batch]
        clips, names, errors = [], [], {}
        t_io = time.perf_counter()
        for path in group:
            name = os.path.splitext(os.path.basename(path))[0]
            try:
                pcm, rate = load_wav(path)
                if rate != 16000:
                    raise ValueError(f"{rate} Hz input (16 kHz only)")
                if pcm.shape[0] == 0:
                    raise ValueError("empty audio")
                clips.append(pcm)
                names.append(name)
            except Exception as ex:      # noqa: BLE001 -- per-file errors are reported and the batch goes on (:121-127)
                errors[name] = str(ex)
        if lanes is not None:
            state["io"] += time.perf_counter() - t_io
            if len(pending) == lanes.n_devices:
                collect_oldest()
            ticket = lanes.submit(clips, max_tokens=max_tokens, language_ids=lang_ids) if clips else None
            pending.append((ticket, group, clips, names, errors))
            continue
        elapsed = 0.0
        toks = []
        if clips:
            t0 = time.perf_counter()
            toks = model.transcribe_batch(clips, max_tokens=max_tokens, language_ids=lang_ids)
            elapsed = time.perf_counter() - t0
        report(group, clips, names, errors, toks, elapsed)
    while pending:
        collect_oldest()
    wall = time.perf_counter() - batch_start
    agg = total_inference / max(total_audio, 1e-3)
    emit("\nBatch complete: %d files, %.1fs audio" % (len(files), total_audio))
    emit("  Total inference: %.2fs, Aggregate RTF: %.4f" % (total_inference, agg))
    emit("  Wall time: %.2fs (includes I/O)" % wall)
    return {"total_inference": total_inference, "total_audio": total_audio, "aggregate_rtf": agg, "wall": wall, "warmup": warmup, "texts": texts}


def main(argv=None):
    ap = argparse.ArgumentParser(prog="transcribe-batch", description="Transcribe a directory of audio files (model loaded once)")
    ap.add_argument("input_dir")
    ap.add_argument("--model-dir", required=True, help="local checkpoint directory (reference cache layout)")
    ap.add_argument("--model", default="0.6B", help="model id / size for preset detection (Qwen3ASR.swift:581-601)")
    ap.add_argument("--output-dir")
    ap.add_argument("--language")
    ap.add_argument("--batch", type=int, default=32, help="files per device pass (1 = the reference's sequential loop)")
    ap.add_argument("--lanes", type=int, default=1, help="passes in flight on the GPU (engines sharing the device, qasr_dp_submit); 2-3 "
                    "hide the decode stage's launch latency: +30 %% throughput at 32 x 30 s")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--max-audio-seconds", type=int, default=120)
    ap.add_argument("--jsonl", action="store_true")
    a = ap.parse_args(argv)
    files = find_audio_files(a.input_dir)
    if not files:
        print(f"No audio files found in {a.input_dir}")
        return 0
    t0 = time.perf_counter()
    lanes = None
    if a.lanes > 1:
        from .dp import Qwen3ASRDataParallel
        lanes = Qwen3ASRDataParallel.from_pretrained(a.model_dir, [a.device] * a.lanes, model_id=a.model, max_batch=a.batch,
                                                     max_audio_seconds=a.max_audio_seconds)
        model = lanes.engine(0)
    else:
        model = Qwen3ASRModel.from_pretrained(a.model_dir, model_id=a.model, device=a.device, max_batch=a.batch,
                                              max_audio_seconds=a.max_audio_seconds)
    load = time.perf_counter() - t0
    print("  Model loaded in %.2fs" % load)
    try:
        r = run(model, files, batch=a.batch, language=a.language, output_dir=a.output_dir, jsonl=a.jsonl, lanes=lanes)
        print("  Model load: %.2fs, Warmup: %.2fs" % (load, r["warmup"]))
    finally:
        model.close()
        if lanes is not None:
            lanes.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())

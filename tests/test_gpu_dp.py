"""Utterance-batch data parallelism behind the C ABI (qasr_dp_*): one process, one engine + host thread per listed device, contiguous clip
blocks, tokens gathered by per-engine device -> host copies into the caller's block.  A one-GPU box can only rehearse it with engines
sharing device 0 (the partition, the threads, the slices through an engine's capacity and the gather are the same code on n GPUs):
results must equal one engine's qasr_transcribe_batch -- clips are independent (Qwen3ASR.swift:131-164)."""
import ctypes as C

import numpy as np
import pytest
from qasr import _lib, config as QC, synth
from qasr.dp import Qwen3ASRDataParallel
from qasr.model import Qwen3ASRModel, QasrError

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd():
    return synth.synth_state_dict(QC.AUDIO_TINY, QC.TEXT_TINY, seed=5, init="stress")


@pytest.fixture(scope="module")
def single(sd):
    m = Qwen3ASRModel.from_state_dict(sd, preset="tiny", max_batch=16, max_audio_seconds=4, max_new_tokens=12)
    yield m
    m.close()


def _clips(n):
    return [synth.synth_waveform(k, 0.5 + 0.21 * (k % 7)) for k in range(n)]


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]], ids=["n1", "n2-shared", "n3-shared"])
def test_dp_equals_single_engine(sd, single, devices):
    dp = Qwen3ASRDataParallel.from_state_dict(sd, devices, preset="tiny", max_batch=4, max_audio_seconds=4, max_new_tokens=12)
    try:
        assert dp.n_devices == len(devices)
        for B in (1, 2, 5, 13):                       # fewer clips than engines, ragged blocks, blocks beyond one engine's capacity (4)
            clips = _clips(B)
            want = single.transcribe_batch(clips, max_tokens=9, ignore_eos=True)
            assert dp.transcribe_batch(clips, max_tokens=9, ignore_eos=True) == want, (devices, B)
            assert dp.transcribe_batch(clips, max_tokens=9) == single.transcribe_batch(clips, max_tokens=9)      # natural EOS: ragged lengths
        assert len(dp.timings()) == len(devices) and all(t >= 0 for t in dp.timings())
        assert dp.transcribe_batch([]) == []
    finally:
        dp.close()


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]], ids=["n1", "n2-shared", "n3-shared"])
def test_batches_in_flight_equal_single_engine(sd, single, devices):
    """qasr_dp_submit / qasr_dp_collect: batch k whole on engine k % n, n passes in flight on the GPU, tokens identical to one engine's."""
    dp = Qwen3ASRDataParallel.from_state_dict(sd, devices, preset="tiny", max_batch=4, max_audio_seconds=4, max_new_tokens=12)
    try:
        n = len(devices)
        batches = [_clips(B)[k % 3:] for k, B in enumerate((4, 7, 3, 9, 4, 5, 2))]      # ragged, some beyond an engine's capacity (4)
        want = [single.transcribe_batch(c, max_tokens=9) for c in batches]
        got, pending = {}, []
        for k, c in enumerate(batches):                                                 # the serving loop: collect the oldest, submit the next
            if len(pending) == n:
                t0, k0 = pending.pop(0)
                got[k0] = dp.collect(t0)
            pending.append((dp.submit(c, max_tokens=9), k))
        with pytest.raises(QasrError, match="collect it first"):                        # the one-batch-over-all-engines form refuses meanwhile
            dp.transcribe_batch(batches[0], max_tokens=9)
        if n == 1:
            with pytest.raises(QasrError, match="still holds ticket"):
                dp.submit(batches[0], max_tokens=9)
        for t, k in reversed(pending):                                                  # any collection order
            got[k] = dp.collect(t)
        assert [got[k] for k in range(len(batches))] == want
        with pytest.raises(QasrError, match="not in flight"):
            dp.collect(pending[0][0])
        t = dp.submit([], max_tokens=9)                                                 # an empty batch is a ticket like any other
        assert dp.collect(t) == []
        assert dp.transcribe_batch(batches[1], max_tokens=9) == want[1]                 # and the synchronous form works again
    finally:
        dp.close()


def test_a_failed_batch_in_flight_surfaces_at_collect(sd):
    """The engine's status and message come back through qasr_dp_collect; the lane is free again afterwards and the next batch runs."""
    dp = Qwen3ASRDataParallel.from_state_dict(sd, [0, 0], preset="tiny", max_batch=2, max_audio_seconds=2, max_new_tokens=8)
    try:
        good = _clips(2)
        t_bad = dp.submit(_clips(1) + [np.zeros(16000 * 3, np.float32)], max_tokens=4)      # a clip beyond the engines' capacity
        t_good = dp.submit(good, max_tokens=4, ignore_eos=True)
        with pytest.raises(QasrError, match="qasr error 5"):
            dp.collect(t_bad)
        got = dp.collect(t_good)
        assert [len(g) for g in got] == [4, 4]
        t = dp.submit(good, max_tokens=4, ignore_eos=True)                                    # the lane that failed takes the next batch
        assert dp.collect(t) == got
    finally:
        dp.close()


def test_destroy_with_a_batch_in_flight(sd):
    dp = Qwen3ASRDataParallel.from_state_dict(sd, [0, 0], preset="tiny", max_batch=4, max_audio_seconds=4, max_new_tokens=12)
    clips = _clips(4)
    dp.submit(clips, max_tokens=9)
    dp.close()                                                                          # waits for the engine's thread, then frees


def test_dp_errors(sd):
    lib = _lib.load(strict=True)
    with pytest.raises(QasrError):
        Qwen3ASRDataParallel([0, 99], preset="tiny", max_batch=2, max_audio_seconds=2, max_new_tokens=8)     # no such device: nothing leaks
    dp = Qwen3ASRDataParallel([0, 0], preset="tiny", max_batch=2, max_audio_seconds=2, max_new_tokens=8)
    try:
        with pytest.raises(QasrError, match="qasr error 3"):                                                # weights never set: every engine refuses
            dp.transcribe_batch(_clips(3), max_tokens=4)
        assert b"engine" in lib.qasr_dp_last_error(dp.h)
    finally:
        dp.close()
    dp = Qwen3ASRDataParallel.from_state_dict(sd, [0, 0], preset="tiny", max_batch=2, max_audio_seconds=2, max_new_tokens=8)
    try:
        with pytest.raises(QasrError, match="qasr error 5"):                                                # a clip beyond the engines' capacity
            dp.transcribe_batch(_clips(3) + [np.zeros(16000 * 3, np.float32)], max_tokens=4)
        assert dp.transcribe_batch(_clips(3), max_tokens=4, ignore_eos=True)                                # and the handle still works
    finally:
        dp.close()

"""The decode layer's linears as one persistent launch with in-launch hand-offs (csrc/dec_chain.hip, tuning knob `chain`) against
the five-launch layer it replaces, through the C ABI.  Both run the same arithmetic per 16-row group (same k order over the waves,
same cross-wave order, same norm sums), so the bar is BIT equality: logits of forced steps at one row, tokens at every batch size
that selects another instantiation (1 and 8 rows: one batch tile, partly empty; 16: one full tile; 17: the second row group nearly
empty; 32: both full).  Three decoder layers at the 0.6B widths: with chain = 3 the first two launches carry the next layer's q|k|v,
the last one does not.  The parity of the five-launch layer itself against the oracle is tests/test_gpu_decoder.py / test_gpu_bench_shapes.py."""
import dataclasses
import numpy as np
import pytest
import torch
from oracle import config as C, precision as P
from qasr import synth
import gpu_util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    t = dataclasses.replace(C.TEXT_SMALL, layers=3)
    sd = synth.synth_state_dict(dataclasses.replace(C.AUDIO_SMALL, layers=1), t, seed=0, init="stress")
    e = gpu_util.Engine("0.6B", max_batch=32, max_audio_seconds=4, max_new_tokens=16, enc_layers=1, dec_layers=3)
    e.load_state_dict(sd)
    yield e
    e.set_tuning("chain", 0)
    e.set_tuning("chain_proto", 0)
    e.set_tuning("chain_pf", 0)
    e.set_tuning("qa", 1)
    e.set_tuning("qa_gran", 1)
    e.close()


def _run(eng, clips, emb, forced):
    first = eng.prefill_logits(emb)
    steps = eng.decode_forced(forced)
    return first, steps, [eng.transcribe_batch(clips[:b], max_tokens=7, ignore_eos=True) for b in (1, 8, 16, 17, 32)]


@pytest.mark.parametrize("proto,pf", [(0, 0), (1, 1)], ids=["sharded-counters-all-requests-at-entry", "replicated-counters-staged-requests"])
def test_chain_equals_five_launch_layer_bit_for_bit(eng, proto, pf):
    emb = P.bf16_round(torch.randn(33, 1024, generator=torch.Generator().manual_seed(7)) * 0.5).numpy()
    clips = [synth.synth_waveform(k, 1.0 + 0.17 * (k % 5)) for k in range(32)]
    eng.set_tuning("chain", 0)
    eng.set_tuning("qa", 0)                                    # base = the five-launch layer
    eng.set_tuning("chain_proto", proto)
    eng.set_tuning("chain_pf", pf)
    base = _run(eng, clips, emb, [11, 151643, 5, 9000, 77])
    assert len({tuple(t) for t in base[2][-1]}) > 1            # the rows do differ
    for mode, qa in ((1, 0), (2, 0), (3, 0), (0, 1), (2, 1), (0, 2)):
        eng.set_tuning("chain", mode)
        eng.set_tuning("qa", min(qa, 1))
        eng.set_tuning("qa_gran", 0 if qa == 2 else 1)            # qa 2 here: the fused launch with its first hand-off form (counter + row loads)
        got = _run(eng, clips, emb, [11, 151643, 5, 9000, 77])
        assert np.array_equal(got[0], base[0]), (mode, qa)
        assert np.array_equal(got[1], base[1]), (mode, qa, float(np.abs(got[1] - base[1]).max()))
        for b, (g, w) in zip((1, 8, 16, 17, 32), zip(got[2], base[2])):
            assert g == w, (mode, qa, b)
    eng.set_tuning("chain", 0)
    eng.set_tuning("qa", 1)                                    # the library's defaults
    eng.set_tuning("qa_gran", 1)


def test_chain_natural_eos_and_reruns(eng):
    """Replayed graphs: the counters are zeroed by a memset node at the start of every step, so any number of steps and reruns count
    from zero; rows that finish early (natural EOS under the stress weights, or the 448-token cap) keep running through the launch."""
    clips = [synth.synth_waveform(40 + k, 0.8 + 0.1 * (k % 7)) for k in range(24)]
    eng.set_tuning("chain", 0)
    eng.set_tuning("qa", 0)
    want = eng.transcribe_batch(clips, max_tokens=16)
    for mode, qa in ((3, 0), (2, 1), (0, 1)):
        eng.set_tuning("chain", mode)
        eng.set_tuning("qa", qa)
        for _ in range(3):
            assert eng.transcribe_batch(clips, max_tokens=16) == want
    eng.set_tuning("chain", 0)
    eng.set_tuning("qa", 1)


def test_qa_request_schedules_agree(eng):
    """The request schedules of the fused q|k|v + attention launch (knobs qa_early, qa_gate: which waves ask for the first K half before the
    projection, whether the rest waits for the hand-off's signal) only move requests in time: same tokens as the two-launch layer."""
    clips = [synth.synth_waveform(70 + k, 0.9 + 0.13 * (k % 6)) for k in range(32)]
    eng.set_tuning("chain", 0)
    eng.set_tuning("qa", 0)
    want = [eng.transcribe_batch(clips[:b], max_tokens=6, ignore_eos=True) for b in (1, 32)]
    eng.set_tuning("qa", 1)
    try:
        for gran in (1, 0):
            eng.set_tuning("qa_gran", gran)
            for early, gate in ((0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1), (3, 0), (3, 1), (4, 0), (4, 1)):
                eng.set_tuning("qa_early", early)
                eng.set_tuning("qa_gate", gate)
                assert [eng.transcribe_batch(clips[:b], max_tokens=6, ignore_eos=True) for b in (1, 32)] == want, (gran, early, gate)
    finally:
        eng.set_tuning("qa_early", 5)
        eng.set_tuning("qa_gate", 2)
        eng.set_tuning("qa_gran", 1)
        eng.set_tuning("qa", 1)


@pytest.mark.parametrize("gran", [1, 0], ids=["tagged-granules", "arrival-counter"])
def test_lost_arrival_ends_in_an_error_not_a_hang(eng, gran):
    """Every in-launch wait is bounded by a wall-clock budget (200 ms): with one workgroup's arrival withheld (test knob chain_fault) the
    waiting workgroups give up, the step's later waits fail at their first poll, qasr_batch_tokens reports QASR_ERR_HIP with a message --
    in well under the time a hang would take -- and the engine serves the next batch normally."""
    import time
    clips = [synth.synth_waveform(90 + k, 1.0) for k in range(32)]
    eng.set_tuning("chain", 0)
    eng.set_tuning("qa", 1)
    eng.set_tuning("qa_gran", gran)
    want = eng.transcribe_batch(clips, max_tokens=6, ignore_eos=True)
    eng.set_tuning("chain_fault", 1)
    try:
        t0 = time.perf_counter()
        with pytest.raises(RuntimeError, match="hand-off wait gave up"):
            eng.transcribe_batch(clips, max_tokens=6, ignore_eos=True)
        assert time.perf_counter() - t0 < 5.0
    finally:
        eng.set_tuning("chain_fault", 0)
    assert eng.transcribe_batch(clips, max_tokens=6, ignore_eos=True) == want
    eng.set_tuning("qa_gran", 1)


def test_qa_long_context_second_round():
    """q|k|v + attention in one launch (csrc/dec_qa.hip): a context beyond the 512 keys that the first round of requests covers (16 chunks
    of 32 keys over 8 waves) takes further rounds inside the sweep, wave 0 requests its chunks only after the hand-off: a 41 s clip
    (16 + 533 prompt positions) next to a short one, bit-equal to the two-launch layer."""
    t = dataclasses.replace(C.TEXT_SMALL, layers=2)
    sd = synth.synth_state_dict(dataclasses.replace(C.AUDIO_SMALL, layers=1), t, seed=2, init="stress")
    e = gpu_util.Engine("0.6B", max_batch=2, max_audio_seconds=42, max_new_tokens=24, enc_layers=1, dec_layers=2)
    try:
        e.load_state_dict(sd)
        clips = [synth.synth_waveform(3, 41.0), synth.synth_waveform(4, 2.0)]
        e.set_tuning("qa", 0)
        want = e.transcribe_batch(clips, max_tokens=20, ignore_eos=True)
        e.set_tuning("qa", 1)
        for gran in (1, 0):
            e.set_tuning("qa_gran", gran)
            assert e.transcribe_batch(clips, max_tokens=20, ignore_eos=True) == want, gran
    finally:
        e.set_tuning("qa_gran", 1)
        e.set_tuning("qa", 1)
        e.close()


@pytest.mark.parametrize("bits", [4, 8], ids=["w4", "w8"])
def test_qa_on_quantised_checkpoints_equals_the_two_launches(bits):
    """MLX affine-quantised decoder (the reference's shipped format, QuantizedTextDecoder.swift:33-44): the fused launch projects with the packed
    q|k|v image (dec_qa.hip qa_project_q = decode_gemvq_kernel's arithmetic, tile for tile) -- logits of forced steps and tokens at every batch
    tile shape bit-equal to the q|k|v GEMV + attention launches."""
    t = dataclasses.replace(C.TEXT_SMALL, layers=3)
    sd = synth.synth_state_dict(dataclasses.replace(C.AUDIO_SMALL, layers=1), t, seed=3, init="stress")
    e = gpu_util.Engine("0.6B", max_batch=32, max_audio_seconds=4, max_new_tokens=16, enc_layers=1, dec_layers=3, bits=bits)
    try:
        e.load_state_dict(synth.quantize_state_dict(sd, bits))
        emb = P.bf16_round(torch.randn(33, 1024, generator=torch.Generator().manual_seed(9)) * 0.5).numpy()
        clips = [synth.synth_waveform(k, 1.0 + 0.17 * (k % 5)) for k in range(32)]
        e.set_tuning("qa", 0)
        base = _run(e, clips, emb, [11, 151643, 5, 9000, 77])
        assert len({tuple(t) for t in base[2][-1]}) > 1
        e.set_tuning("qa", 1)
        assert e.decode_structure()[0] == 1                   # the fused launch is what runs
        for early in (5, 3, 4):
            e.set_tuning("qa_early", early)
            got = _run(e, clips, emb, [11, 151643, 5, 9000, 77])
            assert np.array_equal(got[0], base[0]), early
            assert np.array_equal(got[1], base[1]), (early, float(np.abs(got[1] - base[1]).max()))
            assert got[2] == base[2], early
    finally:
        e.set_tuning("qa_early", 5)
        e.set_tuning("qa", 1)
        e.close()


def test_qa_context_split_at_small_batches(eng):
    """Up to 8 (knob qa_split 2: 16) batch rows an attention unit is spread over 8 / 4 (/ 2) workgroups: every wave keeps the chunks it swept before, the
    partials travel as tagged granules to the unit's first workgroup, whose merge is unchanged -- tokens bit-equal to one workgroup per unit and to
    the two-launch layer at every split factor and at the row counts next to the switches."""
    clips = [synth.synth_waveform(120 + k, 0.9 + 0.21 * (k % 4)) for k in range(17)]
    eng.set_tuning("chain", 0)
    try:
        for b in (1, 2, 4, 5, 8, 9, 16, 17):
            eng.set_tuning("qa", 0)
            want = eng.transcribe_batch(clips[:b], max_tokens=9, ignore_eos=True)
            eng.set_tuning("qa", 1)
            for split in (2, 1, 0):                            # 2: also two workgroups per unit at 9 .. 16 rows
                eng.set_tuning("qa_split", split)
                for _ in range(2):
                    assert eng.transcribe_batch(clips[:b], max_tokens=9, ignore_eos=True) == want, (b, split)
    finally:
        eng.set_tuning("qa_split", 1)
        eng.set_tuning("qa", 1)


def test_hand_off_buffers_survive_changing_batch_sizes(eng):
    """The granule and partial buffers are reused by every layer, step and batch; a tag is (step sequence word, layer), so what an earlier batch of
    another size left behind -- rows the current batch does not write, partials of units that are not split now -- can never read as this launch's.
    One engine, batches of 32 / 3 / 17 / 1 / 8 / 32 rows back to back, each with natural EOS, against the two-launch layer."""
    clips = [synth.synth_waveform(200 + k, 0.7 + 0.19 * (k % 6)) for k in range(32)]
    sizes = (32, 3, 17, 1, 8, 32, 5, 16)
    eng.set_tuning("chain", 0)
    try:
        eng.set_tuning("qa", 0)
        want = [eng.transcribe_batch(clips[:b], max_tokens=12) for b in sizes]
        eng.set_tuning("qa", 1)
        for _ in range(2):
            got = [eng.transcribe_batch(clips[:b], max_tokens=12) for b in sizes]
            assert got == want
    finally:
        eng.set_tuning("qa", 1)

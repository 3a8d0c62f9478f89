"""The data-parallel path on the REAL engine, one rank (the 8-GPU scaling run is the driver's; this makes sure its first
run is not also the first run of this code): qasr.dist.transcribe_sharded and qasr.dist.timed_passes -- the functions
bench.py's N > 1 path is made of -- with world size 1, and under an initialised single-rank process group whose collective
is RCCL (backend "nccl") on the device."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
from qasr import dist as qd, synth, config as QC
from qasr.model import Qwen3ASRModel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    sd = synth.synth_state_dict(QC.AUDIO_TINY, QC.TEXT_TINY, seed=3, init="stress")
    m = Qwen3ASRModel.from_state_dict(sd, preset="tiny", max_audio_seconds=4, max_new_tokens=16, max_batch=6)
    yield m
    m.close()


def _clips(ids):
    return [synth.synth_waveform(k, 0.6 + 0.2 * (k % 3)) for k in ids]


def test_sharded_and_timed_passes_world1(model):
    clips = _clips(qd.weak_scaling_clip_ids(0, 5))
    ref = model.transcribe_batch(clips, max_tokens=6, ignore_eos=True)
    assert qd.transcribe_sharded(model, clips, max_tokens=6, ignore_eos=True) == ref
    S = model.cfg.max_new_tokens + 1
    gathered = torch.empty((5, S), dtype=torch.int32, device="cuda")
    dt, lens = qd.timed_passes(model, clips, 6, steps=2, warmup=1, inclusive=True, gathered=gathered)
    assert dt > 0 and lens.tolist() == [6] * 5
    assert [r[:6] for r in gathered.cpu().tolist()] == ref
    dt2, lens2 = qd.timed_passes(model, clips, 6, steps=2, warmup=1, inclusive=False, gathered=gathered)
    assert dt2 > 0 and lens2.tolist() == [6] * 5 and [r[:6] for r in gathered.cpu().tolist()] == ref


def test_single_rank_process_group_over_rccl(model):
    """world_size 1 process group on backend nccl (= RCCL): the all_gather_into_tensor / all_reduce calls of the N > 1
    path execute on the device through the collective library."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        clips = _clips(qd.weak_scaling_clip_ids(0, 4))
        ref = model.transcribe_batch(clips, max_tokens=5, ignore_eos=True)
        assert qd.transcribe_sharded(model, clips, device="cuda", max_tokens=5, ignore_eos=True) == ref
        S = model.cfg.max_new_tokens + 1
        block = torch.arange(4 * S, dtype=torch.int32, device="cuda").reshape(4, S)
        out = torch.empty_like(block)
        dist.all_gather_into_tensor(out, block)
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert torch.equal(out, block) and float(t.item()) == 1.5
    finally:
        dist.destroy_process_group()

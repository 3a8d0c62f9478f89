"""Utterance-batch data parallelism on CPU: world_size 2 over gloo.  The per-rank engine is replaced
by a deterministic stand-in (there is no GPU here); what is tested is the product's shard / gather
logic (qasr.dist): contiguous partition, ragged tail, fixed-shape token-block all_gather, order."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from qasr import dist as qd


def test_shard_bounds_cover_everything():
    for n in (0, 1, 5, 8, 31, 256):
        for world in (1, 2, 3, 8):
            spans = [qd.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


class _FakeCfg:
    max_new_tokens = 6


class _FakeModel:
    """Token stream = f(clip) only, like the real engine (clips are independent)."""
    cfg = _FakeCfg()

    def transcribe_batch(self, clips, **opt):
        return [[int(c[0] * 1000) % 97 + i for i in range(1 + int(c[1]) % 6)] for c in clips]


def _worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clips = [np.array([0.001 * (k + 1), k], dtype=np.float32) for k in range(n_clips)]
        out = qd.transcribe_sharded(_FakeModel(), clips)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


class _FakeCtcModel:
    """wav2vec2-CTC stand-in: ragged id lists up to the frame capacity (no max_new_tokens in its config)."""
    max_len = 40

    def transcribe_batch(self, clips, **opt):
        return [[(int(c[1]) * 7 + i) % 10288 for i in range((int(c[1]) * 13) % (self.max_len + 1))] for c in clips]


def _ctc_worker(rank, world, port, n_clips, q):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        clips = [np.array([0.0, k], dtype=np.float32) for k in range(n_clips)]
        q.put((rank, qd.transcribe_sharded(_FakeCtcModel(), clips, max_len=_FakeCtcModel.max_len)))
    finally:
        dist.destroy_process_group()


def test_sharded_ctc_ids_world2():
    """The Omnilingual path shards like the Qwen3 one (clips independent); its id lists are ragged up to the frame count,
    including empty ones (a clip that collapses to nothing) and full-capacity ones."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ctc_worker, args=(r, 2, port, 9, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = _FakeCtcModel().transcribe_batch([np.array([0.0, k], dtype=np.float32) for k in range(9)])
    assert any(len(e) == 0 for e in expect) and any(len(e) == 39 for e in expect)
    assert results[0] == expect and results[1] == expect


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_clips", [1, 4, 7])
def test_sharded_transcribe_world2(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    clips = [np.array([0.001 * (k + 1), k], dtype=np.float32) for k in range(n_clips)]
    expect = _FakeModel().transcribe_batch(clips)
    assert results[0] == expect and results[1] == expect


def test_bench_sharding_path_under_torchrun(tmp_path):
    """bench.py's N > 1 path, launched the way the driver launches it (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 ...`), 2 ranks over gloo with a stand-in engine: rank-local clip ids are
    disjoint and contiguous, every rank's [B, S] token block lands at its rank offset of the gathered tensor on BOTH
    ranks, the reported time is the MAX over ranks (the slow rank's), warm-up passes are not timed."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, OUT_DIR=str(tmp_path), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_bench_driver.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    res = [json.load(open(tmp_path / f"rank{k}.json")) for k in range(2)]
    assert [x["ids"] for x in res] == [[0, 1, 2], [3, 4, 5]]
    expect = [[1000 * k + j for j in range(5)] + [-1] * 4 for k in range(6)]
    for x in res:
        assert x["world"] == 2 and x["gathered"] == expect and x["lens"] == [5, 5, 5]
        assert x["begun"] == 3 + 1 and x["ran"] == 3 + 1            # 1 warm-up + 2 timed inclusive passes, then 1 resident
        assert x["staged"] == 3                                     # pipelined loop: every pass staged its successor's clips
        assert x["dt"] >= 2 * 0.20 and x["dt"] < 2 * 0.20 + 0.5       # two passes of the SLOW rank, warm-up excluded
    assert abs(res[0]["dt"] - res[1]["dt"]) < 1e-9                    # MAX-reduced: identical on both ranks
    assert abs(res[0]["dt_res"] - res[1]["dt_res"]) < 1e-9 and res[0]["dt_res"] >= 0.20

"""Launched by tests/test_dist_cpu.py under `python -m torch.distributed.run` (2 ranks, gloo, CPU): drives the code path
bench.py takes for N > 1 -- qasr.dist.weak_scaling_clip_ids, qasr.dist.timed_passes (barrier / all_gather of the
[B, S] int32 token block / MAX-reduce of the elapsed time) -- with a stand-in engine, because there is no GPU here.
Each rank writes what it saw to $OUT_DIR/rank<r>.json."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))
from qasr import dist as qd  # noqa: E402

S, N_DEC = 9, 5


class _Cfg:
    max_new_tokens = S - 1


class StandInEngine:
    """Same surface as qasr.model.Qwen3ASRModel's split batch API; tokens are a function of the clip id only."""
    cfg = _Cfg()

    def __init__(self, delay):
        self.delay, self.begun, self.ran, self.staged = delay, 0, 0, 0

    def batch_begin(self, clips, max_tokens, ignore_eos):
        self.ids = [int(c[0]) for c in clips]
        self.begun += 1

    def batch_stage(self, clips):                       # the pipelined loop of bench.py: next batch staged under the current one
        self.staged_ids = [int(c[0]) for c in clips]
        self.staged += 1

    def batch_begin_staged(self, max_tokens, ignore_eos):
        self.ids = self.staged_ids
        self.begun += 1

    def batch_rewind(self):
        pass

    def batch_sync(self):
        pass

    def batch_run(self):
        time.sleep(self.delay)
        self.ran += 1

    def batch_tokens(self):
        toks = np.full((len(self.ids), S), -1, dtype=np.int32)
        for i, k in enumerate(self.ids):
            toks[i, :N_DEC] = [1000 * k + j for j in range(N_DEC)]
        return toks, np.full(len(self.ids), N_DEC, dtype=np.int32)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    B = 3
    ids = qd.weak_scaling_clip_ids(rank, B)
    clips = [np.array([k], dtype=np.float32) for k in ids]
    eng = StandInEngine(delay=0.05 if rank == 0 else 0.20)          # rank 1 is the slow one: dt must be ITS time on both
    gathered = torch.empty((world * B, S), dtype=torch.int32)
    dt, lens = qd.timed_passes(eng, clips, N_DEC, steps=2, warmup=1, inclusive=True, gathered=gathered, pipelined=True)
    dt_res, _ = qd.timed_passes(eng, clips, N_DEC, steps=1, warmup=0, inclusive=False, gathered=gathered)
    json.dump({"rank": rank, "world": world, "ids": ids, "dt": dt, "dt_res": dt_res, "lens": lens.tolist(),
               "gathered": gathered.tolist(), "begun": eng.begun, "ran": eng.ran, "staged": eng.staged},
              open(os.path.join(os.environ["OUT_DIR"], f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

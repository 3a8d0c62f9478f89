#!/usr/bin/env python3
"""Omnilingual ASR (wav2vec2 + CTC, BASELINE configs[3]) throughput on one MI355X: B clips x S seconds, synthetic
waveforms + seeded random weights of the named variant (float checkpoints are cast to bf16 MFMA operands by the engine,
MLX-quantised ones are expanded to bf16(scale*q+bias): same device work), timed region = pcm in host memory -> collapsed
token ids in host memory.  Prints one JSON line.

usage: python scratch/bench_ctc.py [--variant 300M|1B|3B|7B] [--batch 32] [--seconds 30] [--steps 3]
FLOPs counted: conv stack 2*C*k*C_in per output frame of each layer, projection, positional conv 2*D*KP*cpg, per layer
2*(4 D^2 + 2 D F) + 4 T D attention, head 2 D V -- per encoder frame."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from qasr import synth  # noqa: E402
from qasr.omnilingual import OmnilingualASRMLXModel  # noqa: E402

KERNELS, STRIDES = (10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)


def flops_per_clip(cfg, n):
    C, D, F, V = cfg.feature_dim, cfg.model_dim, cfg.ffn_dim, cfg.vocab
    L, total = n, 0.0
    for i, (k, s) in enumerate(zip(KERNELS, STRIDES)):
        L = (L - k) // s + 1
        total += 2.0 * L * C * k * (1 if i == 0 else C)
    T = L
    cpg = D // cfg.pos_groups
    total += 2.0 * T * D * C + 2.0 * T * D * cfg.pos_kernel * cpg
    total += cfg.layers * (2.0 * T * (4 * D * D + 2 * D * F) + 4.0 * T * T * D)
    total += 2.0 * T * D * V
    return total, T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="300M")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--ab", default="", help="knob=v1,v2,...: repeat the timed region per value of one tuning knob")
    ap.add_argument("--gpus", type=int, default=1, help="ranks (launch with torch.distributed.run for > 1): --batch clips PER GPU, "
                                                          "clips sharded, one all_gather of the padded id block per pass")
    a = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if "QASR_BENCH_DEVICE" in os.environ:                       # rehearsal of the N > 1 path on a one-GPU box (gloo, ranks share the card)
        local = int(os.environ["QASR_BENCH_DEVICE"])
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl" if os.environ.get("QASR_DIST_BACKEND", "nccl") == "nccl" else "gloo")
    m = OmnilingualASRMLXModel(variant=a.variant, device=local if world > 1 else 0, max_batch=a.batch,
                               max_audio_seconds=int(np.ceil(a.seconds)))
    cfg = m.cfg
    t0 = time.perf_counter()
    sd = synth.synth_omnilingual_state_dict(cfg, seed=0, bits=0)
    import ctypes as C
    for name, t in sd.items():
        t = t.contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        m._check(m.lib.qasr_ctc_set_tensor(m.h, name.encode(), C.c_void_p(t.data_ptr()), 0, shape, t.dim()))
    del sd
    m._check(m.lib.qasr_ctc_finalize(m.h))
    print(f"[bench_ctc] weights built + uploaded in {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
    clips = [synth.synth_waveform(k, a.seconds) for k in range(rank * a.batch, (rank + 1) * a.batch)]   # weak scaling: own clips
    if world > 1:
        from qasr import dist as qdist
        S = m.num_frames(len(clips[0])) + 1
        dev = torch.device("cuda", local) if dist.get_backend() == "nccl" else torch.device("cpu")
        gathered = torch.empty((world * a.batch, S), dtype=torch.int32, device=dev)

        def one_pass():
            ids = m.transcribe_batch(clips)
            block = np.full((a.batch, S), -1, np.int32)
            for i, t in enumerate(ids):
                block[i, :len(t)] = t
                block[i, S - 1] = len(t)
            dist.all_gather_into_tensor(gathered, torch.from_numpy(block).to(dev))
            return ids
        one_pass()
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ids = one_pass()
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item()) / a.steps
        if rank == 0:
            print(json.dumps({"metric": f"audio-seconds/sec Omnilingual-ASR-CTC-{a.variant}, {a.seconds:.0f} s @ 16 kHz, b={a.batch} per GPU",
                              "value": round(world * a.batch * a.seconds / dt, 1), "n_gpus": world, "ms_per_step": round(dt * 1e3, 2),
                              "scaling": "weak", "data": "synthetic", "gathered_rows": int((gathered[:, S - 1] >= 0).sum())}), flush=True)
        m.close()
        dist.destroy_process_group()
        return
    if a.ab:
        key, vals = a.ab.split("=")
        for v in vals.split(","):
            m.lib.qasr_set_tuning(key.encode(), int(v))
            m.transcribe_batch(clips)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                m.transcribe_batch(clips)
            dt = (time.perf_counter() - t0) / a.steps
            print(f"[bench_ctc] {key}={v}: {dt * 1e3:.2f} ms/step, stages {[round(x, 2) for x in m.timings()]}", file=sys.stderr, flush=True)
    m.transcribe_batch(clips)                                   # warm-up
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ids = m.transcribe_batch(clips)
    dt = (time.perf_counter() - t0) / a.steps
    ms = m.timings()
    fl, T = flops_per_clip(cfg, len(clips[0]))
    out = {"metric": f"audio-seconds/sec Omnilingual-ASR-CTC-{a.variant}, {a.seconds:.0f} s @ 16 kHz, b={a.batch}, 1 GPU",
           "value": round(a.batch * a.seconds / dt, 1), "ms_per_step": round(dt * 1e3, 2),
           "stage_ms": {"frontend": round(ms[0], 2), "transformer": round(ms[1], 2), "head_argmax": round(ms[2], 2), "device_total": round(ms[3], 2)},
           "frames_per_clip": T, "tflop_per_step": round(fl * a.batch / 1e12, 2),
           "mfma": {"achieved_tflops": round(fl * a.batch / (ms[3] / 1e3) / 1e12, 1), "peak": 2500.0,
                    "frac": round(fl * a.batch / (ms[3] / 1e3) / 1e12 / 2500.0, 4)},
           "ids_per_clip": [len(x) for x in ids[:4]], "data": "synthetic", "dtype": "bf16 MFMA operands, f32 residual / LayerNorm / softmax"}
    print(json.dumps(out), flush=True)
    m.close()


if __name__ == "__main__":
    main()

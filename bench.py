#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Qwen3-ASR transcribe() hot path on MI355X.

Metric (BASELINE.json): audio-seconds / wall-second (RTF^-1), Qwen3-ASR-0.6B, 30 s @ 16 kHz clips,
batch 32 per GPU (the metric's "b=32"; 8 GPUs x 32 = configs[2]'s 256 clips), bf16, synthetic
waveforms + seeded random weights of the real architecture (no checkpoint / dataset offline).

A "step" = one pass of the whole hot path (log-mel -> audio encoder -> prompt pass -> N_dec greedy
decode steps -> token ids on host -> RCCL all_gather of the [B, 449] int32 token block) over one batch
whose PCM sits in host memory when the pass starts (qasr_batch_begin: pinned staging + H2D + planning, all
inside the timed region, SURVEY.md section 8d "pcm on host -> token ids on host"); `resident_value` is
the same pass with the batch already in HBM.  Decode length is forced to
N_dec = 128 tokens per clip (EOS ignored) so the work is deterministic -- SURVEY.md section 8(d).

Launch:  python bench.py [--gpus 1]            or, for N > 1 (one rank per GPU, RCCL):
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))

import numpy as np          # noqa: E402
import torch                # noqa: E402
import torch.distributed as dist  # noqa: E402

from qasr import _lib, synth          # noqa: E402
from qasr import dist as qdist        # noqa: E402
from qasr.model import Qwen3ASRModel  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6300 GB/s achievable)


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def host_cpu_info():
    """Physical cores / logical CPUs of the node and the CPUs this process may use (core count stated)."""
    logical = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = logical
    cores, model = set(), None
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif k == "model name" and model is None:
                model = v
            elif not k and phys is not None and core is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    quota = None                       # cgroup CPU quota of this container (cpu.max: "<quota> <period>" or "max <period>")
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    return {"physical_cores": len(cores) or logical, "logical_cpus": logical, "affinity_cpus": affinity,
            "cgroup_cpu_quota": None if quota is None else round(quota, 2), "model": model}


def cpu_baseline(sd, pcm, n_dec):
    """CPU restatement of the reference (oracle/, B=1 sequential like the Swift reference) timed on
    this host's cores on ONE clip of the benchmark batch.  Checker code, used here only as the
    reported baseline -- never on the product path."""
    from oracle import config as OC, pipeline, precision as P
    host = host_cpu_info()
    avail = host["affinity_cpus"]
    # every physical core this process may run on (SMT siblings add nothing to an fp32 BLAS workload); a GPU box grants a share of
    # its cores per GPU, so the affinity mask, not the socket, is what "the node's own host CPU cores" means for one rank
    threads = max(1, min(avail, host["physical_cores"]))
    if host["cgroup_cpu_quota"]:       # a container's share of the node (the GPU boxes grant 16 CPUs per GPU): more threads than that only thrash
        threads = max(1, min(threads, int(host["cgroup_cpu_quota"])))
    threads = min(threads, 64)         # B = 1 fp32 GEMMs at these widths stop scaling well before that; `host` states the node's real counts
    torch.set_num_threads(threads)
    model = pipeline.OracleModel(sd, OC.AUDIO_SMALL, OC.TEXT_SMALL, OC.TOKENS, P.REFERENCE)
    # bounded sample: the first two thirds of one clip with two thirds of the forced tokens, ~10 s of CPU work on 16 cores
    # (cost is linear in clip length: 100-frame conv chunks, 104-token attention windows, prompt length 16 + 13/s)
    num, frac = 2, 3
    pcm = pcm[: len(pcm) * num // frac]
    n_dec = max(1, n_dec * num // frac)
    model.W("model.embed_tokens.weight")   # materialise the f32 view of the tied head outside the timing
    log(f"cpu_baseline: {threads} threads, {len(pcm) / 16000.0:.1f} s clip, {n_dec} tokens ...")
    t0 = time.perf_counter()
    toks = model.transcribe_tokens(pcm, max_tokens=n_dec, ignore_eos=True)
    dt = time.perf_counter() - t0
    assert len(toks) == n_dec
    return {"value": round(len(pcm) / 16000.0 / dt, 3), "unit": "audio-seconds/sec", "cores": threads,
            "host": host, "kind": "port",
            "sample": f"{num}/{frac} of one of the batch's clips ({len(pcm) / 16000.0:.0f} s, {n_dec} forced tokens), B=1 "
                      f"sequential like the reference, fp32 torch-CPU restatement (not the Swift binary), "
                      f"{dt:.1f} s of CPU work"}


MFMA_PEAK_TFLOPS = 2500.0   # MI355X dense bf16 (MI355X_MICROARCH.md; the 2:1-sparsity figure is not used)


def stage_roofline(B, seconds, n_dec, stage_ms, steps, bits=16):
    """Per-stage achieved rate of the last timed pass against the bound SURVEY.md section 8(d) names for it, from the
    survey's algorithmic work per 30 s clip (scaled linearly with the clip length) and the HIP-event stage times."""
    k = seconds / 30.0
    audio_tok = 13.0 * seconds                       # 390 audio tokens per 30 s
    prompt = 16 + audio_tok
    mel_b = 3.456e6 * k * B
    enc_f = 270.7e9 * k * B
    pre_f = 376.8e9 * k * B                          # (mildly superlinear in T through attention; 30 s is the quoted case)
    # decode: weights once per step for the whole batch + every row's K/V rows; the first token comes from the prompt pass
    # weights: 596.0 M parameters x 2 B (bf16) or bits / 8 + (scale + bias, bf16) / 64 per parameter (MLX group 64)
    w_b = 1.192e9 if bits == 16 else 596.0e6 * (bits / 8.0 + 4.0 / 64.0)
    dec_b = steps * w_b + B * 114688.0 * sum(prompt + i for i in range(steps))
    out = {"mel": {"bound": "hbm", "achieved": round(mel_b / stage_ms[0] / 1e6, 1), "unit": "GB/s", "frac": round(mel_b / stage_ms[0] / 1e6 / HBM_PEAK_GBS, 4)},
           "encoder": {"bound": "mfma", "achieved": round(enc_f / stage_ms[1] / 1e9, 1), "unit": "TFLOP/s",
                       "frac": round(enc_f / stage_ms[1] / 1e9 / MFMA_PEAK_TFLOPS, 4)},
           "prompt_pass": {"bound": "mfma", "achieved": round(pre_f / stage_ms[2] / 1e9, 1), "unit": "TFLOP/s",
                           "frac": round(pre_f / stage_ms[2] / 1e9 / MFMA_PEAK_TFLOPS, 4)}}
    if steps > 0 and stage_ms[3] > 0:
        out["decode"] = {"bound": "hbm", "achieved": round(dec_b / stage_ms[3] / 1e6, 1), "unit": "GB/s",
                         "frac": round(dec_b / stage_ms[3] / 1e6 / HBM_PEAK_GBS, 4),
                         "bytes_per_step": round(dec_b / steps)}
    return out


def kernel_source_stamp(files=("dec_attention.hip", "dec_rope.h", "dec_kernels.h", "common.h")):
    """sha256 (first 16 hex) of a kernel's sources (its own translation unit and the headers it includes; default: the decode attention):
    a PMC traffic file is only quoted for the kernels it was measured on (profiles/*_pmc_traffic.json carries the stamp of the build it
    was taken from)."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "qwen3-asr-swift_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


GEMV_SOURCES = ("dec_gemv.hip", "dec_epilogue.h", "dec_kernels.h", "common.h")
QA_SOURCES = ("dec_qa.hip", "dec_chain_dev.h", "dec_chain.h", "dec_rope.h", "dec_epilogue.h", "dec_kernels.h", "common.h")

# Omnilingual-ASR-CTC (BASELINE configs[3]): FLOPs per clip as scratch/bench_ctc.py counts them -- conv stack 2 C k C_in per output frame
# of each layer, projection, positional conv, per layer 2 (4 D^2 + 2 D F) + 4 T D attention, head 2 D V per encoder frame
W2V_KERNELS, W2V_STRIDES = (10, 3, 3, 3, 3, 2, 2), (5, 2, 2, 2, 2, 2, 2)


def omnilingual_flops(cfg, n):
    C_, D, F, V = cfg.feature_dim, cfg.model_dim, cfg.ffn_dim, cfg.vocab
    L, total = n, 0.0
    for i, (k, st) in enumerate(zip(W2V_KERNELS, W2V_STRIDES)):
        L = (L - k) // st + 1
        total += 2.0 * L * C_ * k * (1 if i == 0 else C_)
    T = L
    total += 2.0 * T * D * C_ + 2.0 * T * D * cfg.pos_kernel * (D // cfg.pos_groups)
    total += cfg.layers * (2.0 * T * (4 * D * D + 2 * D * F) + 4.0 * T * T * D)
    total += 2.0 * T * D * V
    return total, T


def omnilingual_leg(variant, B, seconds, steps, device):
    """configs[3]: `B` clips x `seconds` s through the wav2vec2-CTC engine of the named variant (seeded random weights streamed in, synthetic
    waveforms), timed region = pcm in host memory -> collapsed token ids in host memory; MFMA fraction of the whole pass."""
    from qasr.omnilingual import OmnilingualASRMLXModel
    t0 = time.perf_counter()
    m = OmnilingualASRMLXModel.from_synthetic(variant=variant, device=device, max_batch=B, max_audio_seconds=int(np.ceil(seconds)))
    log(f"omnilingual {variant}: weights built + uploaded in {time.perf_counter() - t0:.1f} s")
    try:
        clips = [synth.synth_waveform(k, seconds) for k in range(B)]
        m.transcribe_batch(clips)                                # warm-up
        t0 = time.perf_counter()
        for _ in range(steps):
            ids = m.transcribe_batch(clips)
        dt = (time.perf_counter() - t0) / steps
        ms = m.timings()
        fl, T = omnilingual_flops(m.cfg, len(clips[0]))
        return {"metric": f"audio-seconds/sec Omnilingual-ASR-CTC-{variant}, {seconds:.0f} s @ 16 kHz, b={B}, 1 GPU (BASELINE configs[3])",
                "value": round(B * seconds / dt, 1), "unit": "audio-seconds/sec", "steps": steps, "ms_per_step": round(dt * 1e3, 2),
                "stage_ms": {"frontend": round(ms[0], 2), "transformer": round(ms[1], 2), "head_argmax": round(ms[2], 2), "device_total": round(ms[3], 2)},
                "frames_per_clip": T, "tflop_per_step": round(B * fl / 1e12, 2),
                "roofline": {"bound": "mfma", "achieved": round(B * fl / dt / 1e12, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(B * fl / dt / 1e12 / MFMA_PEAK_TFLOPS, 4),
                             "note": "whole pass (front end, attention, LayerNorm, head and host collapse included) against the dense bf16 MFMA peak"},
                "ids_per_clip": [len(t) for t in ids[:4]], "data": "synthetic", "dtype": "bf16 MFMA operands, f32 residual / LayerNorm / softmax",
                "parameters": f"{variant} (layers {m.cfg.layers}, model_dim {m.cfg.model_dim}, ffn {m.cfg.ffn_dim}, heads {m.cfg.heads})"}
    finally:
        m.close()


def lanes_leg(sd, clips, n_dec, steps, seconds, cap, lanes):
    """The same passes with `lanes` of them in flight on the one GPU (qasr_dp_submit / qasr_dp_collect over engines that share the device):
    pass k runs whole on engine k % lanes, host pcm -> host tokens, while the passes before it are still decoding.  Reported beside the
    headline: `lanes` batches of the metric's size are resident at a time, so it is a serving-loop figure, not BASELINE's single batch."""
    from qasr.dp import Qwen3ASRDataParallel
    dp = Qwen3ASRDataParallel.from_state_dict(sd, [cap["device"]] * lanes, preset="0.6B", **cap)

    def run(k):
        pending = []
        for _ in range(k):
            if len(pending) == lanes:
                _, lens = dp.collect(pending.pop(0), raw=True)
                assert (lens == n_dec).all(), lens
            pending.append(dp.submit(clips, max_tokens=n_dec, ignore_eos=True))
        for t in pending:
            _, lens = dp.collect(t, raw=True)
            assert (lens == n_dec).all(), lens

    try:
        run(lanes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        dp.close()
    return {"value": round(len(clips) * seconds * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "passes_in_flight": lanes,
            "note": f"{steps} passes of the headline workload, {lanes} in flight on the one GPU (one engine + host thread each, weights "
                    f"replicated); timed from the first submit to the last collect, fill and drain included"}


def run_leg(model, clips, n_dec, steps, warmup, world, gathered, inclusive, pipelined=False):
    """K timed passes (qasr.dist.timed_passes: barrier + synchronize on both sides, MAX over ranks)."""
    dt, lens = qdist.timed_passes(model, clips, n_dec, steps, warmup, inclusive, gathered, pipelined)
    assert (lens == n_dec).all(), lens
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--decode-tokens", type=int, default=128)
    ap.add_argument("--bits", type=int, default=16, choices=[16, 8, 4],
                    help="weights of the HEADLINE engine: 16 = bf16 (BASELINE's config), 4 / 8 = MLX-quantised decoder")
    ap.add_argument("--lanes", type=int, default=3, help="passes in flight on the GPU in the side leg `passes_in_flight` (qasr_dp_submit; 1: 5650, 2: 7440, 3: 7880, 4: 7260 audio-s/s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the b=1 / b=8, MLX-4bit and Omnilingual legs (profiling runs)")
    ap.add_argument("--omnilingual", default="300M,7B", help="comma-separated Omnilingual-ASR-CTC variants for the configs[3] legs ('' = none)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    B, n_dec = args.batch, args.decode_tokens
    log(f"rank {rank}/{world}: generating synthetic weights ...")
    from qasr import config as QC
    sd = synth.synth_state_dict(QC.AUDIO_SMALL, QC.TEXT_SMALL, seed=0, init="hf")
    cap = dict(device=local_rank, max_batch=B, max_audio_seconds=int(np.ceil(args.seconds)), max_new_tokens=448)

    quantised = {}

    def weights(bits):
        if bits == 16:
            return sd
        if bits not in quantised:
            quantised[bits] = synth.quantize_state_dict(sd, bits)
        return quantised[bits]

    def build(bits):
        return Qwen3ASRModel.from_state_dict(weights(bits), preset="0.6B", bits=bits, **cap)

    model = build(args.bits)
    # weak scaling: every rank gets its own B clips (clip ids rank*B ...), no data-path collective
    clips = [synth.synth_waveform(k, args.seconds) for k in qdist.weak_scaling_clip_ids(rank, B)]
    stride = model.cfg.max_new_tokens + 1
    gathered = torch.empty((world * B, stride), dtype=torch.int32, device="cuda")

    log("weights resident; timing host pcm -> host tokens ...")
    # headline: consecutive passes as a serving loop runs them -- pass i + 1's clips are staged from the caller's host buffers and copied
    # to HBM while pass i's kernels run (qasr_batch_stage / qasr_batch_begin_staged); every pass starts at host buffers and ends with
    # token ids on the host, K passes = K stagings + K uploads + K full computations
    dt = run_leg(model, clips, n_dec, args.steps, args.warmup, world, gathered, inclusive=True, pipelined=True)
    log(f"timed {args.steps} steps in {dt * 1e3:.1f} ms")
    stage_ms, steps_done = model.batch_timings()
    # the same passes strictly one after the other (staging + H2D + planning exposed in front of every pass)
    dt_serial = run_leg(model, clips, n_dec, args.steps, 1, world, gathered, inclusive=True)
    # the same passes with the batch already resident in HBM (no staging copy / H2D / planning in the timed region)
    dt_res = run_leg(model, clips, n_dec, args.steps, 1, world, gathered, inclusive=False)
    probe = {name: model.kernel_probe(which, 20) for which, name in ((0, "layer_gemv"), (1, "decode_attn"), (2, "lm_head"))}
    # how the step of this batch is launched: with fused_qa a layer's q|k|v projection and attention are ONE launch (csrc/dec_qa.hip), which
    # probe 1 then times (K / V rows + the q|k|v weights), and probe 0 is the three linears left
    fused_qa, chain_mode, launches_per_layer = model.decode_structure()

    extras = {}
    if world == 1 and not args.no_extras:
        # BASELINE's metric is quoted at batch {1, 8, 32}: the smaller batches on the same engine, same timed region
        for b in (1, 8):
            if b < B:
                d = run_leg(model, clips[:b], n_dec, 3, 1, 1, None, inclusive=True, pipelined=True)
                ms, _ = model.batch_timings()
                extras[f"b{b}"] = {"value": round(b * args.seconds * 3 / d, 1), "ms_per_step": round(d / 3 * 1e3, 3),
                                   "stage_ms": {"mel": round(ms[0], 3), "encoder": round(ms[1], 3), "prompt_pass": round(ms[2], 3),
                                                "decode": round(ms[3], 3)}}
                log(f"b={b}: {extras[f'b{b}']}")

    if rank == 0:
        audio_s = world * B * args.seconds * args.steps
        ms_step = dt / args.steps * 1e3
        # dominant kernel by GPU time (profiles/*_bench_kernel_stats.csv): decode_attention_mfma_kernel
        dom_ms, dom_bytes = probe["decode_attn"]
        traffic, traffic_note = None, None
        import glob
        stamp = kernel_source_stamp(QA_SOURCES) if fused_qa else kernel_source_stamp()
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            pmc = json.load(open(f))
            if pmc.get("qa_source_stamp" if fused_qa else "kernel_source_stamp") != stamp:
                continue                   # measured on other kernel sources: stale, not quoted
            traffic = pmc.get("decode_qa_bytes" if fused_qa else "decode_attention_bytes")
            alg = pmc.get("decode_qa_algorithmic_bytes_at_that_context" if fused_qa else "decode_attention_algorithmic_bytes_at_that_context")
            traffic_note = (f"profiles/{os.path.basename(f)} (kernel sources {stamp}): (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch, "
                            f"separate --pmc passes of a 16-token run; algorithmic bytes at that run's mean context: {alg}")
            break
        if traffic is None:
            traffic_note = f"no PMC file under profiles/ matches the current kernel sources ({stamp}); not quoted"
        gemv_traffic, gemv_traffic_note = None, None
        gstamp = kernel_source_stamp(GEMV_SOURCES)
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            pmc = json.load(open(f))
            if pmc.get("gemv_source_stamp") != gstamp or not pmc.get("layer_gemv_group_bytes"):
                continue
            gemv_traffic = pmc["layer_gemv_group_bytes"]
            if bool(pmc.get("fused_qa")) != bool(fused_qa):
                continue                   # the group had another number of launches in that run
            gemv_traffic_note = (f"profiles/{os.path.basename(f)} (kernel sources {gstamp}): (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over the "
                                 f"{'three' if fused_qa else 'four'} launches of a layer, separate --pmc passes of a 16-token run")
            break
        if gemv_traffic is None:
            gemv_traffic_note = f"no PMC file under profiles/ matches the current GEMV kernel sources ({gstamp}); not quoted"
        # which kernel dominates the GPU time of a pass: the decode attention (one launch per layer and step) or the decode-step GEMV
        # kernel (decode_gemv2_kernel, four launches per layer and step: q|k|v, o-proj, gate|up, down).  Shares from the live probes
        # (HIP events on the engine stream; the GEMV probe walks the layers so that every launch streams its weights from HBM as in situ)
        n_layers = model.cfg.dec_layers
        res_ms = dt_res / args.steps * 1e3
        gemv_ms, gemv_bytes = probe["layer_gemv"]
        share_attn = n_layers * steps_done * dom_ms / res_ms
        share_gemv = n_layers * steps_done * gemv_ms / res_ms
        n_gemv = 3 if fused_qa else 4
        gemv_family = {"kernel": (f"decode_gemv2_kernel x{n_gemv} per layer (" + ("" if fused_qa else "q|k|v + ") + "o-proj + gate|up + down, streamed once per "
                                  "step for all batch rows" + ("; the q|k|v projection runs inside the attention's launch" if fused_qa else "") + ")"),
                       "achieved": round(gemv_bytes / gemv_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(gemv_bytes / gemv_ms / 1e6 / HBM_PEAK_GBS, 4), "bytes_per_layer": gemv_bytes,
                       "avg_ms_per_layer": round(gemv_ms, 5), "avg_ms_per_launch": round(gemv_ms / n_gemv, 5),
                       "share_of_gpu_time": round(share_gemv, 3)}
        wname = "bf16" if args.bits == 16 else f"MLX {args.bits}-bit decoder (packed in HBM), bf16 activations"
        out = {
            "metric": "audio-seconds/sec (RTF^-1) Qwen3-ASR-0.6B, 30 s@16 kHz, b=32 per GPU",
            "value": round(audio_s / dt, 1),
            "unit": "audio-seconds/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"Qwen3-ASR-0.6B geometry ({wname}), {B} clips x {args.seconds:.0f} s per GPU "
                                   f"(BASELINE metric's b=32; x8 GPUs = configs[2]), timed region = pcm in host memory -> "
                                   f"mel + encoder + prompt pass + {n_dec} forced greedy tokens -> token ids in host memory, "
                                   f"tokens gathered over RCCL",
                       "clips_per_gpu": B, "clip_seconds": args.seconds, "decode_tokens": n_dec, "weights": wname,
                       "sharding": f"dp{world} (independent clips, weights replicated)"},
            "rtf": round(dt / audio_s, 7),
            "stage_ms": {"mel": round(stage_ms[0], 3), "encoder": round(stage_ms[1], 3),
                         "prompt_pass": round(stage_ms[2], 3), "decode": round(stage_ms[3], 3),
                         "decode_steps": steps_done},
            "timed_region": "K consecutive passes pipelined like a serving loop: the next pass's host -> pinned -> HBM staging runs under the "
                            "current pass's kernels (qasr_batch_stage); each pass = host pcm -> mel + encoder + prompt pass + decode -> host tokens",
            "serial_inclusive_value": round(audio_s / dt_serial, 1),
            "serial_inclusive_note": "the same passes one strictly after the other: staging copy + H2D + planning exposed in front of every pass",
            "resident_value": round(audio_s / dt_res, 1),
            "resident_note": "same passes with the batch already in HBM (qasr_batch_rewind): no staging copy / H2D / planning timed",
            "stage_roofline": stage_roofline(B, args.seconds, n_dec, stage_ms, steps_done, args.bits),
        }
        attn_obj = {"bound": "hbm",
                    "kernel": ("decode_qa_kernel (one launch = one decoder layer's q|k|v projection AND attention for all batch rows: the K and V "
                               "rows of every row's context and the 8.39 MB of q|k|v weights are streamed once; the stream is requested while the "
                               "projection runs)" if fused_qa else
                               "decode_attention_mfma_kernel (one launch = one decoder layer's attention for all batch rows: "
                               "K and V rows of every row's context are streamed once)"),
                    "achieved": round(dom_bytes / dom_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(dom_bytes / dom_ms / 1e6 / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": traffic_note,
                    "bytes_per_launch": dom_bytes, "avg_ms_per_launch": round(dom_ms, 5),
                    "share_of_gpu_time": round(share_attn, 3),
                    "how": ("algorithmic bytes = sum_b 2 (K,V) x 8 kv heads x 128 x 2 B x ctx_b at the probe's context + 4096 x 1024 x 2 B of q|k|v "
                            "weights; duration = HIP events on the engine stream around each of 20 launches walking the 28 layers, so weights and "
                            "K / V come from HBM as in the step (qasr_kernel_probe)" if fused_qa else
                            "algorithmic bytes = sum_b 2 (K,V) x 8 kv heads x 128 x 2 B x ctx_b at the probe's context; duration = HIP "
                            "events on the engine stream around each of 20 launches, each preceded by an untimed weight-streaming "
                            "launch as in the real step (qasr_kernel_probe)")}
        other = {k: {"avg_ms": round(v[0], 5), "bytes": v[1], "GBps": round(v[1] / v[0] / 1e6, 1)} for k, v in probe.items()}
        if share_gemv > share_attn:
            # the GEMV kernel takes more of the pass than the attention: IT is the dominant kernel and the object's headline figure;
            # the attention (closer to its roofline) is kept beside it, never instead of it
            out["roofline"] = {"bound": "hbm", **gemv_family, "traffic": gemv_traffic, "traffic_source": gemv_traffic_note,
                               "how": "algorithmic bytes = the bf16 weights the group streams (q|k|v 8.39, o-proj 4.19, gate|up 12.58, down 6.29 MB: those of its "
                                      "launches); duration = HIP events on the engine stream around 20 groups walking the 28 layers (weights from HBM, as in the step)",
                               "attention": attn_obj, "other": other}
        else:
            out["roofline"] = {**attn_obj, "family": {**gemv_family, "traffic": gemv_traffic, "traffic_source": gemv_traffic_note}, "other": other}
        # the whole pass against its roofline (MFMA part at the dense bf16 peak + decode bytes at the HBM peak), and the strictly
        # serial inclusive figure beside the pipelined headline -- inside the object the driver keeps
        sr = out["stage_roofline"]
        bound_ms = ((270.7e9 + 376.8e9) * (args.seconds / 30.0) * B / (MFMA_PEAK_TFLOPS * 1e12) * 1e3
                    + sr.get("decode", {}).get("bytes_per_step", 0) * steps_done / (HBM_PEAK_GBS * 1e9) * 1e3)
        out["roofline"]["whole_pass_frac"] = round(bound_ms / ms_step, 4)
        out["roofline"]["whole_pass_bound_ms"] = round(bound_ms, 2)
        out["roofline"]["serial_inclusive_value"] = out["serial_inclusive_value"]
        out["roofline"]["decode_structure"] = {"fused_qkv_attention": bool(fused_qa), "chain": chain_mode, "dependent_launches_per_layer": launches_per_layer}
        if extras:
            out["batches"] = extras
    model.close()
    if rank == 0 and world == 1 and not args.no_extras and args.bits == 16:
        # the reference's shipped checkpoints are MLX 4-bit: same workload on a synthetic 4-bit checkpoint,
        # packed weights in HBM (csrc/dec_quant.hip).  Reported beside the bf16 headline, never as `value`.
        log("building the MLX-4bit engine ...")
        m4 = build(4)
        d4 = run_leg(m4, clips, n_dec, args.steps, 1, 1, None, inclusive=True, pipelined=True)
        ms4, st4 = m4.batch_timings()
        p4 = {name: m4.kernel_probe(which, 20) for which, name in ((0, "layer_gemv"), (2, "lm_head"))}
        fq4, ch4, lpl4 = m4.decode_structure()
        out["mlx_4bit"] = {"value": round(B * args.seconds * args.steps / d4, 1), "ms_per_step": round(d4 / args.steps * 1e3, 3),
                           "stage_ms": {"mel": round(ms4[0], 3), "encoder": round(ms4[1], 3), "prompt_pass": round(ms4[2], 3),
                                        "decode": round(ms4[3], 3), "decode_steps": st4},
                           "stage_roofline_decode": stage_roofline(B, args.seconds, n_dec, ms4, st4, 4).get("decode"),
                           "kernels": {k: {"avg_ms": round(v[0], 5), "bytes": v[1], "GBps": round(v[1] / v[0] / 1e6, 1)} for k, v in p4.items()},
                           "decode_structure": {"fused_qkv_attention": bool(fq4), "dependent_launches_per_layer": lpl4,
                                                "layer_gemv_covers": "o-proj + gate|up + down" if fq4 else "q|k|v + o-proj + gate|up + down"},
                           "note": "synthetic weights quantised with mlx's affine scheme (group 64); decode-step products in the "
                                   "reference's f32-dequantised form, prompt pass on bf16(scale*q+bias) like its many-row kernel"}
        m4.close()
    if rank == 0 and world == 1 and not args.no_extras and args.bits == 16:
        try:
            out["passes_in_flight"] = lanes_leg(sd, clips, n_dec, max(args.steps, 3 * args.lanes), args.seconds, cap, args.lanes)
            log(f"{args.lanes} passes in flight: {out['passes_in_flight']['value']} audio-s/s")
            # the same with the MLX 4-bit decoder: once launches of several passes overlap, the bytes they stream count again
            q = lanes_leg(weights(4), clips, n_dec, max(args.steps, 3 * args.lanes), args.seconds, dict(cap, bits=4), args.lanes)
            out["passes_in_flight"]["mlx_4bit"] = {k: q[k] for k in ("value", "ms_per_step", "steps")}
            log(f"{args.lanes} passes in flight, MLX 4-bit: {q['value']} audio-s/s")
        except Exception as ex:          # noqa: BLE001 -- a failed side leg must not lose the headline line
            out["passes_in_flight"] = {"error": str(ex)}
    if rank == 0 and world == 1 and not args.no_extras and args.omnilingual:
        # BASELINE configs[3]: the wav2vec2-CTC family at the same batch x clip length, one engine at a time (the Qwen3 engines are closed)
        out["omnilingual"] = {}
        for variant in [v for v in args.omnilingual.split(",") if v]:
            try:
                out["omnilingual"][variant] = omnilingual_leg(variant, B, min(args.seconds, 40.0), 3, local_rank)
                log(f"omnilingual {variant}: {out['omnilingual'][variant]['value']} audio-s/s")
            except Exception as ex:      # noqa: BLE001 -- a failed side leg must not lose the headline line
                out["omnilingual"][variant] = {"error": str(ex)}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, clips[0], n_dec)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

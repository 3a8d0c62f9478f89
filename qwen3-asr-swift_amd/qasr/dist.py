"""Utterance-batch data parallelism: independent clips are sharded over ranks (one process per GPU),
weights are replicated, and the only exchange is one all_gather of the fixed-shape int32 token block
[B_local, max_new_tokens + 1] (+ lengths) -- SURVEY.md section 8(e).  The reference has no counterpart
(it is single-process, B = 1).  Works with any torch.distributed backend (nccl == RCCL on ROCm; gloo in
the CPU tests)."""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_clips, world, rank):
    """Contiguous block partition; the first (n_clips % world) ranks take one extra clip."""
    base, extra = divmod(n_clips, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_tokens(tokens, lens, n_clips, device=None):
    """tokens [B_local, S] int32 (numpy), lens [B_local] -> (all tokens [n_clips, S], all lens) on every rank.

    Ranks may hold different B_local (ragged tail): blocks are padded to the largest shard so that the
    collective has a fixed shape, then trimmed."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    S = tokens.shape[1]
    if world == 1:
        return tokens[:n_clips].copy(), lens[:n_clips].copy()
    per = max(shard_bounds(n_clips, world, r)[1] - shard_bounds(n_clips, world, r)[0] for r in range(world))
    block = torch.full((per, S + 1), -1, dtype=torch.int32)
    b = tokens.shape[0]
    block[:b, :S] = torch.from_numpy(np.ascontiguousarray(tokens))
    block[:b, S] = torch.from_numpy(np.ascontiguousarray(lens))
    if device is not None:
        block = block.to(device)
    out = torch.empty((world * per, S + 1), dtype=torch.int32, device=block.device)
    dist.all_gather_into_tensor(out, block)
    out = out.cpu().numpy().reshape(world, per, S + 1)
    rows = []
    for r in range(world):
        lo, hi = shard_bounds(n_clips, world, r)
        rows.append(out[r, :hi - lo])
    allr = np.concatenate(rows, axis=0)
    return allr[:, :S].copy(), allr[:, S].copy()


def transcribe_sharded(model, clips, device=None, max_len=None, **opt):
    """Every rank passes the full clip list; returns the full token lists on every rank.  max_len: longest id list a clip can
    produce (default: the Qwen3 engine's max_new_tokens + 1; the wav2vec2-CTC engine passes its frame capacity)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(len(clips), world, rank)
    S = int(max_len) if max_len is not None else model.cfg.max_new_tokens + 1
    toks = np.full((hi - lo, S), -1, dtype=np.int32)
    lens = np.zeros(hi - lo, dtype=np.int32)
    if hi > lo:
        local = model.transcribe_batch(clips[lo:hi], **opt)
        for i, t in enumerate(local):
            toks[i, :len(t)] = t
            lens[i] = len(t)
    all_t, all_l = gather_tokens(toks, lens, len(clips), device)
    return [all_t[i, :all_l[i]].tolist() for i in range(len(clips))]


def weak_scaling_clip_ids(rank, clips_per_rank):
    """bench.py's weak-scaling assignment: rank r owns synthetic clips r * B .. r * B + B - 1 (no two ranks share a clip)."""
    return list(range(rank * clips_per_rank, (rank + 1) * clips_per_rank))


def _device_sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def timed_passes(model, clips, n_dec, steps, warmup, inclusive, gathered=None, pipelined=False):
    """bench.py's timed region: `warmup` untimed + `steps` timed passes of the hot path over this rank's clips, bracketed
    by barrier + device synchronisation on both sides, MAX over ranks of the elapsed wall time.

    inclusive: every pass starts from the caller's host buffers (qasr_batch_begin: pinned staging + H2D + planning) and
    ends with the token ids on the host -- SURVEY.md section 8(d)'s timed region; otherwise the batch stays resident in HBM
    and qasr_batch_rewind re-arms it (the kernel-only figure).  gathered: [world * B, S] int32 tensor on the collective's
    device; every pass ends with the all_gather of this rank's [B, S] token block into it (the only exchange of the path).
    pipelined (with inclusive): consecutive passes overlap like a serving loop -- while pass i's encoder / prompt pass / decode run, pass
    i + 1's clips are staged from the caller's host buffers into pinned memory and copied to HBM (qasr_batch_stage), and pass i + 1 starts
    with the planning only (qasr_batch_begin_staged).  Every pass still begins at host buffers and ends with token ids on the host; K passes
    do K stagings, K uploads and K full computations.
    -> (seconds, lens of the last pass)."""
    import time
    world = dist.get_world_size() if dist.is_initialized() else 1
    staged = {"ready": False}

    def one_pass():
        if inclusive and pipelined:
            if staged["ready"]:
                model.batch_begin_staged(max_tokens=n_dec, ignore_eos=True)
            else:
                model.batch_begin(clips, max_tokens=n_dec, ignore_eos=True)
            model.batch_run()
            model.batch_stage(clips)                          # the next pass's input, under this pass's kernels
            staged["ready"] = True
            toks, lens = model.batch_tokens()
        else:
            if inclusive:
                model.batch_begin(clips, max_tokens=n_dec, ignore_eos=True)
            else:
                model.batch_rewind()
            model.batch_run()
            toks, lens = model.batch_tokens()                 # D2H, waits for the engine stream
        if gathered is not None:
            block = torch.from_numpy(toks).to(gathered.device, non_blocking=True)
            if world > 1:
                dist.all_gather_into_tensor(gathered, block)  # RCCL over xGMI (gloo in the CPU tests): [B, S] int32 per rank
            else:
                gathered.copy_(block)
        return lens

    if not inclusive:
        model.batch_begin(clips, max_tokens=n_dec, ignore_eos=True)
        model.batch_sync()
    for _ in range(warmup):
        one_pass()
    _device_sync()
    if world > 1:
        dist.barrier()
    _device_sync()
    t0 = time.perf_counter()
    lens = None
    for _ in range(steps):
        lens = one_pass()
    _device_sync()
    if world > 1:
        dist.barrier()
    _device_sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=gathered.device if gathered is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, lens

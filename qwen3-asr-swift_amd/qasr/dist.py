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


def transcribe_sharded(model, clips, device=None, **opt):
    """Every rank passes the full clip list; returns the full token lists on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(len(clips), world, rank)
    S = model.cfg.max_new_tokens + 1
    toks = np.full((hi - lo, S), -1, dtype=np.int32)
    lens = np.zeros(hi - lo, dtype=np.int32)
    if hi > lo:
        local = model.transcribe_batch(clips[lo:hi], **opt)
        for i, t in enumerate(local):
            toks[i, :len(t)] = t
            lens[i] = len(t)
    all_t, all_l = gather_tokens(toks, lens, len(clips), device)
    return [all_t[i, :all_l[i]].tolist() for i in range(len(clips))]

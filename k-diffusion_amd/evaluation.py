"""Batch/gather driver of the sampling path (k_diffusion/evaluation.py:80-90).

Only ``compute_features`` is on the hot path; the FID/KID feature extractors (CLIP / Inception /
DINOv2, need network + pretrained nets) are out of scope.
"""
import math

import torch
from tqdm.auto import trange


def compute_features(accelerator, sample_fn, extractor_fn, n, batch_size):
    """Every rank draws ceil(n / world) samples in batches, each batch is all-gathered (RCCL over xGMI
    when ``accelerator`` is a multi-GPU ``RankContext`` / ``Accelerator``), result cut to ``n``.
    Keeps the reference's batch-size rule ``min(n - i, batch_size)`` (global ``n``, :85)."""
    per_rank = math.ceil(n / accelerator.num_processes)
    gathered = []
    try:
        for i in trange(0, per_rank, batch_size, disable=not accelerator.is_main_process):
            cur = min(n - i, batch_size)
            samples = sample_fn(cur)[:cur]
            gathered.append(accelerator.gather(extractor_fn(samples)))
    except StopIteration:
        pass
    return torch.cat(gathered)[:n]


def indexed_rounds(n, world, batch_size):
    """The gather rounds of ``compute_features_indexed``: a list of ``(width, [(lo_r, count_r) for every rank r])`` -- in a round rank r
    draws the ``count_r`` images ``lo_r .. lo_r + count_r - 1`` of its contiguous shard (``distributed.shard_range``), padded to the
    common ``width``.  A pure function of (n, world, batch_size): every rank computes every rank's share on the host, so no index
    vector has to travel with the images."""
    from .distributed import shard_range
    per = math.ceil(n / world)
    spans = [shard_range(n, world, r) for r in range(world)]
    rounds = []
    for start in range(0, per, batch_size):
        width = min(batch_size, per - start)                       # the same on every rank
        rounds.append((width, [(lo + start, max(0, min(width, hi - lo - start))) for lo, hi in spans]))
    return rounds


def compute_features_indexed(accelerator, sample_fn, n, batch_size, post=None, on_schedule=None):
    """``n`` samples addressed by GLOBAL index, independent of the process count: rank r owns the contiguous shard
    ``distributed.shard_range(n, world, r)`` and draws it in batches of at most ``batch_size``; ``sample_fn(indices)`` (a 1-D
    int64 CPU tensor, possibly empty) returns the samples of exactly those indices; every round is all-gathered (RCCL over xGMI)
    and copied into the result at the positions the HOST knows (``indexed_rounds``), so ``out[i]`` IS sample ``i`` whatever the
    batch size and the number of GPUs -- and no round waits for the device (the reference's loop, evaluation.py:84-88, does not
    either; a mask-indexed scatter would put a device->host sync behind every batch and leave the GPU idle while the next batch
    is prepared).  (``compute_features`` above keeps the reference's schedule, which sizes a rank's batches by the GLOBAL
    remainder -- evaluation.py:85 -- and returns the ranks' batches interleaved: fine for FID features, wrong for "image i of a
    seeded run".)  ``post``: applied to a rank's batch before the gather (e.g. ``ops.to_uint8``: 4x fewer bytes over xGMI).
    ``on_schedule``: called once, before the first round, with this rank's list of index tensors (one per round, in order) -- a
    sample_fn that prepares inputs ahead of time (``synth.NoisePrefetcher``) learns the whole job from it."""
    world, rank = accelerator.num_processes, accelerator.process_index
    rounds = indexed_rounds(n, world, batch_size)
    mine = [torch.arange(shares[rank][0], shares[rank][0] + shares[rank][1]) for _, shares in rounds]
    if on_schedule is not None:
        on_schedule(mine)
    out = None
    for k in trange(len(rounds), disable=not accelerator.is_main_process):
        width, shares = rounds[k]
        x = sample_fn(mine[k])
        x = x if post is None else post(x)
        if world > 1 and len(mine[k]) != width:                   # a short or empty shard: pad to the common width for the gather
            buf = x.new_zeros((width, *x.shape[1:]))
            buf[:len(mine[k])] = x
            x = buf
        g_x = accelerator.gather(x)
        if out is None:
            out = g_x.new_zeros((n, *g_x.shape[1:]))
        for r, (lo, count) in enumerate(shares):
            if count:
                out[lo:lo + count] = g_x[r * width:r * width + count]
    if out is None:                  # n == 0: no round ran and nothing tells the sample shape (the reference's torch.cat of no batches raises here)
        out = torch.empty(0, device=accelerator.device)
    return out

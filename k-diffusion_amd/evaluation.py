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


def compute_features_indexed(accelerator, sample_fn, n, batch_size, post=None):
    """``n`` samples addressed by GLOBAL index, independent of the process count: rank r owns the contiguous shard
    ``distributed.shard_range(n, world, r)`` and draws it in batches of at most ``batch_size``; ``sample_fn(indices)`` (a 1-D
    int64 CPU tensor, possibly empty) returns the samples of exactly those indices; every round is all-gathered together
    with its index vector (RCCL over xGMI) and scattered into the result, so ``out[i]`` IS sample ``i`` whatever the batch size
    and the number of GPUs.  (``compute_features`` above keeps the reference's schedule, which sizes a rank's batches by the
    GLOBAL remainder -- evaluation.py:85 -- and returns the ranks' batches interleaved: fine for FID features, wrong for
    "image i of a seeded run".)  ``post``: applied to a rank's batch before the gather (e.g. ``ops.to_uint8``: 4x fewer bytes
    over xGMI)."""
    from .distributed import shard_range
    world, rank = accelerator.num_processes, accelerator.process_index
    lo, hi = shard_range(n, world, rank)
    per = math.ceil(n / world)
    out = None
    for start in trange(0, per, batch_size, disable=not accelerator.is_main_process):
        width = min(batch_size, per - start)                       # the same on every rank
        idx = torch.arange(lo + start, lo + start + width)
        real = idx[idx < hi]
        x = sample_fn(real)
        x = x if post is None else post(x)
        buf = x.new_zeros((width, *x.shape[1:]))
        buf[:len(real)] = x
        tag = torch.full((width,), -1, dtype=torch.int64)
        tag[:len(real)] = real
        g_x, g_tag = accelerator.gather(buf), accelerator.gather(tag.to(buf.device))
        if out is None:
            out = g_x.new_zeros((n, *g_x.shape[1:]))
        keep = g_tag >= 0
        out[g_tag[keep]] = g_x[keep]
    return out

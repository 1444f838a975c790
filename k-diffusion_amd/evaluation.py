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

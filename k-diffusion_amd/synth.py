"""Deterministic synthetic ("random-init") weights for ImageTransformerDenoiserModelV2.

The reference zero-initialises every residual-branch output projection, every AdaRMSNorm
projection and the final un-patch projection (image_transformer_v2.py:159, 365, 410, 458, 485,
558, 706), so a freshly constructed network outputs exactly zero and says nothing about the
attention / feed-forward kernels.  For benchmarking and parity work there is no network access
to real checkpoints either, so this module defines ONE recipe -- a function of the parameter
*name*, *shape* and a seed only -- that both the HIP model, the CPU oracle and the golden-vector
generator (which loads the result into the real reference model) use.

Every tensor is drawn on the CPU from its own ``torch.Generator`` seeded by
``crc32(name) ^ seed``, so the values do not depend on parameter order, device or world size.
"""
import math
import zlib

import torch

_RESIDUAL_OUT = ("out_proj.weight", "down_proj.weight")


def _gen(name, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFFFFFFFFFF)
    return g


def synth_tensor(name, shape, seed=0, template=None):
    """One synthetic fp32 tensor for parameter/buffer ``name``."""
    g = _gen(name, seed)
    rn = lambda: torch.randn(shape, generator=g, dtype=torch.float32)
    if name.endswith("pos_emb.freqs"):
        # deterministic buffer (image_transformer_v2.py:237-240): keep the model's own values
        if template is None:
            raise ValueError("pos_emb.freqs needs the constructed buffer as template")
        return template.detach().to(torch.float32).clone()
    if name in ("time_emb.weight", "aug_emb.weight"):       # FourierFeatures buffers, std 1
        return rn()
    if name == "class_emb.weight":                           # nn.Embedding default N(0, 1)
        return rn()
    if name.endswith("self_attn.scale"):                     # per-head cosine-sim scale, > 0
        return 10.0 * torch.exp(0.2 * rn())
    if name.endswith(".scale"):                              # RMSNorm gains
        return 1.0 + 0.1 * rn()
    if name.endswith(".fac"):                                # TokenSplit lerp factor
        return 0.5 + 0.1 * rn()
    if name.endswith("norm.linear.weight"):                  # AdaRMSNorm cond projection (zero-init in ref)
        return 0.03 * rn()
    if len(shape) == 2:
        fan_in = shape[1]
        gain = 0.5 if name.endswith(_RESIDUAL_OUT) else 1.0
        return rn() * (gain / math.sqrt(fan_in))
    raise ValueError(f"no synthetic-weight rule for {name} {tuple(shape)}")


def synth_state_dict(template_state_dict, seed=0):
    """Synthetic state dict with the same keys / shapes as ``template_state_dict``."""
    return {k: synth_tensor(k, tuple(v.shape), seed, template=v) for k, v in template_state_dict.items()}


def synth_noise(shape_chw, seed, global_index, sigma_max):
    """Initial noise for the sample with global index ``global_index`` (independent of the GPU
    count, unlike sample.py:59's rank-local torch.randn): randn(C,H,W) * sigma_max on the CPU."""
    g = torch.Generator(device="cpu")
    g.manual_seed(((seed << 32) + global_index) & 0x7FFFFFFFFFFFFFFF)
    return torch.randn(shape_chw, generator=g, dtype=torch.float32) * sigma_max


def synth_noise_batch(shape_chw, seed, first_index, count, sigma_max):
    """[count, C, H, W]: ``synth_noise`` of the global indices first_index .. first_index + count - 1 (one generator per image, so
    a sample's noise does not depend on the batch / rank it is drawn in), written into one pinned-size buffer."""
    out = torch.empty((count, *shape_chw), dtype=torch.float32)
    g = torch.Generator(device="cpu")
    for i in range(count):
        g.manual_seed(((seed << 32) + first_index + i) & 0x7FFFFFFFFFFFFFFF)
        torch.randn(shape_chw, generator=g, dtype=torch.float32, out=out[i])
    return out.mul_(sigma_max)


class NoisePrefetcher:
    """Host-drawn start noise of a seeded job, drawn AHEAD of the sampler: ``synth_noise`` is one CPU generator per image (85 - 105 ms
    for a 32 x 3 x 256 x 256 batch on one thread), which a job that draws batch n + 1 only after batch n has finished pays in full
    beside every pass.  Here the batches of the job's schedule (``plan``: a list of 1-D int64 CPU index tensors) are drawn by worker
    threads (``torch.randn`` releases the GIL; one task per image, so a batch takes draw time / threads) into a ring of pinned
    buffers while the GPU runs the batches before them, and ``take`` hands batch k over as an asynchronous host-to-device copy on
    the current stream.  The values are exactly ``synth_noise(shape, seed, g, sigma_max)`` for every global index g."""

    def __init__(self, shape_chw, seed, sigma_max, device, depth=2, threads=None, width=0):
        import os
        from concurrent.futures import ThreadPoolExecutor
        self.shape, self.seed, self.sigma_max, self.device = tuple(shape_chw), int(seed), float(sigma_max), torch.device(device)
        self.depth = max(1, int(depth))
        self.pool = ThreadPoolExecutor(max_workers=threads or int(os.environ.get('KDIFF_NOISE_THREADS', '0')) or max(1, min(16, (os.cpu_count() or 2) - 1)), thread_name_prefix='kd-noise')
        self.plan, self.pending, self.next_to_fill, self.slots, self.copy_stream = [], {}, 0, [], None
        if width:                         # the pinned ring now (set-up, like the model's construction) instead of at schedule()
            self._ring(int(width))

    def _ring(self, width):
        if not self.slots or self.slots[0]['buf'].shape[0] < width:
            pin = self.device.type == 'cuda'
            self.slots = [{'buf': torch.empty((width, *self.shape), dtype=torch.float32, pin_memory=pin), 'copied': None} for _ in range(self.depth + 1)]

    def schedule(self, plan):
        """The job's batches in the order they will be taken; starts drawing the first ``depth`` of them."""
        self.plan, self.pending, self.next_to_fill = [torch.as_tensor(p, dtype=torch.int64) for p in plan], {}, 0
        self._ring(max([len(p) for p in self.plan] + [1]))
        for _ in range(self.depth):
            self._fill_next()

    def _draw(self, slot, row, g):
        ev = slot['copied']
        if ev is not None:
            ev.synchronize()              # the copy that last read this slot (depth + 1 batches back) has left the host buffer
        gen = torch.Generator(device='cpu')
        gen.manual_seed(((self.seed << 32) + g) & 0x7FFFFFFFFFFFFFFF)
        # (the scaling by sigma_max happens after the copy, on the device -- the same IEEE multiply; on the host it is an intra-op
        # parallel loop, and one OpenMP team per worker thread oversubscribes the cores: 40 ms per batch instead of 8)
        torch.randn(self.shape, generator=gen, dtype=torch.float32, out=slot['buf'][row])

    def _fill_next(self):
        k = self.next_to_fill
        if k >= len(self.plan):
            return
        self.next_to_fill += 1
        slot = self.slots[k % len(self.slots)]
        self.pending[k] = [self.pool.submit(self._draw, slot, row, int(g)) for row, g in enumerate(self.plan[k])]

    def take(self, k):
        """[len(plan[k]), C, H, W] on the device (asynchronous copy from the pinned slot), and the next batch's draw is started."""
        if k not in self.pending:
            raise KeyError(f'batch {k} is not scheduled (take the batches in schedule order)')
        for f in self.pending.pop(k):
            f.result()
        slot = self.slots[k % len(self.slots)]
        n = len(self.plan[k])
        if self.device.type == 'cuda':
            # on a side stream: the host is a batch ahead of the GPU, so this copy (25 MB at the headline shape, ~1 ms) runs beside the
            # previous batch's kernels instead of between two batches; the sampler's stream waits for its event
            if self.copy_stream is None:
                self.copy_stream = torch.cuda.Stream(device=self.device)
            cur = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self.copy_stream):
                x = slot['buf'][:n].to(self.device, non_blocking=True)
                x.mul_(self.sigma_max)
                slot['copied'] = torch.cuda.Event()
                slot['copied'].record(self.copy_stream)
            cur.wait_event(slot['copied'])
            x.record_stream(cur)
        else:
            x = slot['buf'][:n] * self.sigma_max
        self._fill_next()
        return x

    def close(self):
        self.pool.shutdown(wait=True)

"""One process per GPU over torch.distributed (backend "nccl" == RCCL on ROCm, xGMI inside a node).

``RankContext`` offers the slice of ``accelerate.Accelerator`` that the reference's sampling path uses
(sample.py:39-66, evaluation.py:80-90): ``device``, ``num_processes``, ``process_index``,
``is_main_process``, ``is_local_main_process``, ``gather``, ``print``, ``wait_for_everyone`` -- so
``K.evaluation.compute_features`` accepts either.  The sampling path needs exactly one collective: the
all-gather of finished images (evaluation.py:87); everything before it is embarrassingly parallel over
images.  Works on CPU with the gloo backend (tests) and needs nothing when WORLD_SIZE == 1.
"""
import os

import torch
import torch.distributed as dist


class RankContext:
    def __init__(self, device=None, backend=None, force_collectives=False):
        """``force_collectives``: create the process group and issue the collectives even when WORLD_SIZE == 1 (a one-rank RCCL
        communicator: how the 1-GPU test box exercises the very calls the N-GPU job makes)."""
        self.num_processes = int(os.environ.get("WORLD_SIZE", "1"))
        self.process_index = int(os.environ.get("RANK", "0"))
        self.local_process_index = int(os.environ.get("LOCAL_RANK", "0"))
        # RCCL between processes goes over dmabuf IPC (the host driver has no legacy IPC); the runtime reads this when HIP initialises, i.e.
        # at the first torch.cuda call below, not when the process group is made
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if device is None:
            if torch.cuda.is_available():
                torch.cuda.set_device(self.local_process_index % torch.cuda.device_count())
                device = torch.device("cuda", torch.cuda.current_device())
            else:
                device = torch.device("cpu")
        self.device = torch.device(device)
        self._owns_group = False
        self.collectives = self.num_processes > 1 or force_collectives
        if self.collectives and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            backend = backend or ("nccl" if self.device.type == "cuda" else "gloo")
            kw = {"device_id": self.device} if backend == "nccl" else {}
            dist.init_process_group(backend, rank=self.process_index, world_size=self.num_processes, **kw)
            self._owns_group = True

    @property
    def is_main_process(self):
        return self.process_index == 0

    @property
    def is_local_main_process(self):
        return self.local_process_index == 0

    def gather(self, tensor):
        """All-gather along dim 0 (every rank must pass the same shape), like Accelerator.gather."""
        if not self.collectives:
            return tensor
        tensor = tensor.contiguous()
        out = torch.empty((self.num_processes * tensor.shape[0], *tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
        dist.all_gather_into_tensor(out, tensor)
        return out

    def wait_for_everyone(self):
        if self.collectives:
            dist.barrier()

    def print(self, *args, **kwargs):
        if self.is_main_process:
            print(*args, **kwargs)

    def shutdown(self):
        if self._owns_group and dist.is_initialized():
            dist.destroy_process_group()
            self._owns_group = False


def shard_range(n, world_size, rank):
    """Global sample indices [lo, hi) owned by ``rank`` when ``n`` samples are dealt out in
    ceil(n / world)-sized contiguous shards (evaluation.py:81's n_per_proc)."""
    per = -(-n // world_size)
    return min(rank * per, n), min((rank + 1) * per, n)

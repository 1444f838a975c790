"""Small helpers shared by the sampling path (k_diffusion/utils.py:19-85, :446-458 equivalents)."""
import json
import struct
from contextlib import contextmanager

import torch


def append_dims(x, target_dims):
    """Trailing singleton dims until ``x`` has ``target_dims`` dims (utils.py:43-48)."""
    extra = target_dims - x.ndim
    if extra < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x.reshape(x.shape + (1,) * extra)


def n_params(module):
    return sum(p.numel() for p in module.parameters())


def to_pil_image(x):
    """[-1, 1] tensor -> PIL image (utils.py:27-34).  The clamp / rescale / 8-bit conversion runs in the
    HIP ``kd_to_uint8`` kernel when ``x`` lives on the GPU (truncating like torchvision's to_pil_image)."""
    from PIL import Image
    if x.ndim == 4:
        assert x.shape[0] == 1
        x = x[0]
    if x.dtype == torch.uint8:                          # already converted on the device (sample.py --gather-uint8)
        u8 = x.cpu()
    elif x.is_cuda:
        from . import ops
        u8 = ops.to_uint8(x.to(torch.float32).contiguous()).cpu()
    else:
        u8 = (((x.float().clamp(-1, 1) + 1) / 2) * 255).to(torch.uint8)
    arr = u8.numpy()
    if arr.shape[0] == 1:
        return Image.fromarray(arr[0], mode="L")
    return Image.fromarray(arr.transpose(1, 2, 0), mode="RGB" if arr.shape[0] == 3 else None)


@contextmanager
def _mode(model, training):
    previous = [m.training for m in model.modules()]
    try:
        yield model.train(training)
    finally:
        for m, was in zip(model.modules(), previous):
            m.training = was


def train_mode(model, mode=True):
    """Context manager / decorator that puts ``model`` in train (or eval) mode and restores it."""
    return _mode(model, mode)


def eval_mode(model):
    return _mode(model, False)


def get_safetensors_metadata(path):
    """The ``__metadata__`` dict of a safetensors file, read from its JSON header only."""
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        return json.loads(f.read(n)).get("__metadata__", {})

"""MI355X-native k-diffusion sampling hot path.

``import k_diffusion_amd as K`` mirrors ``import k_diffusion as K`` for the sampling path:
K.sampling, K.layers / K.Denoiser, K.config, K.models, K.evaluation, K.utils (+ K.distributed,
K.ops, K.synth, and K.compat: natten's na2d / flash-attn's packed call / SDPA under their own signatures on the HIP cores).  Importing the package never needs a GPU; the first kernel call loads
csrc/libkdiff_hip.so and fails loudly if it is missing (there is no CPU fallback).
"""
from . import _native, checkpoint, compat, config, distributed, evaluation, external, layers, models, ops, sampling, synth, utils  # noqa: F401
from .layers import Denoiser  # noqa: F401

__version__ = "0.1.0"

"""Karras et al. preconditioned denoiser wrapper (k_diffusion/layers.py:45-111, sampling part).

``Denoiser(inner_model, sigma_data)`` keeps the reference's constructor, attributes
(``inner_model``, ``sigma_data``, ``weighting``, ``scales``) and ``forward(input, sigma, **kwargs)``.
With the native HDiT inner model the three preconditioning passes (x * c_in, F * c_out, + x * c_skip)
disappear into the patch-in / patch-out GEMMs; with a foreign inner model they run as two HIP
elementwise kernels around it.  The training-time ``loss`` methods are out of scope.
"""
import torch
from torch import nn

from . import ops


class Denoiser(nn.Module):
    """D(x, sigma) = F(x * c_in, sigma) * c_out + x * c_skip."""

    def __init__(self, inner_model, sigma_data=1., weighting='karras', scales=1):
        super().__init__()
        self.inner_model = inner_model
        self.sigma_data = sigma_data
        self.scales = scales
        if not callable(weighting) and weighting not in ('karras', 'soft-min-snr', 'snr'):
            raise ValueError(f'Unknown weighting type {weighting}')
        self.weighting = weighting

    def get_scalings(self, sigma):
        """(c_skip, c_out, c_in) for a sigma tensor (layers.py:70-74)."""
        var = sigma ** 2 + self.sigma_data ** 2
        return self.sigma_data ** 2 / var, sigma * self.sigma_data / var ** 0.5, 1 / var ** 0.5

    def loss(self, *args, **kwargs):
        raise NotImplementedError('training losses are outside this package\'s scope (sampling hot path only)')

    def prefetch_schedule(self, x_like, sigma_table, **kwargs):
        """Solver-loop hint (see ImageTransformerDenoiserModelV2.prefetch_schedule); a no-op for foreign inner models."""
        hint = getattr(self.inner_model, 'prefetch_schedule', None)
        return hint(x_like, sigma_table, **kwargs) if hint is not None else False

    def prefetch_conditioning(self, x_like, sigma, **kwargs):
        """Solver-loop hint (see ImageTransformerDenoiserModelV2.prefetch_conditioning); a no-op for foreign inner models."""
        hint = getattr(self.inner_model, 'prefetch_conditioning', None)
        if hint is not None:
            hint(x_like, sigma, **kwargs)

    def forward(self, input, sigma, **kwargs):
        inner = self.inner_model
        fused = getattr(inner, 'forward_preconditioned', None)
        if fused is not None:
            return fused(input, sigma, self.sigma_data, **kwargs)
        sigma = sigma.to(device=input.device, dtype=torch.float32).reshape(-1).expand(input.shape[0]).contiguous()
        x = input.contiguous()
        f = inner(ops.precond_in(x, sigma, self.sigma_data), sigma, **kwargs)
        return ops.precond_out(f.contiguous(), x, sigma, self.sigma_data)


class DenoiserWithVariance(Denoiser):
    """What ``make_denoiser_wrapper`` returns for ``has_variance`` configs (config.py:223-224; layers.py:93-101).  The reference's class
    overrides ``loss`` only (the model's log-variance output is a training quantity): ``forward`` / ``get_scalings`` -- the sampling path --
    are ``Denoiser``'s, as here."""


class SimpleLossDenoiser(Denoiser):
    """``loss_config == 'simple'`` (config.py:229-230; layers.py:104-111): again only ``loss`` differs in the reference."""


class FourierFeatures(nn.Module):
    """``cat[cos, sin](2 pi x W^T)`` with ``W`` a ``randn([out_features // 2, in_features]) * std`` BUFFER (k_diffusion/layers.py:285-293; the
    model's ``time_emb`` and ``aug_emb``, image_transformer_v2.py:677-680, whose state_dict entries are this buffer).  The model itself reaches
    the kernel directly (``ops.fourier_sigma`` folds ``c_noise = log(sigma) / 4`` in); this is the stand-alone module of the reference's
    surface, on the same kernel (``kd_fourier_f32``: angles in revolutions, hardware sin / cos).  ROCm tensors only, like every op here."""

    def __init__(self, in_features, out_features, std=1.):
        super().__init__()
        assert out_features % 2 == 0
        self.register_buffer('weight', torch.randn([out_features // 2, in_features]) * std)

    def forward(self, input):
        x = input.to(torch.float32)
        if x.shape[-1] != self.weight.shape[1]:
            raise ValueError(f'FourierFeatures: last dimension {x.shape[-1]} != in_features {self.weight.shape[1]}')
        return ops.fourier_features(x.contiguous(), self.weight.to(torch.float32).contiguous())

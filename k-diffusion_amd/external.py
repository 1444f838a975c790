"""Wrappers that let foreign (v-prediction / discrete-schedule DDPM) models drive the Karras samplers
(k_diffusion/external.py).  Same class names, constructor arguments, attributes and ``forward(input, sigma, **kwargs)``
contract as the reference; the image-sized arithmetic and the sigma <-> t maps run as HIP kernels
(``kd_rows_affine_f32``, ``kd_sigma_to_t_f32``, ``kd_t_to_sigma_f32``), the per-sample scalar algebra ([B]-sized) stays in
torch on the device.  Training losses are outside this package's scope (sampling hot path only).
"""
import math

import torch
from torch import nn

from . import ops, sampling


def _rows(v, batch):
    return v.to(torch.float32).reshape(-1).expand(batch).contiguous()


class _AffineWrapper(nn.Module):
    """D(x, sigma) = inner(x * c_in, cond_of(sigma)) * c_out + x * c_skip, per-sample scalars from get_scalings()."""

    def _scalars(self, sigma):
        raise NotImplementedError

    def _inner(self, x, t, **kwargs):
        return self.inner_model(x, t, **kwargs)

    def loss(self, *args, **kwargs):
        raise NotImplementedError('training losses are outside this package\'s scope (sampling hot path only)')

    def forward(self, input, sigma, **kwargs):
        x = input.contiguous()
        B = x.shape[0]
        c_skip, c_out, c_in = self._scalars(sigma)
        out = self._inner(ops.rows_affine(x, _rows(c_in, B)), self.sigma_to_t(sigma), **kwargs)
        if c_skip is None:            # eps models: x + eps * c_out
            return ops.rows_affine(out.contiguous(), _rows(c_out, B), x, torch.ones(B, device=x.device))
        return ops.rows_affine(out.contiguous(), _rows(c_out, B), x, _rows(c_skip, B))


class VDenoiser(_AffineWrapper):
    """A v-diffusion-pytorch model wrapper (external.py:9-38)."""

    def __init__(self, inner_model):
        super().__init__()
        self.inner_model = inner_model
        self.sigma_data = 1.

    def get_scalings(self, sigma):
        var = sigma ** 2 + self.sigma_data ** 2
        return self.sigma_data ** 2 / var, -sigma * self.sigma_data / var ** 0.5, 1 / var ** 0.5

    _scalars = get_scalings

    def sigma_to_t(self, sigma):
        return sigma.atan() / math.pi * 2

    def t_to_sigma(self, t):
        return (t * math.pi / 2).tan()


class DiscreteSchedule(nn.Module):
    """A mapping between continuous noise levels (sigmas) and a list of discrete noise levels (external.py:41-84)."""

    def __init__(self, sigmas, quantize):
        super().__init__()
        self.register_buffer('sigmas', sigmas.to(torch.float32))
        self.register_buffer('log_sigmas', sigmas.to(torch.float32).log())
        self.quantize = quantize

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def get_sigmas(self, n=None):
        if n is None:
            return sampling.append_zero(self.sigmas.flip(0))
        t = torch.linspace(len(self.sigmas) - 1, 0, n, device=self.sigmas.device)
        return sampling.append_zero(self.t_to_sigma(t))

    def sigma_to_t(self, sigma, quantize=None):
        quantize = self.quantize if quantize is None else quantize
        t = ops.sigma_to_t(sigma.to(torch.float32), self.log_sigmas, quantize).view(sigma.shape)
        return t.long() if quantize else t

    def t_to_sigma(self, t):
        return ops.t_to_sigma(t.float(), self.log_sigmas)


class DiscreteEpsDDPMDenoiser(DiscreteSchedule, _AffineWrapper):
    """Discrete-schedule DDPM models that output eps (external.py:87-113)."""

    def __init__(self, model, alphas_cumprod, quantize):
        super().__init__(((1 - alphas_cumprod) / alphas_cumprod) ** 0.5, quantize)
        self.inner_model = model
        self.sigma_data = 1.

    def get_scalings(self, sigma):
        return -sigma, 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5

    def _scalars(self, sigma):
        c_out, c_in = self.get_scalings(sigma)
        return None, c_out, c_in

    def get_eps(self, *args, **kwargs):
        return self.inner_model(*args, **kwargs)

    def _inner(self, x, t, **kwargs):
        return self.get_eps(x, t, **kwargs)


class OpenAIDenoiser(DiscreteEpsDDPMDenoiser):
    """A wrapper for OpenAI diffusion models (external.py:116-128)."""

    def __init__(self, model, diffusion, quantize=False, has_learned_sigmas=True, device='cpu'):
        super().__init__(model, torch.tensor(diffusion.alphas_cumprod, device=device, dtype=torch.float32), quantize=quantize)
        self.has_learned_sigmas = has_learned_sigmas

    def get_eps(self, *args, **kwargs):
        out = self.inner_model(*args, **kwargs)
        return out.chunk(2, dim=1)[0] if self.has_learned_sigmas else out


class CompVisDenoiser(DiscreteEpsDDPMDenoiser):
    """A wrapper for CompVis diffusion models (external.py:131-138)."""

    def __init__(self, model, quantize=False, device='cpu'):
        super().__init__(model, model.alphas_cumprod, quantize=quantize)

    def get_eps(self, *args, **kwargs):
        return self.inner_model.apply_model(*args, **kwargs)


class DiscreteVDDPMDenoiser(DiscreteSchedule, _AffineWrapper):
    """Discrete-schedule DDPM models that output v (external.py:141-167)."""

    def __init__(self, model, alphas_cumprod, quantize):
        super().__init__(((1 - alphas_cumprod) / alphas_cumprod) ** 0.5, quantize)
        self.inner_model = model
        self.sigma_data = 1.

    def get_scalings(self, sigma):
        var = sigma ** 2 + self.sigma_data ** 2
        return self.sigma_data ** 2 / var, -sigma * self.sigma_data / var ** 0.5, 1 / var ** 0.5

    _scalars = get_scalings

    def get_v(self, *args, **kwargs):
        return self.inner_model(*args, **kwargs)

    def _inner(self, x, t, **kwargs):
        return self.get_v(x, t, **kwargs)


class CompVisVDenoiser(DiscreteVDDPMDenoiser):
    """A wrapper for CompVis diffusion models that output v (external.py:170-177)."""

    def __init__(self, model, quantize=False, device='cpu'):
        super().__init__(model, model.alphas_cumprod, quantize=quantize)

    def get_v(self, x, t, cond, **kwargs):
        return self.inner_model.apply_model(x, t, cond)

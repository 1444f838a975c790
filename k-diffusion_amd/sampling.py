"""Karras ODE/SDE samplers, MI355X-native.

Same call surface as k_diffusion/sampling.py: ``sample_X(model, x, sigmas, extra_args=None,
callback=None, disable=None, **solver_kwargs) -> Tensor`` with ``model(x, sigma[B], **extra_args)
-> denoised``, ``callback({'x','i','sigma','sigma_hat','denoised'})`` once per step, pluggable
``noise_sampler(sigma, sigma_next)``, functional in ``x``.

How the loop runs here (vs. the reference's ~12 scalar + ~6 tensor ATen launches and one
device->host sync per step, sampling.py:594-606):
  * the schedule is read back to the host ONCE; every per-step coefficient is evaluated there with
    0-dim fp32 torch ops in the reference's own operation order (bit-identical scalars, SURVEY.md
    App. A.7), so there is no device-side scalar math and no per-step sync;
  * each step's tensor arithmetic is ONE fused HIP launch (``kd_sampler_step_f32``) whose expression
    tree reproduces the reference's rounding sequence exactly (bit-exact given the same denoised);
  * the per-step sigma vectors handed to the model are rows of one table uploaded before the loop.
"""
import math
import os

import torch
from tqdm.auto import tqdm, trange

from . import _native as nat
from . import ops, utils

_COND_PREFETCH = os.environ.get('KDIFF_COND_PREFETCH', '1') != '0'      # A/B switch (benchmarks/): side-stream conditioning
_COND_SCHEDULE = os.environ.get('KDIFF_COND_SCHEDULE', '1') != '0'      # A/B switch: conditioning of the whole sigma table at once
# KDIFF_STRICT_RNG=1: euler / heun / dpm_2 draw ``randn_like(x)`` on every step even with s_churn == 0, exactly like the reference
# (sampling.py:124,165,195), so that code reading the global generator AFTER a run sees the reference's stream position.  Off by
# default: the draw is unused there and costs one launch + 25 MB of writes per step (documented in INTEGRATION.md).
_STRICT_RNG = os.environ.get('KDIFF_STRICT_RNG', '0') == '1'

# --------------------------------------------------------------------------------- schedules


def append_zero(x):
    return torch.cat([x, x.new_zeros([1])])


def _to_device_with_host_copy(sig, device):
    """``sig.to(device)`` that remembers the CPU tensor it came from (``_kd_host``: the copy and the device tensor's version at
    that moment): a sampler loop needs the schedule's VALUES on the host, and reading them back from the device is a blocking
    device->host transfer -- in a multi-batch job it would hold the host until the previous batch's GPU pass has drained."""
    dev = sig.to(device)
    if dev is not sig:
        if not dev.is_inference():        # (inference-mode tensors carry no version counter: no shortcut for them)
            dev._kd_host = (sig, dev._version)
    return dev


def _host_values(sigmas):
    """fp32 CPU copy of a schedule: the remembered one when the tensor still is what ``get_sigmas_*`` returned (same version: no
    in-place edit since), else one device->host transfer.  LIMIT: the version counter is what tells an edit.  Writes that bypass it --
    ``sigmas.data[i] = v``, a custom kernel or any write through the raw pointer -- are not seen, and the sampler would step with the
    schedule as ``get_sigmas_*`` returned it; edit schedules with ordinary tensor ops (or pass a fresh tensor)."""
    kept = getattr(sigmas, '_kd_host', None)
    if kept is not None and not sigmas.is_inference() and kept[1] == sigmas._version and kept[0].shape == sigmas.shape \
            and kept[0].dtype == torch.float32:
        return kept[0]
    return sigmas.detach().to('cpu', torch.float32)


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7., device='cpu'):
    """Karras et al. (2022) schedule.  Evaluated on the CPU (fp32 ramp, double scalars) exactly as the
    reference does (sampling.py:17-23) and then moved, so the schedule is bit-identical everywhere."""
    ramp = torch.linspace(0, 1, n)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    return _to_device_with_host_copy(append_zero((hi + ramp * (lo - hi)) ** rho), device)


def get_sigmas_exponential(n, sigma_min, sigma_max, device='cpu'):
    return _to_device_with_host_copy(append_zero(torch.linspace(math.log(sigma_max), math.log(sigma_min), n).exp()), device)


def get_sigmas_polyexponential(n, sigma_min, sigma_max, rho=1., device='cpu'):
    ramp = torch.linspace(1, 0, n) ** rho
    return _to_device_with_host_copy(append_zero(torch.exp(ramp * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min))), device)


def get_sigmas_vp(n, beta_d=19.9, beta_min=0.1, eps_s=1e-3, device='cpu'):
    t = torch.linspace(1, eps_s, n)
    return _to_device_with_host_copy(append_zero(torch.sqrt(torch.exp(beta_d * t ** 2 / 2 + beta_min * t) - 1)), device)


# --------------------------------------------------------------------------------- helpers

def to_d(x, sigma, denoised):
    """Karras ODE derivative (x - denoised) / sigma."""
    if x.is_cuda and (not torch.is_tensor(sigma) or sigma.numel() == 1):
        return ops.sampler_step(nat.STEP_TO_D, x.contiguous(), denoised.contiguous(), c0=float(sigma))
    return (x - denoised) / utils.append_dims(sigma, x.ndim)


def get_ancestral_step(sigma_from, sigma_to, eta=1.):
    """(sigma_down, sigma_up) of an ancestral step (sampling.py:51-58)."""
    if not eta:
        return sigma_to, 0.
    sigma_up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    return (sigma_to ** 2 - sigma_up ** 2) ** 0.5, sigma_up


def default_noise_sampler(x):
    return lambda sigma, sigma_next: torch.randn_like(x)


def _f(v):
    return float(v)


class BatchedBrownianTree:
    """Brownian increments W(t1) - W(t0) for a tensor shaped like ``x`` (sampling.py:65-89).

    The reference wraps one host-side ``torchsde.BrownianTree`` per seed.  Here every element owns a
    virtual Brownian tree evaluated on demand by the HIP kernel ``kd_brownian_f32`` (Philox4x32-10
    keyed by the seed, counter = (element, tree node)); a list of seeds gives one tree per batch item,
    a single seed one tree over the whole tensor.  The random stream is this package's own
    (torchsde's is generator- and device-dependent and not reproducible here)."""

    def __init__(self, x, t0, t1, seed=None, **kwargs):
        t0, t1, self.sign = self.sort(_f(t0), _f(t1))
        self.t0, self.t1 = t0, t1
        if seed is None:
            seed = torch.randint(0, 2 ** 63 - 1, []).item()
        try:
            seeds = [int(s) for s in seed]
            assert len(seeds) == x.shape[0]
            self.batched = True
        except TypeError:
            seeds, self.batched = [int(seed)], False
        self.shape, self.device = tuple(x.shape), x.device
        self.depth = int(kwargs.get('depth', 36))
        self.seeds = torch.tensor([s & 0x7FFFFFFFFFFFFFFF for s in seeds], dtype=torch.int64, device=x.device)
        # W(t) tensors of the most recent end points (torchsde caches its visited nodes on the host,
        # sampling.py:72-79): each sigma_i bounds 2-4 queries, so most descents are skipped
        self.cache_points = int(kwargs.get('cache_points', 3))
        self._points = []                                        # [t, tensor], most recently used last

    @staticmethod
    def sort(a, b):
        return (a, b, 1) if a < b else (b, a, -1)

    def _clip(self, t):
        # queries are derived from the schedule through exp(log(sigma)) round trips: allow roundoff
        # beyond the end points, reject anything really outside the tree's interval
        slack = 1e-4 * max(abs(self.t0), abs(self.t1), 1e-12)
        if t < self.t0 - slack or t > self.t1 + slack:
            raise ValueError(f'Brownian query {t} outside the tree interval [{self.t0}, {self.t1}]')
        return min(max(t, self.t0), self.t1)

    def increment(self, t0, t1, mult=1.0):
        t0, t1, sign = self.sort(self._clip(_f(t0)), self._clip(_f(t1)))
        out = torch.empty(self.shape, device=self.device, dtype=torch.float32)
        view = out if self.batched else out.view(1, -1)
        if self.cache_points < 2:
            ops.brownian(view, self.seeds, self.t0, self.t1, t0, t1, mult * self.sign * sign, self.depth)
            return out
        w0, have0 = self._point(t0, None)
        w1, have1 = (w0, have0) if t1 == t0 else self._point(t1, w0)
        ops.brownian_cached(view, w0.view(view.shape), have0, w1.view(view.shape), have1, self.seeds, self.t0, self.t1, t0, t1,
                            mult * self.sign * sign, self.depth)
        return out

    def _point(self, t, keep):
        """Buffer for W(t) and whether it is already filled; recycles the least recently used buffer
        (never ``keep``, the other end point of the query being served)."""
        for k, entry in enumerate(self._points):
            if entry[0] == t:
                self._points.append(self._points.pop(k))
                return entry[1], True
        if len(self._points) < self.cache_points:
            buf = torch.empty(self.shape, device=self.device, dtype=torch.float32)
        else:
            k = 0 if self._points[0][1] is not keep else 1
            buf = self._points.pop(k)[1]
        self._points.append([t, buf])
        return buf, False

    def __call__(self, t0, t1):
        return self.increment(t0, t1)


class BrownianTreeNoiseSampler:
    """Unit-variance, path-consistent noise between two sigmas (sampling.py:92-114)."""

    def __init__(self, x, sigma_min, sigma_max, seed=None, transform=lambda x: x):
        self.transform = transform
        t0, t1 = self.transform(torch.as_tensor(_f(sigma_min))), self.transform(torch.as_tensor(_f(sigma_max)))
        self.tree = BatchedBrownianTree(x, t0, t1, seed)

    def __call__(self, sigma, sigma_next):
        """``sigma`` / ``sigma_next``: python floats or tensors.  The sampler loops hand over their HOST copy of the schedule
        (``_noise_args``): a device scalar here costs a blocking device->host read per query."""
        t0, t1 = _f(self.transform(torch.as_tensor(_f(sigma)))), _f(self.transform(torch.as_tensor(_f(sigma_next))))
        d = abs(t1 - t0)
        # equal end points: the reference divides a zero increment by sqrt(0) (nan); same here instead of a ZeroDivisionError
        return self.tree.increment(t0, t1, mult=1.0 / math.sqrt(d) if d > 0 else float('inf'))


class _Loop:
    """State shared by the sampler loops: host copy of the schedule, working buffer, model calls."""

    def __init__(self, model, x, sigmas, extra_args, callback):
        if not x.is_cuda:
            raise RuntimeError('the samplers run on the HIP path only: x must live on a ROCm device (no CPU fallback)')
        if x.dtype != torch.float32:
            raise TypeError(f'fp32 latents only (got {x.dtype})')
        self.model, self.extra, self.callback = model, ({} if extra_args is None else extra_args), callback
        self.sigmas = sigmas
        self.sig = _host_values(sigmas)      # the ONE device->host transfer -- none when get_sigmas_* kept the host copy
        self.x = x.contiguous()
        self.B = x.shape[0]
        self._owned = False

    def __len__(self):
        return len(self.sig) - 1

    def sigma_rows(self, values):
        """[len(values), B] device table of per-sample sigma vectors (sigma * s_in): row i is the sigma argument of one
        model call of the run.  The model is told so (``prefetch_schedule``): the conditioning chain is a function of
        sigma / class / ... only, so a model that takes the hint runs it once for the whole table instead of once per call."""
        if not len(values):                  # a schedule of the terminal sigma only: no model call, the loop body never runs (the reference's trange(0))
            return torch.empty((0, self.B), device=self.x.device, dtype=torch.float32)
        v = torch.stack([torch.as_tensor(s, dtype=torch.float32).reshape(()) for s in values])
        v = v.pin_memory() if self.x.is_cuda and len(values) else v      # (a pageable .to(device) ends with a stream synchronisation)
        table = v.to(self.x.device, non_blocking=True)[:, None].expand(len(values), self.B).contiguous()
        if _COND_PREFETCH and _COND_SCHEDULE and len(values):
            hint = getattr(self.model, 'prefetch_schedule', None)
            if hint is not None:
                hint(self.x, table, **self.extra)
        return table

    def denoise(self, sigma_row, x=None, next_row=None):
        """model(x, sigma).  ``next_row``: the sigma vector of the NEXT model call, when the loop knows it: the
        conditioning chain of that call (a function of sigma / class / ... only) is then started on a side stream now,
        behind the main chain just enqueued."""
        den = self.model(self.x if x is None else x, sigma_row, **self.extra).contiguous()
        if next_row is not None and _COND_PREFETCH:
            hint = getattr(self.model, 'prefetch_conditioning', None)
            if hint is not None:
                hint(self.x, next_row, **self.extra)
        return den

    def report(self, i, denoised, sigma_hat=None):
        if self.callback is not None:
            s = self.sigmas[i]
            self.callback({'x': self.x, 'i': i, 'sigma': s, 'sigma_hat': s if sigma_hat is None else sigma_hat, 'denoised': denoised})

    def fresh(self):
        return torch.empty_like(self.x)

    def update(self, op, den, in2=None, aux=None, **c):
        """x <- step(op)(x, ...) ; never writes into the caller's tensor, and hands callbacks a tensor
        that later steps do not overwrite."""
        out = self.x if (self._owned and self.callback is None) else self.fresh()
        ops.sampler_step(op, self.x, den, in2=in2, out=out, aux=aux, **{k: _f(v) for k, v in c.items()})
        self.x, self._owned = out, True

    def add_noise(self, noise, c0, c1, c2=1.0):
        self.update(nat.STEP_ADD_NOISE, noise.contiguous(), c0=c0, c1=c1, c2=c2)

    def noise_args(self, noise_sampler, i, j=None):
        """Arguments of ``noise_sampler(sigma_i, sigma_{i+1})``: the caller's schedule entries (device tensors, what the
        reference passes to a user-supplied sampler) -- except for the built-in Brownian sampler, which only needs the values
        and gets the host copies (no per-step device->host sync)."""
        j = i + 1 if j is None else j
        if isinstance(noise_sampler, BrownianTreeNoiseSampler):
            return self.sig[i], self.sig[j]
        return self.sigmas[i], self.sigmas[j]


def _churn_plan(sig, s_churn, s_tmin, s_tmax):
    """Per-step (gamma, sigma_hat) of Karras Alg. 2 (sampling.py:123-125)."""
    n = len(sig) - 1
    gammas = [min(s_churn / n, 2 ** 0.5 - 1) if s_tmin <= sig[i] <= s_tmax else 0. for i in range(n)]
    return gammas, [sig[i] * (g + 1) for i, g in enumerate(gammas)]


def _apply_churn(lp, i, gamma, sigma_hat, s_noise, s_churn):
    """Karras Alg. 2 noise injection.  The reference draws ``eps = randn_like(x)`` on EVERY step (sampling.py:124,165,195) and
    uses it only where gamma > 0.  With churn requested the draw is made on every step too, so the generator advances exactly
    like the reference's (steps outside [s_tmin, s_tmax] included); with s_churn == 0 (the default) no noise is ever used and
    none is drawn -- the one deliberate deviation: code that reads the global generator AFTER such a run sees a stream that
    is len(sigmas) - 1 draws behind the reference's (KDIFF_STRICT_RNG=1 restores the reference's behaviour)."""
    if s_churn > 0 or _STRICT_RNG:
        eps = torch.randn_like(lp.x)
        if gamma > 0:
            lp.add_noise(eps, s_noise, (sigma_hat ** 2 - lp.sig[i] ** 2) ** 0.5)


# --------------------------------------------------------------------------------- samplers

@torch.no_grad()
def sample_euler(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0., s_tmin=0., s_tmax=float('inf'), s_noise=1.):
    """Algorithm 2 (Euler steps) from Karras et al. (2022)."""
    lp = _Loop(model, x, sigmas, extra_args, callback)
    sig = lp.sig
    gammas, hats = _churn_plan(sig, s_churn, s_tmin, s_tmax)
    rows = lp.sigma_rows(hats)
    for i in trange(len(lp), disable=disable):
        _apply_churn(lp, i, gammas[i], hats[i], s_noise, s_churn)
        den = lp.denoise(rows[i], next_row=rows[i + 1] if i + 1 < len(lp) else None)
        lp.report(i, den, hats[i])
        lp.update(nat.STEP_EULER, den, c0=hats[i], c1=sig[i + 1] - hats[i])
    return lp.x


@torch.no_grad()
def sample_euler_ancestral(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1., s_noise=1., noise_sampler=None):
    """Ancestral sampling with Euler method steps."""
    lp = _Loop(model, x, sigmas, extra_args, callback)
    sig = lp.sig
    noise_sampler = default_noise_sampler(lp.x) if noise_sampler is None else noise_sampler
    rows = lp.sigma_rows(sig[:-1])
    for i in trange(len(lp), disable=disable):
        den = lp.denoise(rows[i])
        sigma_down, sigma_up = get_ancestral_step(sig[i], sig[i + 1], eta=eta)
        lp.report(i, den)
        lp.update(nat.STEP_EULER, den, c0=sig[i], c1=sigma_down - sig[i])
        if sig[i + 1] > 0:
            lp.add_noise(noise_sampler(*lp.noise_args(noise_sampler, i)), s_noise, sigma_up)
    return lp.x


@torch.no_grad()
def sample_heun(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0., s_tmin=0., s_tmax=float('inf'), s_noise=1.):
    """Algorithm 2 (Heun steps) from Karras et al. (2022)."""
    lp = _Loop(model, x, sigmas, extra_args, callback)
    sig = lp.sig
    gammas, hats = _churn_plan(sig, s_churn, s_tmin, s_tmax)
    rows, rows_next = lp.sigma_rows(hats), lp.sigma_rows(sig[1:])
    d, x_2 = lp.fresh(), lp.fresh()
    for i in trange(len(lp), disable=disable):
        _apply_churn(lp, i, gammas[i], hats[i], s_noise, s_churn)
        last = sig[i + 1] == 0
        den = lp.denoise(rows[i], next_row=None if last else rows_next[i])
        lp.report(i, den, hats[i])
        dt = sig[i + 1] - hats[i]
        if last:
            lp.update(nat.STEP_EULER, den, c0=hats[i], c1=dt)
        else:
            ops.sampler_step(nat.STEP_HEUN_PRED, lp.x, den, out=x_2, aux=d, c0=_f(hats[i]), c1=_f(dt))
            den_2 = lp.denoise(rows_next[i], x_2, next_row=rows[i + 1] if i + 1 < len(lp) else None)
            lp.update(nat.STEP_HEUN_CORR, den_2, in2=x_2, aux=d, c0=sig[i + 1], c1=dt)
    return lp.x


@torch.no_grad()
def sample_dpm_2(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0., s_tmin=0., s_tmax=float('inf'), s_noise=1.):
    """A sampler inspired by DPM-Solver-2 and Algorithm 2 from Karras et al. (2022)."""
    lp = _Loop(model, x, sigmas, extra_args, callback)
    sig = lp.sig
    gammas, hats = _churn_plan(sig, s_churn, s_tmin, s_tmax)
    mids = [hats[i].log().lerp(sig[i + 1].log(), 0.5).exp() if sig[i + 1] != 0 else sig[i + 1] for i in range(len(lp))]
    rows, rows_mid = lp.sigma_rows(hats), lp.sigma_rows(mids)
    d, x_2 = lp.fresh(), lp.fresh()
    for i in trange(len(lp), disable=disable):
        _apply_churn(lp, i, gammas[i], hats[i], s_noise, s_churn)
        den = lp.denoise(rows[i])
        lp.report(i, den, hats[i])
        if sig[i + 1] == 0:
            lp.update(nat.STEP_EULER, den, c0=hats[i], c1=sig[i + 1] - hats[i])
        else:
            ops.sampler_step(nat.STEP_HEUN_PRED, lp.x, den, out=x_2, aux=d, c0=_f(hats[i]), c1=_f(mids[i] - hats[i]))
            den_2 = lp.denoise(rows_mid[i], x_2)
            lp.update(nat.STEP_EULER_FROM, den_2, in2=x_2, c0=mids[i], c1=sig[i + 1] - hats[i])
    return lp.x


@torch.no_grad()
def sample_dpm_2_ancestral(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1., s_noise=1., noise_sampler=None):
    """Ancestral sampling with DPM-Solver second-order steps."""
    lp = _Loop(model, x, sigmas, extra_args, callback)
    sig = lp.sig
    noise_sampler = default_noise_sampler(lp.x) if noise_sampler is None else noise_sampler
    steps = [get_ancestral_step(sig[i], sig[i + 1], eta=eta) for i in range(len(lp))]
    mids = [sig[i].log().lerp(torch.as_tensor(sd).log(), 0.5).exp() if sd != 0 else torch.zeros(()) for i, (sd, _) in enumerate(steps)]
    rows, rows_mid = lp.sigma_rows(sig[:-1]), lp.sigma_rows(mids)
    d, x_2 = lp.fresh(), lp.fresh()
    for i in trange(len(lp), disable=disable):
        den = lp.denoise(rows[i])
        sigma_down, sigma_up = steps[i]
        lp.report(i, den)
        if sigma_down == 0:
            lp.update(nat.STEP_EULER, den, c0=sig[i], c1=sigma_down - sig[i])
        else:
            ops.sampler_step(nat.STEP_HEUN_PRED, lp.x, den, out=x_2, aux=d, c0=_f(sig[i]), c1=_f(mids[i] - sig[i]))
            den_2 = lp.denoise(rows_mid[i], x_2)
            lp.update(nat.STEP_EULER_FROM, den_2, in2=x_2, c0=mids[i], c1=sigma_down - sig[i])
            lp.add_noise(noise_sampler(*lp.noise_args(noise_sampler, i)), s_noise, sigma_up)
    return lp.x


def linear_multistep_coeff(order, t, i, j):
    """Adams-Bashforth weight of history entry j at step i (host quadrature, sampling.py:247-257)."""
    from scipy import integrate
    if order - 1 > i:
        raise ValueError(f'Order {order} too high for step {i}')

    def lagrange(tau):
        out = 1.
        for k in range(order):
            if k != j:
                out *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return out
    return integrate.quad(lagrange, t[i], t[i + 1], epsrel=1e-4)[0]


@torch.no_grad()
def sample_lms(model, x, sigmas, extra_args=None, callback=None, disable=None, order=4):
    lp = _Loop(model, x, sigmas, extra_args, callback)
    sig = lp.sig
    t = sig.numpy()
    rows = lp.sigma_rows(sig[:-1])
    ds = []
    for i in trange(len(lp), disable=disable):
        den = lp.denoise(rows[i])
        ds.append(ops.sampler_step(nat.STEP_TO_D, lp.x, den, c0=_f(sig[i])))
        ds = ds[-order:]
        lp.report(i, den)
        cur = min(i + 1, order)
        coeffs = [linear_multistep_coeff(cur, t, i, j) for j in range(cur)]
        hist = list(reversed(ds))
        if cur == 1:
            lp.update(nat.STEP_AXPY, hist[0], c0=coeffs[0])
        else:
            acc = ops.sampler_step(nat.STEP_LERP2, None, hist[0], in2=hist[1], c0=coeffs[0], c1=coeffs[1])
            for c, dj in zip(coeffs[2:], hist[2:]):
                ops.sampler_step(nat.STEP_AXPY, acc, dj, out=acc, c0=c)
            lp.update(nat.STEP_AXPY, acc, c0=1.0)
    return lp.x


# ---- DPM-Solver (fixed-step "fast" and adaptive), sampling.py:304-507 -----------------------------------------------

class PIDStepSizeController:
    """PID controller for the adaptive solver's step size (sampling.py:304-330): host scalars only."""

    def __init__(self, h, pcoeff, icoeff, dcoeff, order=1, accept_safety=0.81, eps=1e-8):
        self.h = h
        self.b1 = (pcoeff + icoeff + dcoeff) / order
        self.b2 = -(pcoeff + 2 * dcoeff) / order
        self.b3 = dcoeff / order
        self.accept_safety, self.eps = accept_safety, eps
        self.errs = []

    def limiter(self, x):
        return 1 + math.atan(x - 1)

    def propose_step(self, error):
        inv_error = 1 / (float(error) + self.eps)
        if not self.errs:
            self.errs = [inv_error] * 3
        self.errs[0] = inv_error
        factor = self.limiter(self.errs[0] ** self.b1 * self.errs[1] ** self.b2 * self.errs[2] ** self.b3)
        accept = factor >= self.accept_safety
        if accept:
            self.errs[2], self.errs[1] = self.errs[1], self.errs[0]
        self.h *= factor
        return accept


class DPMSolver:
    """DPM-Solver-1/2/3 (https://arxiv.org/abs/2206.00927) behind the reference's class surface (sampling.py:333-480).

    Same formulation as the reference (eps = (x - D) / sigma, states x - a eps - b (eps_r - eps)) and the same rounding
    order, but each tensor expression is ONE fused HIP launch (``kd_dpm_eps_f32`` / ``kd_dpm_combine_f32``) fed with the
    reference's own 0-dim fp32 scalar products computed on the host; the adaptive solver's error norm is one more launch
    (``kd_dpm_error_f32``) and the step's only device -> host value.  Given the same denoised tensors the states are
    bit-identical to the reference's, which matters for the adaptive solver: its accept / reject decisions amplify
    rounding differences."""

    def __init__(self, model, extra_args=None, eps_callback=None, info_callback=None):
        self.model = model
        self.extra_args = {} if extra_args is None else extra_args
        self.eps_callback, self.info_callback = eps_callback, info_callback

    def t(self, sigma):
        return -torch.as_tensor(sigma, dtype=torch.float32).log()

    def sigma(self, t):
        return torch.as_tensor(t, dtype=torch.float32).neg().exp()

    def eps(self, eps_cache, key, x, t, *args, **kwargs):
        """eps(x, t) memoised per step under ``key`` (:350-357).  Returns (eps, cache)."""
        if key in eps_cache:
            return eps_cache[key], eps_cache
        if not x.is_cuda:
            raise RuntimeError('the samplers run on the HIP path only: x must live on a ROCm device (no CPU fallback)')
        x = x.contiguous()
        sig = torch.full((x.shape[0],), _f(self.sigma(t)), device=x.device, dtype=torch.float32)
        den = self.model(x, sig, *args, **self.extra_args, **kwargs).contiguous()
        eps = ops.dpm_eps(x, den, _f(self.sigma(t)))
        if self.eps_callback is not None:
            self.eps_callback()
        return eps, {key: eps, **eps_cache}

    def dpm_solver_1_step(self, x, t, t_next, eps_cache=None):
        eps_cache = {} if eps_cache is None else eps_cache
        h = t_next - t
        eps, eps_cache = self.eps(eps_cache, 'eps', x, t)
        return ops.dpm_combine(x.contiguous(), eps, _f(self.sigma(t_next) * h.expm1())), eps_cache

    def dpm_solver_2_step(self, x, t, t_next, r1=1 / 2, eps_cache=None):
        eps_cache = {} if eps_cache is None else eps_cache
        x = x.contiguous()
        h = t_next - t
        eps, eps_cache = self.eps(eps_cache, 'eps', x, t)
        s1 = t + r1 * h
        u1 = ops.dpm_combine(x, eps, _f(self.sigma(s1) * (r1 * h).expm1()))
        eps_r1, eps_cache = self.eps(eps_cache, 'eps_r1', u1, s1)
        x_2 = ops.dpm_combine(x, eps, _f(self.sigma(t_next) * h.expm1()), eps_r1, _f(self.sigma(t_next) / (2 * r1) * h.expm1()))
        return x_2, eps_cache

    def dpm_solver_3_step(self, x, t, t_next, r1=1 / 3, r2=2 / 3, eps_cache=None):
        eps_cache = {} if eps_cache is None else eps_cache
        x = x.contiguous()
        h = t_next - t
        eps, eps_cache = self.eps(eps_cache, 'eps', x, t)
        s1, s2 = t + r1 * h, t + r2 * h
        u1 = ops.dpm_combine(x, eps, _f(self.sigma(s1) * (r1 * h).expm1()))
        eps_r1, eps_cache = self.eps(eps_cache, 'eps_r1', u1, s1)
        u2 = ops.dpm_combine(x, eps, _f(self.sigma(s2) * (r2 * h).expm1()), eps_r1,
                             _f(self.sigma(s2) * (r2 / r1) * ((r2 * h).expm1() / (r2 * h) - 1)))
        eps_r2, eps_cache = self.eps(eps_cache, 'eps_r2', u2, s2)
        x_3 = ops.dpm_combine(x, eps, _f(self.sigma(t_next) * h.expm1()), eps_r2, _f(self.sigma(t_next) / r2 * (h.expm1() / h - 1)))
        return x_3, eps_cache

    # -- drivers -----------------------------------------------------------------------------------------------------------
    def _ancestral(self, t, t_next, t_end, eta):
        """Deterministic target time and the noise scale added back (:411-416, :448-453)."""
        if not eta:
            return t_next, 0.
        sd, su = get_ancestral_step(self.sigma(t), self.sigma(t_next), eta)
        t_next_ = torch.minimum(t_end, self.t(sd))
        return t_next_, (self.sigma(t_next) ** 2 - self.sigma(t_next_) ** 2) ** 0.5

    def _noised(self, x, su, s_noise, noise):
        out = torch.empty_like(x)
        ops.sampler_step(nat.STEP_ADD_NOISE, x, noise.contiguous(), out=out, c0=_f(su), c1=_f(s_noise), c2=1.0)
        return out

    @torch.no_grad()
    def dpm_solver_fast(self, x, t_start, t_end, nfe, eta=0., s_noise=1., noise_sampler=None):
        noise_sampler = default_noise_sampler(x) if noise_sampler is None else noise_sampler
        t_start, t_end = torch.as_tensor(t_start, dtype=torch.float32), torch.as_tensor(t_end, dtype=torch.float32)
        if not t_end > t_start and eta:
            raise ValueError('eta must be 0 for reverse sampling')
        m = math.floor(nfe / 3) + 1
        ts = torch.linspace(t_start, t_end, m + 1)
        orders = [3] * (m - 2) + [2, 1] if nfe % 3 == 0 else [3] * (m - 1) + [nfe % 3]
        x = x.contiguous()
        for i, order in enumerate(orders):
            t, t_next = ts[i], ts[i + 1]
            t_next_, su = self._ancestral(t, t_next, t_end, eta)
            eps, eps_cache = self.eps({}, 'eps', x, t)
            if self.info_callback is not None:
                self.info_callback({'x': x, 'i': i, 't': ts[i], 't_up': t, 'denoised': ops.dpm_combine(x, eps, _f(self.sigma(t)))})
            if order == 1:
                x, eps_cache = self.dpm_solver_1_step(x, t, t_next_, eps_cache=eps_cache)
            elif order == 2:
                x, eps_cache = self.dpm_solver_2_step(x, t, t_next_, eps_cache=eps_cache)
            else:
                x, eps_cache = self.dpm_solver_3_step(x, t, t_next_, eps_cache=eps_cache)
            if eta:
                x = self._noised(x, su, s_noise, noise_sampler(self.sigma(t), self.sigma(t_next)))
        return x

    @torch.no_grad()
    def dpm_solver_adaptive(self, x, t_start, t_end, order=3, rtol=0.05, atol=0.0078, h_init=0.05, pcoeff=0., icoeff=1., dcoeff=0.,
                            accept_safety=0.81, eta=0., s_noise=1., noise_sampler=None):
        noise_sampler = default_noise_sampler(x) if noise_sampler is None else noise_sampler
        if order not in {2, 3}:
            raise ValueError('order should be 2 or 3')
        t_start, t_end = torch.as_tensor(t_start, dtype=torch.float32), torch.as_tensor(t_end, dtype=torch.float32)
        forward = bool(t_end > t_start)
        if not forward and eta:
            raise ValueError('eta must be 0 for reverse sampling')
        h_init = abs(h_init) * (1 if forward else -1)
        s, x = t_start, x.contiguous()
        x_prev = x
        pid = PIDStepSizeController(h_init, pcoeff, icoeff, dcoeff, 1.5 if eta else order, accept_safety)
        info = {'steps': 0, 'nfe': 0, 'n_accept': 0, 'n_reject': 0}
        while (s < t_end - 1e-5) if forward else (s > t_end + 1e-5):
            t = torch.minimum(t_end, s + pid.h) if forward else torch.maximum(t_end, s + pid.h)
            t_, su = self._ancestral(s, t, t_end, eta)
            eps, eps_cache = self.eps({}, 'eps', x, s)
            denoised = ops.dpm_combine(x, eps, _f(self.sigma(s))) if self.info_callback is not None else None
            if order == 2:
                x_low, eps_cache = self.dpm_solver_1_step(x, s, t_, eps_cache=eps_cache)
                x_high, eps_cache = self.dpm_solver_2_step(x, s, t_, eps_cache=eps_cache)
            else:
                x_low, eps_cache = self.dpm_solver_2_step(x, s, t_, r1=1 / 3, eps_cache=eps_cache)
                x_high, eps_cache = self.dpm_solver_3_step(x, s, t_, eps_cache=eps_cache)
            error = ops.dpm_error(x_low, x_high, x_prev, atol, rtol)        # the step's one device -> host value
            accept = pid.propose_step(error)
            if accept:
                x_prev = x_low
                x = self._noised(x_high, su, s_noise, noise_sampler(self.sigma(s), self.sigma(t))) if eta else x_high
                s = t
                info['n_accept'] += 1
            else:
                info['n_reject'] += 1
            info['nfe'] += order
            info['steps'] += 1
            if self.info_callback is not None:
                self.info_callback({'x': x, 'i': info['steps'] - 1, 't': s, 't_up': s, 'denoised': denoised, 'error': error, 'h': pid.h, **info})
        return x, info


def _dpm_solver_for(model, extra_args, callback, pbar):
    solver = DPMSolver(model, extra_args, eps_callback=pbar.update)
    if callback is not None:
        solver.info_callback = lambda info: callback({'sigma': solver.sigma(info['t']), 'sigma_hat': solver.sigma(info['t_up']), **info})
    return solver


@torch.no_grad()
def sample_dpm_fast(model, x, sigma_min, sigma_max, n, extra_args=None, callback=None, disable=None, eta=0., s_noise=1., noise_sampler=None):
    """DPM-Solver-Fast (fixed step size)."""
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError('sigma_min and sigma_max must not be 0')
    with tqdm(total=n, disable=disable) as pbar:
        solver = _dpm_solver_for(model, extra_args, callback, pbar)
        return solver.dpm_solver_fast(x, solver.t(torch.tensor(sigma_max)), solver.t(torch.tensor(sigma_min)), n, eta, s_noise, noise_sampler)


@torch.no_grad()
def sample_dpm_adaptive(model, x, sigma_min, sigma_max, extra_args=None, callback=None, disable=None, order=3, rtol=0.05, atol=0.0078,
                        h_init=0.05, pcoeff=0., icoeff=1., dcoeff=0., accept_safety=0.81, eta=0., s_noise=1., noise_sampler=None,
                        return_info=False):
    """DPM-Solver-12 and 23 (adaptive step size)."""
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError('sigma_min and sigma_max must not be 0')
    with tqdm(disable=disable) as pbar:
        solver = _dpm_solver_for(model, extra_args, callback, pbar)
        x, info = solver.dpm_solver_adaptive(x, solver.t(torch.tensor(sigma_max)), solver.t(torch.tensor(sigma_min)), order, rtol, atol,
                                             h_init, pcoeff, icoeff, dcoeff, accept_safety, eta, s_noise, noise_sampler)
    return (x, info) if return_info else x


@torch.no_grad()
def sample_dpmpp_2s_ancestral(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1., s_noise=1., noise_sampler=None):
    """Ancestral sampling with DPM-Solver++(2S) second-order steps."""
    lp = _Loop(model, x, sigmas, extra_args, callback)
    sig = lp.sig
    noise_sampler = default_noise_sampler(lp.x) if noise_sampler is None else noise_sampler
    sigma_fn, t_fn = (lambda t: t.neg().exp()), (lambda s: s.log().neg())
    plan = []
    for i in range(len(lp)):
        sigma_down, sigma_up = get_ancestral_step(sig[i], sig[i + 1], eta=eta)
        if sigma_down == 0:
            plan.append((sigma_down, sigma_up, None))
        else:
            t, t_next = t_fn(sig[i]), t_fn(torch.as_tensor(sigma_down))
            h = t_next - t
            s = t + 0.5 * h
            plan.append((sigma_down, sigma_up, (sigma_fn(s), sigma_fn(s) / sigma_fn(t), (-h * 0.5).expm1(),
                                                sigma_fn(t_next) / sigma_fn(t), (-h).expm1())))
    rows = lp.sigma_rows(sig[:-1])
    rows_s = lp.sigma_rows([p[2][0] if p[2] is not None else torch.zeros(()) for p in plan])
    x_2 = lp.fresh()
    for i in trange(len(lp), disable=disable):
        den = lp.denoise(rows[i])
        sigma_down, sigma_up, second = plan[i]
        lp.report(i, den)
        if second is None:
            lp.update(nat.STEP_EULER, den, c0=sig[i], c1=sigma_down - sig[i])
        else:
            _, a1, b1, a2, b2 = second
            ops.sampler_step(nat.STEP_DPMPP_2M1, lp.x, den, out=x_2, c0=_f(a1), c1=_f(b1))
            den_2 = lp.denoise(rows_s[i], x_2)
            lp.update(nat.STEP_DPMPP_2M1, den_2, c0=a2, c1=b2)
        if sig[i + 1] > 0:
            lp.add_noise(noise_sampler(*lp.noise_args(noise_sampler, i)), s_noise, sigma_up)
    return lp.x


@torch.no_grad()
def sample_dpmpp_sde(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1., s_noise=1., noise_sampler=None, r=1 / 2):
    """DPM-Solver++ (stochastic)."""
    lp = _Loop(model, x, sigmas, extra_args, callback)
    sig = lp.sig
    if noise_sampler is None:
        noise_sampler = BrownianTreeNoiseSampler(lp.x, sig[sig > 0].min(), sig.max())
    sigma_fn, t_fn = (lambda t: t.neg().exp()), (lambda s: s.log().neg())
    mids = []
    for i in range(len(lp)):
        if sig[i + 1] == 0:
            mids.append(torch.zeros(()))
        else:
            t, t_next = t_fn(sig[i]), t_fn(sig[i + 1])
            mids.append(sigma_fn(t + (t_next - t) * r))
    rows, rows_mid = lp.sigma_rows(sig[:-1]), lp.sigma_rows(mids)
    x_2, den_d = lp.fresh(), lp.fresh()
    for i in trange(len(lp), disable=disable):
        den = lp.denoise(rows[i])
        lp.report(i, den)
        if sig[i + 1] == 0:
            lp.update(nat.STEP_EULER, den, c0=sig[i], c1=sig[i + 1] - sig[i])
            continue
        t, t_next = t_fn(sig[i]), t_fn(sig[i + 1])
        h = t_next - t
        s = t + h * r
        fac = 1 / (2 * r)
        # step 1: ancestral sub-step to the midpoint
        sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(s), eta)
        s_ = t_fn(torch.as_tensor(sd))
        ops.sampler_step(nat.STEP_DPMPP_2M1, lp.x, den, out=x_2, c0=_f(sigma_fn(s_) / sigma_fn(t)), c1=_f((t - s_).expm1()))
        ops.sampler_step(nat.STEP_ADD_NOISE, x_2, noise_sampler(sigma_fn(t), sigma_fn(s)).contiguous(), out=x_2, c0=_f(s_noise), c1=_f(su))
        den_2 = lp.denoise(rows_mid[i], x_2)
        # step 2: full ancestral step with the blended denoised
        sd, su = get_ancestral_step(sigma_fn(t), sigma_fn(t_next), eta)
        t_next_ = t_fn(torch.as_tensor(sd))
        ops.sampler_step(nat.STEP_LERP2, None, den, in2=den_2, out=den_d, c0=1 - fac, c1=fac)
        lp.update(nat.STEP_DPMPP_2M1, den_d, c0=sigma_fn(t_next_) / sigma_fn(t), c1=(t - t_next_).expm1())
        lp.add_noise(noise_sampler(sigma_fn(t), sigma_fn(t_next)), s_noise, su)
    return lp.x


@torch.no_grad()
def sample_dpmpp_2m(model, x, sigmas, extra_args=None, callback=None, disable=None):
    """DPM-Solver++(2M)."""
    lp = _Loop(model, x, sigmas, extra_args, callback)
    sig = lp.sig
    sigma_fn, t_fn = (lambda t: t.neg().exp()), (lambda s: s.log().neg())
    rows = lp.sigma_rows(sig[:-1])
    old = None
    for i in trange(len(lp), disable=disable):
        den = lp.denoise(rows[i], next_row=rows[i + 1] if i + 1 < len(lp) else None)
        lp.report(i, den)
        t, t_next = t_fn(sig[i]), t_fn(sig[i + 1])
        h = t_next - t
        a, b = sigma_fn(t_next) / sigma_fn(t), (-h).expm1()
        if old is None or sig[i + 1] == 0:
            lp.update(nat.STEP_DPMPP_2M1, den, c0=a, c1=b)
        else:
            ratio = (t - t_fn(sig[i - 1])) / h
            lp.update(nat.STEP_DPMPP_2M2, den, in2=old, c0=a, c1=b, c2=1 + 1 / (2 * ratio), c3=1 / (2 * ratio))
        old = den
    return lp.x


@torch.no_grad()
def sample_dpmpp_2m_sde(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1., s_noise=1., noise_sampler=None, solver_type='midpoint'):
    """DPM-Solver++(2M) SDE."""
    if solver_type not in {'heun', 'midpoint'}:
        raise ValueError('solver_type must be \'heun\' or \'midpoint\'')
    lp = _Loop(model, x, sigmas, extra_args, callback)
    sig = lp.sig
    if noise_sampler is None:
        noise_sampler = BrownianTreeNoiseSampler(lp.x, sig[sig > 0].min(), sig.max())
    rows = lp.sigma_rows(sig[:-1])
    old, h_last = None, None
    for i in trange(len(lp), disable=disable):
        den = lp.denoise(rows[i])
        lp.report(i, den)
        if sig[i + 1] == 0:
            lp.update(nat.STEP_AXPBY, den, c0=0.0, c1=1.0)               # x = denoised
        else:
            t, s = -sig[i].log(), -sig[i + 1].log()
            h = s - t
            eta_h = eta * h
            lp.update(nat.STEP_AXPBY, den, c0=sig[i + 1] / sig[i] * (-eta_h).exp(), c1=(-h - eta_h).expm1().neg())
            if old is not None:
                ratio = h_last / h
                if solver_type == 'heun':
                    coef = ((-h - eta_h).expm1().neg() / (-h - eta_h) + 1) * (1 / ratio)
                else:
                    coef = 0.5 * (-h - eta_h).expm1().neg() * (1 / ratio)
                lp.update(nat.STEP_ADD_DIFF, den, in2=old, c0=coef)
            if eta:
                lp.add_noise(noise_sampler(*lp.noise_args(noise_sampler, i)), sig[i + 1], (-2 * eta_h).expm1().neg().sqrt(), s_noise)
            h_last = h
        old = den
        if sig[i + 1] == 0:
            h_last = None
    return lp.x


@torch.no_grad()
def sample_dpmpp_3m_sde(model, x, sigmas, extra_args=None, callback=None, disable=None, eta=1., s_noise=1., noise_sampler=None):
    """DPM-Solver++(3M) SDE (sampling.py:655-700).  The two history corrections phi_2 * d1 - phi_3 * d2 are folded on the
    host into one coefficient per denoised difference (x += cu * (D_i - D_{i-1}) + cv * (D_{i-1} - D_{i-2})), so a step is
    three fused launches plus the noise injection."""
    lp = _Loop(model, x, sigmas, extra_args, callback)
    sig = lp.sig
    if noise_sampler is None:
        noise_sampler = BrownianTreeNoiseSampler(lp.x, sig[sig > 0].min(), sig.max())
    rows = lp.sigma_rows(sig[:-1])
    den_1, den_2, h_1, h_2 = None, None, None, None
    for i in trange(len(lp), disable=disable):
        den = lp.denoise(rows[i], next_row=rows[i + 1] if i + 1 < len(lp) else None)
        lp.report(i, den)
        if sig[i + 1] == 0:
            lp.update(nat.STEP_AXPBY, den, c0=0.0, c1=1.0)               # x = denoised
            h = None
        else:
            t, s = -sig[i].log(), -sig[i + 1].log()
            h = s - t
            h_eta = h * (eta + 1)
            lp.update(nat.STEP_AXPBY, den, c0=torch.exp(-h_eta), c1=(-h_eta).expm1().neg())
            if h_2 is not None:
                r0, r1 = h_1 / h, h_2 / h
                phi_2 = h_eta.neg().expm1() / h_eta + 1
                phi_3 = phi_2 / h_eta - 0.5
                k = (phi_2 * r0 - phi_3) / (r0 + r1)
                lp.update(nat.STEP_ADD_DIFF, den, in2=den_1, c0=phi_2 / r0 + k / r0)
                lp.update(nat.STEP_ADD_DIFF, den_1, in2=den_2, c0=-k / r1)
            elif h_1 is not None:
                r = h_1 / h
                phi_2 = h_eta.neg().expm1() / h_eta + 1
                lp.update(nat.STEP_ADD_DIFF, den, in2=den_1, c0=phi_2 / r)
            if eta:
                lp.add_noise(noise_sampler(*lp.noise_args(noise_sampler, i)), sig[i + 1], (-2 * h * eta).expm1().neg().sqrt(), s_noise)
        den_1, den_2 = den, den_1
        h_1, h_2 = h, h_1
    return lp.x


def make_cfg_model_fn(model, cfg_scale, num_classes):
    """Classifier-free guidance wrapper of the reference's demo / evaluation path (train.py:333-344): the batch is
    doubled (unconditional class id ``num_classes`` first), and the two halves are combined as
    uncond + (cond - uncond) * cfg_scale -- here in one fused HIP launch."""
    if cfg_scale == 1:
        return model

    def cfg_model_fn(x, sigma, class_cond):
        x_in = torch.cat([x, x])
        sigma_in = torch.cat([sigma, sigma])
        class_in = torch.cat([torch.full_like(class_cond, num_classes), class_cond])
        out = model(x_in, sigma_in, class_cond=class_in)
        out_uncond, out_cond = out.chunk(2)
        out_uncond, out_cond = out_uncond.contiguous(), out_cond.contiguous()
        return ops.sampler_step(nat.STEP_ADD_DIFF, out_uncond, out_cond, in2=out_uncond, c0=float(cfg_scale))
    return cfg_model_fn

"""CPU oracle (test infrastructure): the Karras schedule, preconditioner and solver loops.

Restates /root/reference/k_diffusion/sampling.py and layers.py:70-90 in plain torch on the CPU.
Pinned against the real reference by tests/test_oracle_vs_golden.py (known answers recorded by
oracle/make_golden.py).  ``model`` is any callable ``model(x, sigma[B], **extra) -> denoised``.

The solver arithmetic is written with explicit 0-dim fp32 tensors in the same operation order as
the reference so that results are bit-identical on the CPU.
"""
import math

import torch


def sigmas_karras(n, sigma_min, sigma_max, rho=7.0):
    """sampling.py:17-23 (+ append_zero :13-14).  fp32 ramp, python-double scalar powers."""
    ramp = torch.linspace(0, 1, n)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    s = (hi + ramp * (lo - hi)) ** rho
    return torch.cat([s, s.new_zeros([1])])


def sigmas_exponential(n, sigma_min, sigma_max):
    """sampling.py:26-29."""
    s = torch.linspace(math.log(sigma_max), math.log(sigma_min), n).exp()
    return torch.cat([s, s.new_zeros([1])])


def sigmas_polyexponential(n, sigma_min, sigma_max, rho=1.0):
    """sampling.py:32-36."""
    ramp = torch.linspace(1, 0, n) ** rho
    s = torch.exp(ramp * (math.log(sigma_max) - math.log(sigma_min)) + math.log(sigma_min))
    return torch.cat([s, s.new_zeros([1])])


def sigmas_vp(n, beta_d=19.9, beta_min=0.1, eps_s=1e-3):
    """sampling.py:39-43."""
    t = torch.linspace(1, eps_s, n)
    s = torch.sqrt(torch.exp(beta_d * t ** 2 / 2 + beta_min * t) - 1)
    return torch.cat([s, s.new_zeros([1])])


def _bc(v, x):
    return v.reshape(v.shape + (1,) * (x.ndim - v.ndim)) if torch.is_tensor(v) and v.ndim else v


def to_d(x, sigma, denoised):
    """sampling.py:46-48."""
    return (x - denoised) / _bc(sigma, x)


def ancestral_step(sigma_from, sigma_to, eta=1.0):
    """sampling.py:51-58."""
    if not eta:
        return sigma_to, 0.0
    up = min(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    down = (sigma_to ** 2 - up ** 2) ** 0.5
    return down, up


def karras_scalings(sigma, sigma_data):
    """layers.py:70-74 -> (c_skip, c_out, c_in)."""
    var = sigma ** 2 + sigma_data ** 2
    return sigma_data ** 2 / var, sigma * sigma_data / var ** 0.5, 1 / var ** 0.5


def denoiser(inner, sigma_data):
    """layers.py:88-90 as a closure: D(x, sigma) = F(x * c_in, sigma) * c_out + x * c_skip."""
    def model(x, sigma, **kw):
        c_skip, c_out, c_in = (_bc(c, x) for c in karras_scalings(sigma, sigma_data))
        return inner(x * c_in, sigma, **kw) * c_out + x * c_skip
    return model


def _report(callback, x, i, sigma, sigma_hat, denoised):
    if callback is not None:
        callback({"x": x, "i": i, "sigma": sigma, "sigma_hat": sigma_hat, "denoised": denoised})


def _churn(x, sigmas, i, s_churn, s_tmin, s_tmax, s_noise):
    """sampling.py:123-127 (shared by euler/heun/dpm_2).  Always draws eps (RNG parity)."""
    gamma = min(s_churn / (len(sigmas) - 1), 2 ** 0.5 - 1) if s_tmin <= sigmas[i] <= s_tmax else 0.0
    eps = torch.randn_like(x) * s_noise
    sigma_hat = sigmas[i] * (gamma + 1)
    if gamma > 0:
        x = x + eps * (sigma_hat ** 2 - sigmas[i] ** 2) ** 0.5
    return x, sigma_hat


def sample_euler(model, x, sigmas, extra_args=None, callback=None, s_churn=0.0, s_tmin=0.0,
                 s_tmax=float("inf"), s_noise=1.0):
    """sampling.py:117-135."""
    extra = extra_args or {}
    ones = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        x, sigma_hat = _churn(x, sigmas, i, s_churn, s_tmin, s_tmax, s_noise)
        den = model(x, sigma_hat * ones, **extra)
        d = to_d(x, sigma_hat, den)
        _report(callback, x, i, sigmas[i], sigma_hat, den)
        x = x + d * (sigmas[i + 1] - sigma_hat)
    return x


def sample_heun(model, x, sigmas, extra_args=None, callback=None, s_churn=0.0, s_tmin=0.0,
                s_tmax=float("inf"), s_noise=1.0):
    """sampling.py:158-184."""
    extra = extra_args or {}
    ones = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        x, sigma_hat = _churn(x, sigmas, i, s_churn, s_tmin, s_tmax, s_noise)
        den = model(x, sigma_hat * ones, **extra)
        d = to_d(x, sigma_hat, den)
        _report(callback, x, i, sigmas[i], sigma_hat, den)
        dt = sigmas[i + 1] - sigma_hat
        if sigmas[i + 1] == 0:
            x = x + d * dt
        else:
            x_2 = x + d * dt
            den_2 = model(x_2, sigmas[i + 1] * ones, **extra)
            d_2 = to_d(x_2, sigmas[i + 1], den_2)
            x = x + ((d + d_2) / 2) * dt
    return x


def sample_dpm_2(model, x, sigmas, extra_args=None, callback=None, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0):
    """sampling.py:187-215: Euler half step to the log-midpoint sigma, second slope taken there."""
    extra = extra_args or {}
    ones = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        x, sigma_hat = _churn(x, sigmas, i, s_churn, s_tmin, s_tmax, s_noise)
        den = model(x, sigma_hat * ones, **extra)
        d = to_d(x, sigma_hat, den)
        _report(callback, x, i, sigmas[i], sigma_hat, den)
        if sigmas[i + 1] == 0:
            x = x + d * (sigmas[i + 1] - sigma_hat)
        else:
            sigma_mid = sigma_hat.log().lerp(sigmas[i + 1].log(), 0.5).exp()
            x_2 = x + d * (sigma_mid - sigma_hat)
            d_2 = to_d(x_2, sigma_mid, model(x_2, sigma_mid * ones, **extra))
            x = x + d_2 * (sigmas[i + 1] - sigma_hat)
    return x


def sample_dpm_2_ancestral(model, x, sigmas, noise_sampler, extra_args=None, callback=None, eta=1.0, s_noise=1.0):
    """sampling.py:218-245."""
    extra = extra_args or {}
    ones = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        den = model(x, sigmas[i] * ones, **extra)
        sigma_down, sigma_up = ancestral_step(sigmas[i], sigmas[i + 1], eta)
        _report(callback, x, i, sigmas[i], sigmas[i], den)
        d = to_d(x, sigmas[i], den)
        if sigma_down == 0:
            x = x + d * (sigma_down - sigmas[i])
        else:
            sigma_mid = sigmas[i].log().lerp(sigma_down.log(), 0.5).exp()
            x_2 = x + d * (sigma_mid - sigmas[i])
            d_2 = to_d(x_2, sigma_mid, model(x_2, sigma_mid * ones, **extra))
            x = x + d_2 * (sigma_down - sigmas[i])
            x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * s_noise * sigma_up
    return x


def sample_dpmpp_2s_ancestral(model, x, sigmas, noise_sampler, extra_args=None, callback=None, eta=1.0, s_noise=1.0):
    """sampling.py:508-539."""
    extra = extra_args or {}
    ones = x.new_ones([x.shape[0]])
    sigma_fn, t_fn = (lambda t: t.neg().exp()), (lambda s: s.log().neg())
    for i in range(len(sigmas) - 1):
        den = model(x, sigmas[i] * ones, **extra)
        sigma_down, sigma_up = ancestral_step(sigmas[i], sigmas[i + 1], eta)
        _report(callback, x, i, sigmas[i], sigmas[i], den)
        if sigma_down == 0:
            x = x + to_d(x, sigmas[i], den) * (sigma_down - sigmas[i])
        else:
            t, t_next = t_fn(sigmas[i]), t_fn(sigma_down)
            h = t_next - t
            s = t + 0.5 * h
            x_2 = (sigma_fn(s) / sigma_fn(t)) * x - (-h * 0.5).expm1() * den
            den_2 = model(x_2, sigma_fn(s) * ones, **extra)
            x = (sigma_fn(t_next) / sigma_fn(t)) * x - (-h).expm1() * den_2
        if sigmas[i + 1] > 0:
            x = x + noise_sampler(sigmas[i], sigmas[i + 1]) * s_noise * sigma_up
    return x


def sample_dpmpp_2m(model, x, sigmas, extra_args=None, callback=None):
    """sampling.py:584-607."""
    extra = extra_args or {}
    ones = x.new_ones([x.shape[0]])
    t_of = lambda s: s.log().neg()
    s_of = lambda t: t.neg().exp()
    old = None
    for i in range(len(sigmas) - 1):
        den = model(x, sigmas[i] * ones, **extra)
        _report(callback, x, i, sigmas[i], sigmas[i], den)
        t, t_next = t_of(sigmas[i]), t_of(sigmas[i + 1])
        h = t_next - t
        if old is None or sigmas[i + 1] == 0:
            x = (s_of(t_next) / s_of(t)) * x - (-h).expm1() * den
        else:
            r = (t - t_of(sigmas[i - 1])) / h
            den_d = (1 + 1 / (2 * r)) * den - (1 / (2 * r)) * old
            x = (s_of(t_next) / s_of(t)) * x - (-h).expm1() * den_d
        old = den
    return x


def sample_dpmpp_sde(model, x, sigmas, noise_sampler, extra_args=None, callback=None, eta=1.0,
                     s_noise=1.0, r=0.5):
    """sampling.py:542-581 with an explicit ``noise_sampler(sigma, sigma_next)``."""
    extra = extra_args or {}
    ones = x.new_ones([x.shape[0]])
    t_of = lambda s: s.log().neg()
    s_of = lambda t: t.neg().exp()
    for i in range(len(sigmas) - 1):
        den = model(x, sigmas[i] * ones, **extra)
        _report(callback, x, i, sigmas[i], sigmas[i], den)
        if sigmas[i + 1] == 0:
            x = x + to_d(x, sigmas[i], den) * (sigmas[i + 1] - sigmas[i])
            continue
        t, t_next = t_of(sigmas[i]), t_of(sigmas[i + 1])
        h = t_next - t
        s = t + h * r
        fac = 1 / (2 * r)
        sd, su = ancestral_step(s_of(t), s_of(s), eta)
        s_ = t_of(sd)
        x_2 = (s_of(s_) / s_of(t)) * x - (t - s_).expm1() * den
        x_2 = x_2 + noise_sampler(s_of(t), s_of(s)) * s_noise * su
        den_2 = model(x_2, s_of(s) * ones, **extra)
        sd, su = ancestral_step(s_of(t), s_of(t_next), eta)
        t_next_ = t_of(sd)
        den_d = (1 - fac) * den + fac * den_2
        x = (s_of(t_next_) / s_of(t)) * x - (t - t_next_).expm1() * den_d
        x = x + noise_sampler(s_of(t), s_of(t_next)) * s_noise * su
    return x


def lms_coeff(order, t, i, j):
    """sampling.py:247-257."""
    from scipy import integrate
    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")

    def basis(tau):
        p = 1.0
        for k in range(order):
            if k != j:
                p *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return p
    return integrate.quad(basis, t[i], t[i + 1], epsrel=1e-4)[0]


def sample_lms(model, x, sigmas, extra_args=None, callback=None, order=4):
    """sampling.py:260-277."""
    extra = extra_args or {}
    ones = x.new_ones([x.shape[0]])
    t = sigmas.detach().cpu().numpy()
    hist = []
    for i in range(len(sigmas) - 1):
        den = model(x, sigmas[i] * ones, **extra)
        hist.append(to_d(x, sigmas[i], den))
        hist = hist[-order:]
        _report(callback, x, i, sigmas[i], sigmas[i], den)
        cur = min(i + 1, order)
        cs = [lms_coeff(cur, t, i, j) for j in range(cur)]
        x = x + sum(c * d for c, d in zip(cs, reversed(hist)))
    return x


# ---- DPM-Solver (fixed-step "fast" and adaptive), sampling.py:304-507 -------------------------------
# Restated in the reference's eps formulation and operation order (t = -log sigma as 0-dim fp32 tensors).
# `noise_sampler(sigma, sigma_next)` is required whenever eta > 0 (the tests inject recorded noise).

def _dpm_sigma(t):
    return t.neg().exp()


def _dpm_eps(model, x, t, extra, cache, key, count):
    """:350-357: eps = (x - D(x, sigma(t))) / sigma(t), memoised per step under `key`."""
    if key not in cache:
        sigma = _dpm_sigma(t) * x.new_ones([x.shape[0]])
        cache[key] = (x - model(x, sigma, **extra)) / _dpm_sigma(t)
        count[0] += 1
    return cache[key]


def _dpm_step1(model, x, t, t_next, extra, cache, count):
    """:359-364."""
    h = t_next - t
    eps = _dpm_eps(model, x, t, extra, cache, "eps", count)
    return x - _dpm_sigma(t_next) * h.expm1() * eps


def _dpm_step2(model, x, t, t_next, extra, cache, count, r1=1 / 2):
    """:366-374."""
    h = t_next - t
    eps = _dpm_eps(model, x, t, extra, cache, "eps", count)
    s1 = t + r1 * h
    u1 = x - _dpm_sigma(s1) * (r1 * h).expm1() * eps
    eps_r1 = _dpm_eps(model, u1, s1, extra, cache, "eps_r1", count)
    return x - _dpm_sigma(t_next) * h.expm1() * eps - _dpm_sigma(t_next) / (2 * r1) * h.expm1() * (eps_r1 - eps)


def _dpm_step3(model, x, t, t_next, extra, cache, count, r1=1 / 3, r2=2 / 3):
    """:376-388."""
    h = t_next - t
    eps = _dpm_eps(model, x, t, extra, cache, "eps", count)
    s1, s2 = t + r1 * h, t + r2 * h
    u1 = x - _dpm_sigma(s1) * (r1 * h).expm1() * eps
    eps_r1 = _dpm_eps(model, u1, s1, extra, cache, "eps_r1", count)
    u2 = x - _dpm_sigma(s2) * (r2 * h).expm1() * eps - _dpm_sigma(s2) * (r2 / r1) * ((r2 * h).expm1() / (r2 * h) - 1) * (eps_r1 - eps)
    eps_r2 = _dpm_eps(model, u2, s2, extra, cache, "eps_r2", count)
    return x - _dpm_sigma(t_next) * h.expm1() * eps - _dpm_sigma(t_next) / r2 * (h.expm1() / h - 1) * (eps_r2 - eps)


def _dpm_ancestral(t, t_next, t_end, eta):
    """:411-416 / :448-453: the deterministic part stops at t(sigma_down); su is the noise added back."""
    if not eta:
        return t_next, 0.0
    sd, su = ancestral_step(_dpm_sigma(t), _dpm_sigma(t_next), eta)
    t_next_ = torch.minimum(t_end, -sd.log())
    su = (_dpm_sigma(t_next) ** 2 - _dpm_sigma(t_next_) ** 2) ** 0.5
    return t_next_, su


def sample_dpm_fast(model, x, sigma_min, sigma_max, n, extra_args=None, callback=None, eta=0.0, s_noise=1.0, noise_sampler=None):
    """:390-430, :482-491."""
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError("sigma_min and sigma_max must not be 0")
    extra = {} if extra_args is None else extra_args
    t_start, t_end = -torch.tensor(sigma_max).log(), -torch.tensor(sigma_min).log()
    if not t_end > t_start and eta:
        raise ValueError("eta must be 0 for reverse sampling")
    m = math.floor(n / 3) + 1
    ts = torch.linspace(t_start, t_end, m + 1)
    orders = [3] * (m - 2) + [2, 1] if n % 3 == 0 else [3] * (m - 1) + [n % 3]
    count = [0]
    for i, order in enumerate(orders):
        cache = {}
        t, t_next = ts[i], ts[i + 1]
        t_next_, su = _dpm_ancestral(t, t_next, t_end, eta)
        eps = _dpm_eps(model, x, t, extra, cache, "eps", count)
        if callback is not None:
            callback({"x": x, "i": i, "t": ts[i], "t_up": t, "denoised": x - _dpm_sigma(t) * eps, "sigma": _dpm_sigma(t), "sigma_hat": _dpm_sigma(t)})
        step = (_dpm_step1, _dpm_step2, _dpm_step3)[order - 1]
        x = step(model, x, t, t_next_, extra, cache, count)
        if eta:
            x = x + su * s_noise * noise_sampler(_dpm_sigma(t), _dpm_sigma(t_next))
    return x


class PIDStepSizeController:
    """:304-330."""

    def __init__(self, h, pcoeff, icoeff, dcoeff, order=1, accept_safety=0.81, eps=1e-8):
        self.h = h
        self.b1, self.b2, self.b3 = (pcoeff + icoeff + dcoeff) / order, -(pcoeff + 2 * dcoeff) / order, dcoeff / order
        self.accept_safety, self.eps, self.errs = accept_safety, eps, []

    def propose_step(self, error):
        inv_error = 1 / (float(error) + self.eps)
        if not self.errs:
            self.errs = [inv_error, inv_error, inv_error]
        self.errs[0] = inv_error
        factor = self.errs[0] ** self.b1 * self.errs[1] ** self.b2 * self.errs[2] ** self.b3
        factor = 1 + math.atan(factor - 1)
        accept = factor >= self.accept_safety
        if accept:
            self.errs[2], self.errs[1] = self.errs[1], self.errs[0]
        self.h *= factor
        return accept


def sample_dpm_adaptive(model, x, sigma_min, sigma_max, extra_args=None, callback=None, order=3, rtol=0.05, atol=0.0078, h_init=0.05,
                        pcoeff=0.0, icoeff=1.0, dcoeff=0.0, accept_safety=0.81, eta=0.0, s_noise=1.0, noise_sampler=None, return_info=False):
    """:432-480, :494-507."""
    if sigma_min <= 0 or sigma_max <= 0:
        raise ValueError("sigma_min and sigma_max must not be 0")
    if order not in {2, 3}:
        raise ValueError("order should be 2 or 3")
    extra = {} if extra_args is None else extra_args
    t_start, t_end = -torch.tensor(sigma_max).log(), -torch.tensor(sigma_min).log()
    forward = bool(t_end > t_start)
    if not forward and eta:
        raise ValueError("eta must be 0 for reverse sampling")
    h_init = abs(h_init) * (1 if forward else -1)
    atol_t, rtol_t = torch.tensor(atol), torch.tensor(rtol)
    s, x_prev = t_start, x
    pid = PIDStepSizeController(h_init, pcoeff, icoeff, dcoeff, 1.5 if eta else order, accept_safety)
    info = {"steps": 0, "nfe": 0, "n_accept": 0, "n_reject": 0}
    count = [0]
    while (s < t_end - 1e-5) if forward else (s > t_end + 1e-5):
        cache = {}
        t = torch.minimum(t_end, s + pid.h) if forward else torch.maximum(t_end, s + pid.h)
        t_, su = _dpm_ancestral(s, t, t_end, eta)
        eps = _dpm_eps(model, x, s, extra, cache, "eps", count)
        denoised = x - _dpm_sigma(s) * eps
        if order == 2:
            x_low = _dpm_step1(model, x, s, t_, extra, cache, count)
            x_high = _dpm_step2(model, x, s, t_, extra, cache, count)
        else:
            x_low = _dpm_step2(model, x, s, t_, extra, cache, count, r1=1 / 3)
            x_high = _dpm_step3(model, x, s, t_, extra, cache, count)
        delta = torch.maximum(atol_t, rtol_t * torch.maximum(x_low.abs(), x_prev.abs()))
        error = torch.linalg.norm((x_low - x_high) / delta) / x.numel() ** 0.5
        accept = pid.propose_step(error)
        if accept:
            x_prev = x_low
            x = x_high + su * s_noise * noise_sampler(_dpm_sigma(s), _dpm_sigma(t)) if eta else x_high
            s = t
            info["n_accept"] += 1
        else:
            info["n_reject"] += 1
        info["nfe"] += order
        info["steps"] += 1
        if callback is not None:
            callback({"x": x, "i": info["steps"] - 1, "t": s, "t_up": s, "denoised": denoised, "error": error, "h": pid.h,
                      "sigma": _dpm_sigma(s), "sigma_hat": _dpm_sigma(s), **info})
    return (x, info) if return_info else x

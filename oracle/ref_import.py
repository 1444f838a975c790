"""Load the real reference (read-only, /root/reference) with its absent third-party modules stubbed.

Test infrastructure: only usable in the build container (the GPU box has no /root/reference).
The stubs give *no* behaviour except ``jsonmerge.merge`` (recursive dict merge, head wins, lists
replaced), which ``k_diffusion/config.py:6,116-146`` needs, and an optional ``natten`` whose
``na2d`` is the oracle's restated op so that the reference's own NeighborhoodSelfAttentionBlock
(image_transformer_v2.py:415-443) can run.
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "k_diffusion"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _merge(base, head):
    if isinstance(base, dict) and isinstance(head, dict):
        out = dict(base)
        for k, v in head.items():
            out[k] = _merge(base[k], v) if k in base else v
        return out
    return head


def load(with_natten=True):
    """Returns the reference ``k_diffusion`` package (imported once per process)."""
    if "k_diffusion" in sys.modules:
        return sys.modules["k_diffusion"]
    if not available():
        raise RuntimeError("reference not present at " + REFERENCE_ROOT)
    os.environ["K_DIFFUSION_USE_COMPILE"] = "0"  # keep Inductor/Triton out of the oracle
    for name in ("skimage", "dctorch", "cleanfid", "clip", "torchvision", "torchsde", "torchdiffeq"):
        assert name not in sys.modules or getattr(sys.modules[name], "__file__", None) is None
    _stub("skimage", transform=_stub("skimage.transform"))
    _stub("dctorch", functional=_stub("dctorch.functional"))
    _stub("jsonmerge", merge=_merge)
    _stub("cleanfid")
    _stub("cleanfid.inception_torchscript", InceptionV3W=object)
    _stub("clip", available_models=lambda: [])
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", functional=_stub("torchvision.transforms.functional"))
    tv.datasets = _stub("torchvision.datasets")
    _stub("torchsde", BrownianTree=None)   # default SDE noise sampler unusable: pass noise_sampler=
    _stub("torchdiffeq", odeint=None)      # only log_likelihood needs it
    if with_natten:
        from . import hdit as _hdit

        def na2d(q, k, v, kernel_size, scale=None, **kw):
            return _hdit.na2d(q, k, v, kernel_size, 1.0 if scale is None else scale)
        _stub("natten", functional=_stub("natten.functional", na2d=na2d), has_fused_na=lambda: True)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import k_diffusion  # noqa
    finally:
        sys.path.remove(REFERENCE_ROOT)
    return sys.modules["k_diffusion"]

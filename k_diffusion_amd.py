"""Import alias: ``import k_diffusion_amd as K`` -> the ``k-diffusion_amd/`` package.

The package directory carries the repository's hyphenated name, which the ``import`` statement
cannot spell; this one-line shim (and ``importlib.import_module("k-diffusion_amd")``) can.
"""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("k-diffusion_amd")

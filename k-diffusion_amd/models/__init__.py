"""Denoiser networks.  Only the hourglass transformer (image_transformer_v2) is on the MI355X hot
path; the reference's U-Net (image_v1) and transformer v1 families are out of scope."""
from . import axial_rope, flops, image_transformer_v2  # noqa: F401
from .image_transformer_v2 import ImageTransformerDenoiserModelV2  # noqa: F401

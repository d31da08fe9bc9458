"""Drop-in for the reference's ``libs/gan/StyleGAN2/op`` package (op/__init__.py:1-2)."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d

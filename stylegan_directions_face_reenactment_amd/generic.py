"""Latent-shift glue around the generator: counterpart of the reference's
``libs/utilities/generic.py:116-151`` (``get_shifted_latent_code`` / ``generate_image``), same names,
argument order and return values, so ``run_inference.py:180`` / ``libs/trainer.py:160-177`` call sites work
unchanged.  Shift add, W->W+ broadcast and truncation are one HIP launch (sgdfr_latent_prepare_f32)."""
import os

import torch

from . import autograd as AG
from . import functional as F_


def _prepare(w, n_latent, shift, shift_layers):
    if torch.is_grad_enabled() and (w.requires_grad or shift.requires_grad):
        return AG.LatentPrepareFn.apply(w, shift, None, n_latent, shift_layers, 1.0)
    return F_.latent_prepare(w, n_latent, shift=shift, shift_layers=shift_layers)


def get_shifted_latent_code(G, z, shift, input_is_latent=False, truncation=1, truncation_latent=None,
                            w_plus=False, num_layers=None):
    """[B, n_latent, 512] latent with `shift` added (generic.py:116-135).  As in the reference the
    `truncation*` arguments are accepted and ignored here: truncation happens inside G afterwards."""
    w = z if input_is_latent else G.get_latent(z)
    if w_plus:                       # shift [B, L, 512] added to the first L rows (:133)
        return _prepare(w, G.n_latent, shift, shift.shape[1])
    layers = G.n_latent if num_layers is None else num_layers   # shift [B, 512] (:123-130)
    return _prepare(w, G.n_latent, shift, layers)


# generate_image hands back verified frames: a no-grad forward is awaited and, had any fp16 operand left the generator's range
# plan, re-rendered in bf16x3 first (Generator.forward(verify_range=True); fp32 -- the reference's arithmetic -- never clamps,
# model.py:232-273).  SGDFR_VERIFY_RANGE=0 / generic.VERIFY_RANGE = False returns at once and leaves the check to the
# generator's non-blocking poll (callers that keep the device queue full).
VERIFY_RANGE = os.environ.get('SGDFR_VERIFY_RANGE', '1') != '0'


def generate_image(G, latent_code, truncation, trunc, w_plus=True, num_layers_shift=8, shift_code=None,
                   input_is_latent=False, return_latents=False):
    """generic.py:137-151."""
    extra = {'verify_range': bool(VERIFY_RANGE)} if hasattr(G, 'range_ok') else {}
    if shift_code is None:
        imgs = G([latent_code], return_latents=return_latents, truncation=truncation, truncation_latent=trunc,
                 input_is_latent=input_is_latent, **extra)
    else:
        shifted = get_shifted_latent_code(G, latent_code, shift_code, input_is_latent=input_is_latent,
                                          truncation=truncation, truncation_latent=trunc, w_plus=w_plus,
                                          num_layers=num_layers_shift)
        imgs = G([shifted], return_latents=return_latents, truncation=truncation, truncation_latent=trunc,
                 input_is_latent=True, **extra)
    image, latent_w = imgs[0], imgs[1]
    if image.shape[2] > 256:   # only for the 1024 generators (generic.py:146-148); stock pooling
        image = torch.nn.functional.adaptive_avg_pool2d(image, (256, 256))
    return (image, latent_w) if return_latents else image

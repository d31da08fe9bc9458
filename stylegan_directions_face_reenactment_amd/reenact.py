"""Batched reenactment driver: the MI355X counterpart of the per-frame loop of the reference's
``run_inference.py:170-181``.

The reference re-renders ONE target frame per generator call:
    shift = A(shift_vector [1,15]) ; generate_image(G, source_code, 0.7, trunc, w_plus, 8, shift_code=shift,
                                                     input_is_latent=True)
Every frame depends only on the fixed source code and its own shift vector, so N frames are exactly N rows of one
batch.  `ReenactmentSession.frames` chunks the shift vectors, broadcasts the source W+ code inside
``sgdfr_latent_prepare_f32`` (shift add + truncation in the same launch) and yields [b,3,H,W] images, optionally
converted on the GPU to the uint8 HWC layout the reference writes (libs/utilities/image_utils.py:97-110).
"""
import torch

from . import _native as N
from . import functional as F_


def images_to_uint8(images):
    """[B,3,H,W] fp32 in [-1,1] -> [B,H,W,3] uint8 with the reference's scaling (image_utils.py:87-110)."""
    N.require_device(images)
    x = N.f32c(images)
    B, C, H, W = x.shape
    if C != 3:
        raise RuntimeError('expected RGB images, got %d channels' % C)
    y = torch.empty(B, H, W, 3, device=x.device, dtype=torch.uint8)
    N.call('sgdfr_image_to_u8_f32', N.ptr(x), N.ptr(y), B, H, W, N.stream())
    return y


class ReenactmentSession:
    """One source identity, many target poses/expressions."""

    def __init__(self, G, A, source_code, truncation=0.7, trunc=None, batch=32):
        if source_code.ndim == 2:
            source_code = source_code.unsqueeze(0)
        if source_code.shape[0] != 1 or source_code.shape[1] != G.n_latent:
            raise RuntimeError('source_code must be one W+ code [1,%d,512]' % G.n_latent)
        if truncation < 1 and trunc is None:
            raise RuntimeError('truncation < 1 needs the truncation latent')
        self.G, self.A = G, A
        self.source = source_code.contiguous()
        self.truncation, self.trunc, self.batch = truncation, trunc, batch

    @torch.no_grad()
    def frames(self, shift_vectors, as_uint8=False):
        """shift_vectors [N, input_dim] -> generator of image batches (same result as N generate_image calls)."""
        n = shift_vectors.shape[0]
        for lo in range(0, n, self.batch):
            sv = shift_vectors[lo:lo + self.batch]
            shift = self.A(sv)                                              # [b, L, 512] (w_plus) or [b, 512]
            b = sv.shape[0]
            w = self.source.expand(b, -1, -1).contiguous()
            layers = shift.shape[1] if shift.ndim == 3 else self.A.num_layers
            latent = F_.latent_prepare(w, self.G.n_latent, shift=shift, shift_layers=layers)
            img, _ = self.G([latent], input_is_latent=True, truncation=self.truncation, truncation_latent=self.trunc)
            yield images_to_uint8(img) if as_uint8 else img

    def render(self, shift_vectors, as_uint8=False):
        return torch.cat(list(self.frames(shift_vectors, as_uint8=as_uint8)), 0)

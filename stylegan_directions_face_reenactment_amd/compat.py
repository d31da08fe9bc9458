"""Mount this package at the reference's import paths so its scripts run unchanged:

    import stylegan_directions_face_reenactment_amd.compat as compat; compat.install()
    from libs.gan.StyleGAN2.model import Generator          # -> the MI355X generator
    from libs.models.direction_matrix import DirectionMatrix

Only the modules of the hot path are aliased; everything else under ``libs`` keeps
resolving to the reference checkout on sys.path.  ``libs.utilities.generic`` is NOT replaced wholesale
(it also holds DECA glue); call ``patch_generic(module)`` to swap in the two fused functions.
"""
import importlib
import sys

ALIASES = {
    'libs.gan.StyleGAN2.model': 'stylegan_directions_face_reenactment_amd.model',
    'libs.gan.StyleGAN2.op': 'stylegan_directions_face_reenactment_amd.op',
    'libs.gan.StyleGAN2.op.fused_act': 'stylegan_directions_face_reenactment_amd.op.fused_act',
    'libs.gan.StyleGAN2.op.upfirdn2d': 'stylegan_directions_face_reenactment_amd.op.upfirdn2d',
    'libs.models.direction_matrix': 'stylegan_directions_face_reenactment_amd.direction_matrix',
}


def install():
    for alias, target in ALIASES.items():
        sys.modules[alias] = importlib.import_module(target)
    return sorted(ALIASES)


def patch_generic(generic_module):
    from . import generic
    generic_module.get_shifted_latent_code = generic.get_shifted_latent_code
    generic_module.generate_image = generic.generate_image
    return generic_module

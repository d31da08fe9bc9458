"""ctypes binding of csrc/libsgdfr_hip.so (C ABI declared in include/sgdfr.h).

The library handle lives at module level (never on an nn.Module) so modules stay
deepcopy-able and picklable (the reference deep-copies G in libs/optimization.py:28 and
torch.save()s A in libs/utilities/utils_train.py:594-603).

There is NO fallback: if the shared library is missing the first native call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libsgdfr_hip.so')
# SGDFR_LIB redirects the load to a probe build (scripts/build_probe.py, timing scripts only); it is honoured only together
# with SGDFR_ALLOW_LIB_OVERRIDE=1 and announced on stderr -- a stray variable must not swap the production library silently
if os.environ.get('SGDFR_LIB'):
    if os.environ.get('SGDFR_ALLOW_LIB_OVERRIDE') == '1':
        import sys as _sys
        LIB_PATH = os.environ['SGDFR_LIB']
        _sys.stderr.write('stylegan_directions_face_reenactment_amd: loading the native library from SGDFR_LIB=%s\n' % LIB_PATH)
    else:
        import warnings as _warnings
        _warnings.warn('SGDFR_LIB is set but ignored (set SGDFR_ALLOW_LIB_OVERRIDE=1 to load a probe build)', RuntimeWarning)
ABI_VERSION = 22

_c_f32p = ctypes.c_void_p
_i, _i64, _f = ctypes.c_int, ctypes.c_int64, ctypes.c_float

# name -> argtypes ; every function returns int (0 = ok)
SIGNATURES = {
    'sgdfr_fused_bias_act_f32': [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _i, _i, _i, _i, _f, _f, ctypes.c_void_p],
    'sgdfr_upfirdn2d_f32': [_c_f32p, _c_f32p, _c_f32p] + [_i] * 14 + [ctypes.c_void_p],
    'sgdfr_linear_f32': [_c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _i64, _i, _i, _i, _f, _f, _i, _f, _f,
                         ctypes.c_void_p],
    'sgdfr_pixelnorm_f32': [_c_f32p, _c_f32p, _i, _i, _f, ctypes.c_void_p],
    'sgdfr_pixelnorm_bwd_f32': [_c_f32p, _c_f32p, _c_f32p, _i, _i, _f, ctypes.c_void_p],
    'sgdfr_latent_prepare_f32': [_c_f32p, _i, _c_f32p, _i, _i, _c_f32p, _f, _c_f32p, _i, _i, _i, ctypes.c_void_p],
    'sgdfr_modconv_prepack_f32': [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _i, _i, _i, ctypes.c_void_p],
    'sgdfr_modconv_prepack_t_f32': [_c_f32p, _c_f32p, _i, _i, _i, _i, ctypes.c_void_p],
    'sgdfr_act_grad_reduce_f32': [_c_f32p, _c_f32p, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i, _i, _i, _f,
                                  _f, _i, ctypes.c_void_p, ctypes.c_void_p],
    'sgdfr_blur_adjoint_f32': [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i, _i, _i, _i, ctypes.c_void_p],
    'sgdfr_scale_reduce_f32': [_c_f32p, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _i, _i, _i, ctypes.c_void_p],
    'sgdfr_torgb_bwd_f32': [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i, _i, _i, _i, ctypes.c_void_p],
    'sgdfr_modconv_wgrad_f32': [_c_f32p, _c_f32p, _c_f32p, _i64, _c_f32p, _c_f32p, _i, _i, _i, _i, _i, _i, ctypes.c_void_p],
    'sgdfr_modconv_wgrad_finish_f32': [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _i, _i, ctypes.c_void_p],
    'sgdfr_modconv_wgrad_parts_f32': [_c_f32p, _c_f32p, _c_f32p, _i64, _c_f32p, _c_f32p, _i, _i, _i, _i, _i, _i, ctypes.c_void_p],
    'sgdfr_modconv_wgrad_finish_parts_f32': [_c_f32p, _i, _c_f32p, _c_f32p, _c_f32p, _i, _i, ctypes.c_void_p],
    'sgdfr_modconv_wgrad_finish_parts_oik_f32': [_c_f32p, _i, _c_f32p, _c_f32p, _c_f32p, _i64, _c_f32p, _c_f32p, _i, _c_f32p, _i, _i,
                                                 ctypes.c_void_p],
    'sgdfr_modconv_prepack_wino_f32': [_c_f32p, _c_f32p, _i, _i, _i, ctypes.c_void_p],
    'sgdfr_modconv2d_wino_supported': [_i, _i, _i, _i, _i],
    'sgdfr_modconv2d_wino_f32': [_c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p,
                                 _c_f32p, _i, _i, _i, _i, _i, _i, _f, _f, ctypes.c_void_p],
    'sgdfr_modconv_prepack_split_f32': [_c_f32p, ctypes.c_void_p, _i, _i, _i, _i, ctypes.c_void_p, ctypes.c_void_p],
    'sgdfr_modconv2d_split_supported': [_i, _i, _i, _i, _i, _i],
    'sgdfr_modconv2d_split_f32': [_c_f32p, _i64, ctypes.c_void_p, _c_f32p, _c_f32p, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p,
                                  _c_f32p, _c_f32p, _i, _c_f32p, _c_f32p, _c_f32p, _i, ctypes.c_void_p, _c_f32p, _i, _i, _i, _i,
                                  _i, _i, _i64, _i, _i, _f, _f, ctypes.c_void_p, ctypes.c_void_p],
    'sgdfr_to_split_f32': [_c_f32p, _c_f32p, ctypes.c_void_p, _i, _i, _i, _i, _i, ctypes.c_void_p, ctypes.c_void_p],
    'sgdfr_planes_to_split_f32': [_c_f32p, _c_f32p, ctypes.c_void_p, _i, _i, _i, _i, _i, ctypes.c_void_p, ctypes.c_void_p],
    'sgdfr_blur_adjoint_split_f32': [_c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p, _c_f32p, _i, _i, _i, _i, _i, ctypes.c_void_p,
                                     ctypes.c_void_p],
    'sgdfr_modconv2d_split_cout_tiles': [_i, _i, _i, _i, _i, _i],
    'sgdfr_modconv2d_split_cout_tiles_xin': [_i, _i, _i, _i, _i, _i],
    'sgdfr_modconv2d_split_xin_supported': [_i, _i, _i, _i, _i, _i],
    'sgdfr_modconv2d_split_ksplit_hint': [_i, _i, _i, _i, _i, _i],
    'sgdfr_torgb_finish_f32': [_c_f32p, _i, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i, _i, _i, ctypes.c_void_p],
    'sgdfr_torgb_finish_u8_f32': [_c_f32p, _i, _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p, _i64, _i, _i, _i, _i, _i, ctypes.c_void_p],
    'sgdfr_image_to_u8_f32': [_c_f32p, ctypes.c_void_p, _i, _i, _i, ctypes.c_void_p],
    'sgdfr_grid_to_u8_f32': [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64), _i, ctypes.c_void_p, _i, _i, _i,
                             _i, ctypes.c_void_p],
    'sgdfr_modconv2d_splitk_hint': [_i, _i, _i, _i, _i, _i],
    'sgdfr_modconv2d_splitk_f32': [_c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p,
                                   _c_f32p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, ctypes.c_void_p],
    'sgdfr_demod_grad_f32': [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i, _i, _i, ctypes.c_void_p],
    'sgdfr_style_demod_f32': [_c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i, _i, _i, _i,
                              ctypes.c_void_p],
    'sgdfr_modconv2d_fwd_f32': [_c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p,
                                _i, _i, _i, _i, _i, _i, _i, _f, _f, ctypes.c_void_p],
    'sgdfr_blur_bias_act_f32': [_c_f32p, _c_f32p, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _i, _i, _i, _i, _i, _f,
                                _f, ctypes.c_void_p, ctypes.c_void_p],
    'sgdfr_blur_bias_act_split_f32': [_c_f32p, _c_f32p, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p, _i, _i, _i, _i, _i64,
                                      _i, _i, _i, _f, _f, ctypes.c_void_p, ctypes.c_void_p],
    'sgdfr_torgb_fwd_f32': [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, _i, _i, _i, _i,
                            ctypes.c_void_p],
}



class Direction(ctypes.Structure):
    """struct sgdfr_direction (include/sgdfr.h)."""
    _fields_ = [('kind', ctypes.c_int), ('col', ctypes.c_int), ('a', ctypes.c_double), ('b', ctypes.c_double)]


DIR_ZERO, DIR_ANGLE, DIR_JAW, DIR_EXP = 0, 1, 2, 3
MAX_DIRECTIONS = 64
SIGNATURES['sgdfr_make_shift_f32'] = [_c_f32p, _i64, _c_f32p, _i64, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _i, _i,
                                      ctypes.POINTER(Direction), _i, _c_f32p, _i, _i, ctypes.c_void_p]
SIGNATURES['sgdfr_make_shift_random_f32'] = [_c_f32p, _c_f32p, _c_f32p, _i, _i, ctypes.c_void_p, _c_f32p, _f,
                                             ctypes.POINTER(Direction), _i, _c_f32p, _i, ctypes.c_void_p]


class StyleLayer(ctypes.Structure):
    """struct sgdfr_style_layer (include/sgdfr.h)."""
    _fields_ = [('mod_w', ctypes.c_void_p), ('mod_b', ctypes.c_void_p), ('q', ctypes.c_void_p),
                ('s', ctypes.c_void_p), ('d', ctypes.c_void_p), ('cin', ctypes.c_int), ('cout', ctypes.c_int),
                ('latent_index', ctypes.c_int), ('s_n', ctypes.c_void_p), ('d_n', ctypes.c_void_p),
                ('x_absmax', ctypes.c_void_p), ('x_log2', ctypes.c_int), ('headroom', ctypes.c_int)]


SIGNATURES['sgdfr_split_range_f32'] = [_c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p, _i, _i, _i, _i, _i, _i, ctypes.c_void_p]
SIGNATURES['sgdfr_absmax_f32'] = [_c_f32p, _i64, _i64, _i, ctypes.c_void_p, _i, ctypes.c_void_p]
SIGNATURES['sgdfr_styles_batched_f32'] = [_c_f32p, _i, _i, _i, ctypes.POINTER(StyleLayer), _i, ctypes.c_void_p]
MAX_STYLE_LAYERS = 40


class StyleGradLayer(ctypes.Structure):
    """struct sgdfr_style_grad_layer (include/sgdfr.h)."""
    _fields_ = [('gs', ctypes.c_void_p), ('rgb_r', ctypes.c_void_p), ('rgb_w', ctypes.c_void_p), ('a', ctypes.c_void_p),
                ('d', ctypes.c_void_p), ('s', ctypes.c_void_p), ('qt', ctypes.c_void_p), ('mod_w', ctypes.c_void_p),
                ('ds', ctypes.c_void_p), ('gmod_w', ctypes.c_void_p), ('gmod_b', ctypes.c_void_p), ('a_stride', ctypes.c_longlong),
                ('cin', ctypes.c_int), ('cout', ctypes.c_int), ('latent_index', ctypes.c_int)]


class ParamGrad(ctypes.Structure):
    """struct sgdfr_param_grad (include/sgdfr.h)."""
    _fields_ = [('inp', ctypes.c_void_p), ('aux', ctypes.c_void_p), ('out', ctypes.c_void_p), ('kind', ctypes.c_int), ('C', ctypes.c_int),
                ('HW', ctypes.c_int), ('scale', ctypes.c_float)]


class AdamTensor(ctypes.Structure):
    """struct sgdfr_adam_tensor (include/sgdfr.h)."""
    _fields_ = [('p', ctypes.c_void_p), ('g', ctypes.c_void_p), ('m', ctypes.c_void_p), ('v', ctypes.c_void_p), ('n', ctypes.c_int64)]


MAX_ADAM_TENSORS = 96
SIGNATURES['sgdfr_adam_f32'] = [ctypes.POINTER(AdamTensor), _i, _c_f32p, _f, _f, _f, _f, ctypes.c_void_p]
PGRAD_BIAS, PGRAD_NOISE, PGRAD_RGB_W, PGRAD_RGB_B = 0, 1, 2, 3
MAX_PARAM_GRADS = 64
SIGNATURES['sgdfr_styles_batched_bwd_f32'] = [ctypes.POINTER(StyleGradLayer), _i, _c_f32p, _c_f32p, _i, _i, _i, ctypes.c_void_p]
SIGNATURES['sgdfr_demod_dq_f32'] = [_c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _i, _i, _i, ctypes.c_void_p]
SIGNATURES['sgdfr_param_grads_f32'] = [ctypes.POINTER(ParamGrad), _i, _i, ctypes.c_void_p]
SIGNATURES['sgdfr_grad_join_f32'] = ([_c_f32p] * 7 + [_c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p,
                                                                  _c_f32p, _c_f32p, _i, _i, _i, _f, _f, _i, _i, ctypes.c_void_p])
SIGNATURES['sgdfr_modconv2d_wsplit_supported'] = [_i, _i, _i, _i, _i, _i]
SIGNATURES['sgdfr_modconv2d_wsplit_wide'] = [_i, _i, _i, _i, _i]
SIGNATURES['sgdfr_modconv2d_split_f8_ok'] = [_i, _i, _i, _i, _i, _i]
SPLIT_HANDOVER_F8 = 0x100
SIGNATURES['sgdfr_modconv_prepack_wsplit_f32'] = [_c_f32p, ctypes.c_void_p, _i, _i, _i, _i, ctypes.c_void_p, ctypes.c_void_p]
SIGNATURES['sgdfr_to_wsplit_f32'] = [_c_f32p, _c_f32p, ctypes.c_void_p, _i, _i, _i, _i, _i, _i, ctypes.c_void_p, ctypes.c_void_p]
SIGNATURES['sgdfr_modconv2d_wsplit_f32'] = [ctypes.c_void_p, ctypes.c_void_p, _c_f32p, _c_f32p, _i64, _c_f32p, _c_f32p, _c_f32p, _c_f32p,
                                            _c_f32p, _c_f32p, _c_f32p, ctypes.c_void_p, _c_f32p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f,
                                            ctypes.c_void_p, ctypes.c_void_p]
SIGNATURES['sgdfr_fused_bias_act'] = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, _i64, _i, _i, _i, _i, _f, _f, _i,
                                      ctypes.c_void_p]
SIGNATURES['sgdfr_upfirdn2d'] = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [_i] * 15 + [ctypes.c_void_p]
DTYPES = {torch.float32: 0, torch.float16: 1, torch.float64: 2}      # SGDFR_DTYPE_* of the two reference natives
# measurement-only symbols: bound when present, never required of a production library (bench.py's measured_mfma_ceiling)
OPTIONAL_SIGNATURES = {'sgdfr_mfma_ceiling_probe': [_i, _i, _i, _i, _i, _c_f32p, ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]}

MODE_PLAIN3, MODE_UP3, MODE_DOWN3 = 0, 1, 2
SPLIT_BF16, SPLIT_FP16, SPLIT_FP16F8 = 0, 1, 2      # include/sgdfr.h SGDFR_SPLIT_*
ACT_NONE, ACT_LRELU = 0, 1

_lib = None


def load():
    """Load (once) and return the ctypes library; raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'native library %s is missing: run `python -m stylegan_directions_face_reenactment_amd.build_native` '
            '(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for this path.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.sgdfr_abi_version.restype = ctypes.c_int
    lib.sgdfr_last_error.restype = ctypes.c_char_p
    lib.sgdfr_split_saturation_count.argtypes = [ctypes.c_int]
    lib.sgdfr_split_saturation_count.restype = ctypes.c_longlong
    lib.sgdfr_modconv_prepack_split_elems.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.sgdfr_modconv_prepack_split_elems.restype = ctypes.c_int64
    lib.sgdfr_modconv_prepack_wsplit_elems.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.sgdfr_modconv_prepack_wsplit_elems.restype = ctypes.c_int64
    if lib.sgdfr_abi_version() != ABI_VERSION:
        raise RuntimeError('libsgdfr_hip.so ABI %d != expected %d: rebuild' % (lib.sgdfr_abi_version(), ABI_VERSION))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    for name, argtypes in OPTIONAL_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (name, rc, lib.sgdfr_last_error().decode()))


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """hipStream_t of torch's current stream on the current device (as void*).  torch.cuda.current_stream() builds a Python
    Stream object per call (~8 us, once per launch: a tenth of a small-batch forward's host time); the raw getter does not."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def require_device(*tensors, dtype=torch.float32):
    """The reference's natives raise RuntimeError for non-CUDA tensors (CHECK_CUDA in
    op/fused_bias_act.cpp, op/upfirdn2d.cpp); same contract here -- and no silent CPU path."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('expected a GPU (HIP) tensor, got device %s: this package has no CPU path' % t.device)
        if t.dtype != dtype:
            raise RuntimeError('expected %s, got %s' % (str(dtype).replace('torch.', ''), t.dtype))


def native_dtype(t):
    """SGDFR_DTYPE_* of a tensor for the two natives the reference dispatches over float / double / half
    (AT_DISPATCH_FLOATING_TYPES_AND_HALF, fused_bias_act_kernel.cu:79, upfirdn2d_kernel.cu:225)."""
    if t.dtype not in DTYPES:
        raise RuntimeError('expected float32, float16 or float64, got %s' % t.dtype)
    return DTYPES[t.dtype]


def f32c(t):
    """contiguous float32 view/copy (the reference natives force .contiguous() too)."""
    return t if t.is_contiguous() else t.contiguous()

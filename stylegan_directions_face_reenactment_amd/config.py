"""The configuration of the path: ONE frozen object instead of module-level switches (functional.py re-exports these names).
"""
import dataclasses
import os
import threading

# Every switch of the path lives in ONE frozen object.  `config()` is the configuration in force for the calling thread: the
# innermost `with using(cfg):` block, else the process default DEFAULT (seeded from SGDFR_* environment variables once, at import).
# A Generator holds its own (`G.config`, None = follow the ambient one) and runs its forward -- and, through the autograd
# Functions, its backward -- under it, so two generators with different arithmetics interleave in one process; launch plans, range
# plans and hipGraph captures are keyed on the object itself (hashable), not on an enumeration of switches.
@dataclasses.dataclass(frozen=True)
class Config:
    # Arithmetic of the 3x3 modulated convs (inference path, autograd forward, dL/dx of the plain convs): 'fp16x3' | 'fp32' | 'bf16x3'
    # (see the PRECISION comment below)
    precision: str = 'fp16x3'
    # Range plan of the fp16-split conv: True (calibrated per weight version) | 'exact' (measured per layer and image) | False
    range_plan: object = True
    backward_arith: str = 'fp16x3'      # dL/dx convs of the split kernels: 'bf16x3' | 'fp16x3' (ranged per image, autograd.py)
    use_plane_padding: bool = True      # inference chain: parity planes of the transposed conv padded to whole lines + interleaved
    use_split_chain: bool = True        # activations between split convs only in split form
    use_rgb_fusion: bool = True         # ToRGB partial sums in the epilogue of the split conv that feeds it (no-grad path)
    use_splitk: bool = True             # K-sliced launches for convs that cannot fill the chip (small batch / 4x4, 8x8 layers)
    use_winograd: bool = True           # fp32 path, plain 3x3 layers: Winograd F(2x2,3x3) MFMA kernel when the shape allows it
    winograd_min_blocks: int = 256      # below this many (64 cout x 64 tile) blocks the direct kernel's smaller tiles win
    use_wsplit: bool = True             # inference chain, plain layers fed by a transposed conv + blur: 1-D Winograd form (wsplit.hip)
    wsplit_f: int = 4                   # outputs per Winograd tile the chain prefers: 2 = F(2,3), 4 = F(4,3)
    wsplit_min_cin: int = 128           # ... for layers with at least this many input channels (0 = never)
    # Cross terms (hi*lo, lo*hi) of the fp16x3 product on the F(4,3) layers that take the wide-tile kernel: 'fp16' (three fp16
    # products) | 'fp8' (both cross terms in one e4m3 MFMA: 2 MFMA units per product instead of 3, ~5e-5 of max|y| per such layer
    # instead of 4e-6; include/sgdfr.h SGDFR_SPLIT_FP16F8)
    cross_terms: str = 'fp16'

    def replace(self, **changes):
        return dataclasses.replace(self, **changes)

    @classmethod
    def from_env(cls, env=None):
        e = os.environ if env is None else env
        rp = e.get('SGDFR_RANGE_PLAN', '1')
        return cls(precision=e.get('SGDFR_PRECISION', 'fp16x3'),
                   range_plan=False if rp == '0' else ('exact' if rp == 'exact' else True),
                   backward_arith=e.get('SGDFR_BWD_ARITH', 'fp16x3'),
                   use_plane_padding=e.get('SGDFR_PLANE_PADDING', '1') != '0',
                   use_split_chain=e.get('SGDFR_SPLIT_CHAIN', '1') != '0',
                   use_wsplit=e.get('SGDFR_WSPLIT', '1') != '0', wsplit_f=int(e.get('SGDFR_WSPLIT_F', '4')),
                   wsplit_min_cin=int(e.get('SGDFR_WSPLIT_MIN_CIN', '128')), cross_terms=e.get('SGDFR_CROSS_TERMS', 'fp16'))

    def __post_init__(self):
        if self.precision not in ('fp32', 'fp16x3', 'bf16x3'):
            raise ValueError("precision must be 'fp32', 'fp16x3' or 'bf16x3', got %r" % (self.precision,))
        if self.backward_arith not in ('fp16x3', 'bf16x3'):
            raise ValueError("backward_arith must be 'fp16x3' or 'bf16x3', got %r" % (self.backward_arith,))
        if self.cross_terms not in ('fp16', 'fp8'):
            raise ValueError("cross_terms must be 'fp16' or 'fp8', got %r" % (self.cross_terms,))
        if self.cross_terms == 'fp8' and self.precision == 'fp16x3' and self.range_plan is not True:
            raise ValueError("cross_terms='fp8' needs the calibrated range plan (range_plan=True): its fixed fp8 exponents assume "
                             "range-shifted operands")
        if self.wsplit_f not in (2, 4):
            raise ValueError('wsplit_f must be 2 or 4')


DEFAULT = Config.from_env()
_ambient = threading.local()


def config():
    """The Config in force for this thread (innermost `using` block, else DEFAULT)."""
    return getattr(_ambient, 'cfg', None) or DEFAULT


def set_default(cfg):
    """Replace the process default (what bench.py --precision does); `using` blocks and generator-held configs are unaffected."""
    global DEFAULT
    if not isinstance(cfg, Config):
        raise TypeError('set_default takes a functional.Config')
    DEFAULT = cfg


class using:
    """`with functional.using(cfg):` -- cfg is the configuration of every launch issued by this thread inside the block."""

    def __init__(self, cfg):
        self.cfg, self.prev = cfg, None

    def __enter__(self):
        self.prev = getattr(_ambient, 'cfg', None)
        _ambient.cfg = self.cfg
        return self.cfg

    def __exit__(self, *exc):
        _ambient.cfg = self.prev


# The old module-level switches (functional.PRECISION, USE_WSPLIT, ...) survive as READ-ONLY views of the ambient config, served
# by functional's module class: reading functional.PRECISION gives config().precision, assigning to it raises (an assignment would
# otherwise create a real module attribute that shadows the view while every launch ignores it).
LEGACY = {n: n.lower() for n in ('PRECISION', 'RANGE_PLAN', 'USE_PLANE_PADDING', 'USE_SPLIT_CHAIN', 'USE_RGB_FUSION', 'BACKWARD_ARITH',
                                 'USE_SPLITK', 'USE_WINOGRAD', 'USE_WSPLIT', 'WSPLIT_F', 'WSPLIT_MIN_CIN', 'WINOGRAD_MIN_BLOCKS')}


def set_precision(mode):
    """Process default arithmetic of the 3x3 convs (bench.py --precision): replaces the DEFAULT config's `precision`."""
    if mode not in ('fp32', 'fp16x3', 'bf16x3'):
        raise ValueError("precision must be 'fp32', 'fp16x3' or 'bf16x3', got %r" % (mode,))
    set_default(DEFAULT.replace(precision=mode))


class precision(using):
    """`with functional.precision('fp32'):` -- the arithmetic of the 3x3 convs inside the block (the ambient config otherwise)."""

    def __init__(self, mode):
        if mode not in ('fp32', 'fp16x3', 'bf16x3'):
            raise ValueError("precision must be 'fp32', 'fp16x3' or 'bf16x3', got %r" % (mode,))
        self.mode = mode
        super().__init__(None)

    def __enter__(self):
        self.cfg = config().replace(precision=self.mode)
        return super().__enter__()

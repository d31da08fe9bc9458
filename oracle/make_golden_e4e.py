"""ORACLE tooling -- tests/golden/kat6_e4e.npz from the REAL reference e4e encoder, in the build container.

Run:  python oracle/make_golden_e4e.py         (needs /root/reference; CPU only, ~1 min)

Imports `Encoder4Editing(50, 'ir_se', 256)` (libs/gan/encoder4editing/psp_encoders.py:122) through the same shims as
oracle/make_golden.py, loads the build's synthetic encoder state, encodes two synthetic 256x256 inputs, asserts that
oracle/e4e_oracle.py agrees on the full tensor and writes outputs + feature probes.  Inputs and the 190 MB state are
regenerated from the seed by the tests; the fixture holds numbers only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True

from oracle import make_golden as MG                                   # noqa: E402
from oracle import e4e_oracle as E                                     # noqa: E402
from stylegan_directions_face_reenactment_amd import synthetic as S   # noqa: E402

SEED = MG.SEED


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    MG.import_reference()
    from libs.gan.encoder4editing.psp_encoders import Encoder4Editing
    enc = Encoder4Editing(50, 'ir_se', 256).eval()
    keys = list(enc.state_dict().keys())
    P = S.synthetic_encoder_state(enc.state_dict(), seed=SEED)
    missing = enc.load_state_dict(P, strict=True)
    x = S.counter_tensor(SEED, 'e4e.x', (2, 3, 256, 256), 0.0, 0.5).clamp_(-1, 1)
    with torch.no_grad():
        ref = enc(x)
        ours = E.encoder_forward(P, x)
    print('reference W+ %s  |max| %.4f  mean|.| %.4f' % (tuple(ref.shape), float(ref.abs().max()), float(ref.abs().mean())))
    MG.check('e4e encoder [2,14,512]', ours, ref, 2e-5, rel=True)
    # a small-resolution case exercises odd FPN sizes (64 -> taps 16/8/4) and a shorter style list
    enc64 = Encoder4Editing(50, 'ir_se', 64).eval()
    P64 = S.synthetic_encoder_state(enc64.state_dict(), seed=SEED + 1)
    enc64.load_state_dict(P64, strict=True)
    x64 = S.counter_tensor(SEED, 'e4e.x64', (3, 3, 64, 64), 0.0, 0.5).clamp_(-1, 1)
    with torch.no_grad():
        ref64 = enc64(x64)
        ours64 = E.encoder_forward(P64, x64)
    MG.check('e4e encoder res 64 [3,10,512]', ours64, ref64, 2e-5, rel=True)
    np.savez_compressed(os.path.join(MG.OUT, 'kat6_e4e.npz'), seed=SEED, n_keys=len(keys),
                        w256=MG.npy(ref), w64=MG.npy(ref64), key_crc=np.array([hash_keys(keys)], dtype=np.int64))
    print('wrote kat6_e4e.npz (%d state keys)' % len(keys))


def hash_keys(keys):
    import zlib
    return zlib.crc32('\n'.join(keys).encode())


if __name__ == '__main__':
    main()

"""Builds the two shelved kernels (experiments/csrc/upfir.hip, uppp.hip) into experiments/libsgdfr_experiments.so for gfx950.

    python experiments/build.py [--force]

The library links against the product library (set_error / check_launch / the split kernel template's helpers live there); it is
NOT part of libsgdfr_hip.so, of include/sgdfr.h or of __graft_entry__.build().
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from stylegan_directions_face_reenactment_amd import build_native as B      # noqa: E402

LIB = os.path.join(HERE, 'libsgdfr_experiments.so')


def build(force=False):
    product = B.build()
    srcs = [os.path.join(HERE, 'csrc', f) for f in ('upfir.hip', 'uppp.hip')]
    objs = []
    for src in srcs:
        obj = src[:-4] + '.o'
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), os.path.getmtime(product)):
            subprocess.run([B._hipcc()] + B.FLAGS + ['-I', B.CSRC, '-I', HERE, '-c', src, '-o', obj], check=True)
        objs.append(obj)
    subprocess.run([B._hipcc(), '--offload-arch=' + B.ARCH, '-shared', '-fPIC', '-o', LIB] + objs +
                   ['-L', B.CSRC, '-lsgdfr_hip', '-Wl,-rpath,' + B.CSRC], check=True)
    return LIB


if __name__ == '__main__':
    print(build('--force' in sys.argv))

"""Probe build of the library for the split-Winograd kernels: build/libsgdfr_hip_wprobe.so = the same sources with wsplit.hip and
wswide.hip compiled -DSGDFR_WSPLIT_PROBE (the SGDFR_WSPLIT_DBG ablation switches: 2 = no epilogue, 16 = no operand DMA, 32 = never
wait for a DMA, 64 = no MFMAs; wrong results).  Cross-compiles without a GPU; build/ is git-ignored but travels with the gpurun
snapshot.  Select it with SGDFR_LIB=build/libsgdfr_hip_wprobe.so SGDFR_ALLOW_LIB_OVERRIDE=1 (scripts/wsplit_env_ab.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stylegan_directions_face_reenactment_amd import build_native as b


def build():
    b.build()
    out = os.path.join(ROOT, 'build')
    os.makedirs(out, exist_ok=True)
    objs = []
    for src in b.sources():
        name = os.path.basename(src)[:-4]
        if name in ('wsplit', 'wswide'):
            obj = os.path.join(out, name + '_probe.o')
            subprocess.run([b._hipcc()] + b.FLAGS + ['-DSGDFR_WSPLIT_PROBE', '-c', src, '-o', obj], check=True)
        else:
            obj = src[:-4] + '.o'
        objs.append(obj)
    lib = os.path.join(out, 'libsgdfr_hip_wprobe.so')
    subprocess.run([b._hipcc(), '--offload-arch=' + b.ARCH, '-shared', '-fPIC', '-o', lib] + objs, check=True)
    return lib


if __name__ == '__main__':
    print(build())

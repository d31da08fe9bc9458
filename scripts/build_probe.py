"""Probe build of the library, next to the real one: csrc/libsgdfr_hip_probe.so = the same sources with split.hip compiled
-DSGDFR_SPLIT_PROBE (the SGDFR_SPLIT_DBG ablation switches and the s_memtime trace).  Cross-compiles without a GPU, so it is
built in the container and travels with the snapshot; select it with SGDFR_LIB=<path> (scripts/tile_probe.py, tile_trace.py)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import build_native as b

def build(extra=()):
    b.build()
    src = os.path.join(b.CSRC, 'split.hip')
    obj = os.path.join(b.CSRC, 'split_probe.o')
    subprocess.run([b._hipcc()] + b.FLAGS + ['-DSGDFR_SPLIT_PROBE'] + list(extra) + ['-c', src, '-o', obj], check=True)
    objs = [s[:-4] + '.o' for s in b.sources() if not s.endswith('split.hip')] + [obj]
    lib = os.path.join(b.CSRC, 'libsgdfr_hip_probe.so')
    subprocess.run([b._hipcc(), '--offload-arch=' + b.ARCH, '-shared', '-fPIC', '-o', lib] + objs, check=True)
    return lib

if __name__ == '__main__':
    print(build(sys.argv[1:]))

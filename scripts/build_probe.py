"""Probe build of the library: build/libsgdfr_hip_probe.so = the same sources with split.hip compiled
-DSGDFR_SPLIT_PROBE (the SGDFR_SPLIT_DBG ablation switches and the s_memtime trace).  Cross-compiles without a GPU, so it is
built in the container and travels with the snapshot (build/ is git-ignored, not gpurun-ignored); select it with
SGDFR_LIB=build/libsgdfr_hip_probe.so SGDFR_ALLOW_LIB_OVERRIDE=1 (scripts/tile_probe.py, tile_trace.py)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import build_native as b

def build(extra=()):
    b.build()
    src = os.path.join(b.CSRC, 'split.hip')
    out = os.path.join(os.path.dirname(b.CSRC.rstrip('/')), '..', 'build')
    out = os.path.normpath(out)
    os.makedirs(out, exist_ok=True)
    obj = os.path.join(out, 'split_probe.o')
    subprocess.run([b._hipcc()] + b.FLAGS + ['-DSGDFR_SPLIT_PROBE'] + list(extra) + ['-c', src, '-o', obj], check=True)
    objs = [s[:-4] + '.o' for s in b.sources() if os.path.basename(s) != 'split.hip'] + [obj]
    lib = os.path.join(out, 'libsgdfr_hip_probe.so')
    subprocess.run([b._hipcc(), '--offload-arch=' + b.ARCH, '-shared', '-fPIC', '-o', lib] + objs, check=True)
    return lib

if __name__ == '__main__':
    print(build(sys.argv[1:]))

"""Builds csrc/*.hip into csrc/libsgdfr_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m stylegan_directions_face_reenactment_amd.build_native [--force] [--verbose]

The .so stays in-tree (git-ignored, but it travels to the GPU box with the snapshot).
"""
import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libsgdfr_hip.so')
ARCH = 'gfx950'
# -fno-slp-vectorize: packed f32 VALU (v_pk_add_f32 / v_pk_mul_f32) beside MFMAs costs more issue time than the
# scalar pair it replaces (cdna_hip_programming.md, "price of one filler beside MFMAs")
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-ffp-contract=on', '-fno-slp-vectorize', '-Wall',
         '-Wno-unused-function']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC)')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force, verbose, extra):
    obj = src[:-4] + '.o'
    deps = [src] + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '..', '..', 'include', '*.h'))
    if force or _stale(obj, deps):
        cmd = [_hipcc()] + FLAGS + extra + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
        if verbose and r.stderr:
            print(r.stderr)
        return obj, True
    return obj, False


def build(force=False, verbose=False, extra=()):
    srcs = sources()
    if not srcs:
        raise RuntimeError('no HIP sources under ' + CSRC)
    with cf.ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, verbose, list(extra)), srcs))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or _stale(LIB, objs):
        cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose='--verbose' in sys.argv or '-v' in sys.argv))

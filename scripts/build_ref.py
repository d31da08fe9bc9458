"""Same-box A/B: builds the csrc/ of another commit into csrc/libsgdfr_hip_ref.so (travels with the snapshot, git-ignored);
timing scripts then run once with SGDFR_LIB=<that> and once without.   python scripts/build_ref.py <commit> [extra hipcc flags]"""
import os, subprocess, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylegan_directions_face_reenactment_amd import build_native as b
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
commit = sys.argv[1]
tmp = tempfile.mkdtemp()
rel = os.path.relpath(b.CSRC, ROOT)
subprocess.run('git archive %s %s include | tar -x -C %s' % (commit, rel, tmp), shell=True, check=True, cwd=ROOT)
src = os.path.join(tmp, rel)
objs = []
procs = []
for f in sorted(os.listdir(src)):
    if f.endswith('.hip'):
        o = os.path.join(src, f[:-4] + '.o')
        procs.append(subprocess.Popen([b._hipcc()] + b.FLAGS + sys.argv[2:] + ['-c', os.path.join(src, f), '-o', o]))
        objs.append(o)
assert all(p.wait() == 0 for p in procs)
lib = os.path.join(b.CSRC, 'libsgdfr_hip_ref.so')
subprocess.run([b._hipcc(), '--offload-arch=' + b.ARCH, '-shared', '-fPIC', '-o', lib] + objs, check=True)
shutil.rmtree(tmp)
print(lib, 'from', commit)

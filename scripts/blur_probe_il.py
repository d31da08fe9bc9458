"""What bounds the interleaved-plane blur?  Builds probe copies of the library (upfirdn2d.hip with -DSGDFR_BLUR_PROBE=1 no stores,
=2 no plane loads, =3 five adds instead of sixteen FMAs per output, =4 / 5 / 6 non-temporal loads / stores / both) in the container and times scripts/blur_time.py with each:
    python scripts/blur_probe_il.py build       (container)      python scripts/blur_probe_il.py       (GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, 'stylegan_directions_face_reenactment_amd', 'csrc')
if len(sys.argv) > 1 and sys.argv[1] == 'build':
    from stylegan_directions_face_reenactment_amd import build_native as b
    b.build()
    for n in (1, 2, 3, 4, 5, 6):
        obj = os.path.join(CSRC, 'upfirdn2d_probe%d.o' % n)
        subprocess.run([b._hipcc()] + b.FLAGS + ['-DSGDFR_BLUR_PROBE=%d' % n, '-c', os.path.join(CSRC, 'upfirdn2d.hip'), '-o', obj], check=True)
        objs = [s[:-4] + '.o' for s in b.sources() if not s.endswith('upfirdn2d.hip')] + [obj]
        subprocess.run([b._hipcc(), '--offload-arch=' + b.ARCH, '-shared', '-fPIC', '-o', os.path.join(CSRC, 'libsgdfr_hip_blurprobe%d.so' % n)] + objs, check=True)
    print('built')
else:
    for n in (0, 1, 2, 3, 4, 5, 6, 0):
        env = dict(os.environ)
        if n:
            env.update(SGDFR_LIB=os.path.join(CSRC, 'libsgdfr_hip_blurprobe%d.so' % n), SGDFR_ALLOW_LIB_OVERRIDE='1')
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'blur_time.py')], env=env, capture_output=True, text=True)
        print('probe %d: %s' % (n, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)

"""Did a change touch the machine code of a kernel it was not meant to touch?  Compiles one csrc/*.hip of two git revisions (or of a
revision and the working tree) to gfx950 assembly and compares every kernel that exists in both: instruction count, spill traffic
(v_writelane / v_readlane = SGPR spills, scratch_* = VGPR spills), waits, MFMAs, and whether the instruction streams are identical after
label normalisation.  Cross-compiles: no GPU needed.

Round 5 found two regressions of the DEFAULT path this way that no test could see (bit-identical outputs): a block-uniform runtime branch
in wswide.hip's epilogue (870 spilled registers in every instantiation), and a new field in the middle of WsParams (wsplit_kernel<., 6>:
971 instead of 516 SGPR reloads, + 12 % on the 16 x 16 layer).

    python scripts/kernel_asm_diff.py wsplit.hip 4a34bf3            # revision vs working tree
    python scripts/kernel_asm_diff.py split.hip 4a34bf3 HEAD [substring of the kernel names to show]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = 'stylegan_directions_face_reenactment_amd/csrc'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=on', '-fno-slp-vectorize', '-Wno-unused-function', '-S',
         '--cuda-device-only']
COUNT = ('v_mfma', 'ds_read_b128', 'v_writelane_b32', 'v_readlane_b32', 'scratch_load', 'scratch_store', 's_waitcnt', 's_barrier')


def checkout(rev, dst):
    """csrc/ and include/ of a revision (None: the working tree) under dst, laid out so that the relative include of common.h resolves"""
    for sub in (CSRC, 'include'):
        os.makedirs(os.path.join(dst, sub), exist_ok=True)
        if rev is None:
            for f in os.listdir(os.path.join(ROOT, sub)):
                if f.endswith(('.h', '.hip')):
                    with open(os.path.join(ROOT, sub, f), 'rb') as src, open(os.path.join(dst, sub, f), 'wb') as out:
                        out.write(src.read())
        else:
            names = subprocess.run(['git', 'ls-tree', '--name-only', rev, sub + '/'], cwd=ROOT, capture_output=True, text=True, check=True).stdout.split()
            for n in names:
                if n.endswith(('.h', '.hip')):
                    blob = subprocess.run(['git', 'show', '%s:%s' % (rev, n)], cwd=ROOT, capture_output=True, check=True).stdout
                    with open(os.path.join(dst, n), 'wb') as out:
                        out.write(blob)


def kernels(asm_path):
    lines = open(asm_path).read().split('\n')
    out, i = {}, 0
    while i < len(lines):
        m = re.match(r'^(_Z\w+):', lines[i])
        if m and i + 1 < len(lines):
            j = i + 1
            while j < len(lines) and 's_endpgm' not in lines[j] and not re.match(r'^_Z\w+:', lines[j]):
                j += 1
            if j < len(lines) and 's_endpgm' in lines[j]:
                body = [re.sub(r'\.LBB\d+_', '.LBB_', re.sub(r';.*', '', x).strip()) for x in lines[i + 1:j]]
                out[m.group(1)] = [x for x in body if x and not x.startswith('.')]
            i = j
        i += 1
    return out


def demangle(names):
    for tool in ('/opt/rocm/lib/llvm/bin/llvm-cxxfilt', 'c++filt'):
        try:
            r = subprocess.run([tool] + names, capture_output=True, text=True, check=True).stdout.split('\n')
            return dict(zip(names, r))
        except Exception:
            continue
    return {n: n for n in names}


def strip_defaults(name):
    """wswide_kernel<1, false>(..) and wswide_kernel<1>(..) are the same kernel when a defaulted template parameter was added"""
    return re.sub(r'(, (false|0))+>\(', '>(', name)


def main():
    src = sys.argv[1]
    rev_a = sys.argv[2]
    rev_b = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith('-') and len(sys.argv[3]) >= 4 and sys.argv[3] != 'WORK' else None
    show = sys.argv[4] if len(sys.argv) > 4 else (sys.argv[3] if len(sys.argv) > 3 and rev_b is None and sys.argv[3] != 'WORK' else '')
    res = []
    with tempfile.TemporaryDirectory() as tmp:
        for tag, rev in (('a', rev_a), ('b', rev_b)):
            d = os.path.join(tmp, tag)
            checkout(rev, d)
            asm = os.path.join(tmp, tag + '.s')
            subprocess.run(['/opt/rocm/bin/hipcc'] + FLAGS + [os.path.join(d, CSRC, src), '-o', asm], check=True, stderr=subprocess.DEVNULL)
            res.append(kernels(asm))
    a, b = res
    dem = demangle(sorted(set(a) | set(b)))
    by_name_b = {strip_defaults(dem[k]): k for k in b}
    print('%s: %s -> %s' % (src, rev_a, rev_b or 'working tree'))
    for ka in sorted(a, key=lambda k: dem[k]):
        name = strip_defaults(dem[ka])
        if show and show not in name:
            continue
        kb = by_name_b.get(name)
        if kb is None:
            print('  (only in %s) %s' % (rev_a, name[:110]))
            continue
        ca, cb = collections.Counter(), collections.Counter()
        for body, c in ((a[ka], ca), (b[kb], cb)):
            for line in body:
                op = line.split()[0]
                for key in COUNT:
                    if op.startswith(key):
                        c[key] += 1
        same = a[ka] == b[kb]
        delta = ', '.join('%s %d -> %d' % (k, ca[k], cb[k]) for k in COUNT if ca[k] != cb[k])
        ops_same = [x.split()[0] for x in a[ka]] == [x.split()[0] for x in b[kb]]
        verdict = 'IDENTICAL' if same else ('same ops' if ops_same else 'differs')      # same ops: only immediates / registers changed (e.g. kernarg offsets)
        print('  %-9s %6d -> %6d instr  %s   %s' % (verdict, len(a[ka]), len(b[kb]), name[:100], '' if same else ('[' + delta + ']' if delta else '')))


if __name__ == '__main__':
    main()

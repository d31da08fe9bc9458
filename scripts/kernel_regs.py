"""Register / scratch usage of every kernel of one csrc file: python scripts/kernel_regs.py split.hip [filter]"""
import re, subprocess, sys, os
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'stylegan_directions_face_reenactment_amd', 'csrc', sys.argv[1])
out = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=on', '-fno-slp-vectorize', '-c', src, '-o', '/tmp/_regs.o',
                      '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip(); rows[cur] = {}
    for key in ('VGPRs', 'AGPRs', 'VGPR Spill', 'ScratchSize \[bytes/lane\]', 'LDS Size \[bytes/block\]', 'Occupancy \[waves/SIMD\]'):
        m = re.search(r'remark:\s+' + key + r': (\d+)', line)
        if m and cur: rows[cur][key.split(' [')[0].replace('\\', '')] = int(m.group(1))
for k, v in rows.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    print('%-110s %s' % (k[:110], ' '.join('%s=%s' % kv for kv in v.items())))

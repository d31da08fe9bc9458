"""Turns the rocprofv3 outputs of one profiling session (gpurun_out/prof_<tag>, pmc_<tag>_{fetch,write,sq}) into the
committed summaries under profiles/:  python scripts/summarize_pmc.py r1e r01_e [fp16x3|fp32]"""
import collections, csv, json, shutil, sys

tag, out = sys.argv[1], sys.argv[2]
CONV = ('modconv_mfma', 'wino_mfma', 'wino2_mfma', 'split_mfma', 'wsplit_kernel')
precision = sys.argv[3] if len(sys.argv) > 3 else 'fp16x3'


def load(dirn):
    rows = list(csv.DictReader(open('gpurun_out/%s/pmc_counter_collection.csv' % dirn)))
    d = collections.OrderedDict()
    for r in rows:
        key = (int(r['Dispatch_Id']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('sgdfr::', ''), int(r['Grid_Size']))
        d.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
    kt = {int(r['Dispatch_Id']): int(r['End_Timestamp']) - int(r['Start_Timestamp'])
          for r in csv.DictReader(open('gpurun_out/%s/pmc_kernel_trace.csv' % dirn))}
    return d, kt


def last(d, cname):
    """Every dispatch of the LAST forward in the trace (from its first style launch on), in launch order: launches that share
    a kernel name and grid (the capped grids of the blur levels) stay separate rows."""
    keys = list(d.keys())
    starts = [i for i, (disp, name, grid) in enumerate(keys) if name.startswith('styles_batched_kernel<0>')]
    first = starts[-1] if starts else 0
    o = collections.OrderedDict()
    for n, (disp, name, grid) in enumerate(keys[first:]):
        if cname in d[(disp, name, grid)]:
            o[(name, grid, n)] = (d[(disp, name, grid)][cname], disp)
    return o


shutil.copy('gpurun_out/prof_%s/bench_kernel_stats.csv' % tag, 'profiles/%s_kernel_stats.csv' % out)
f, _ = load('pmc_%s_fetch' % tag)
w, _ = load('pmc_%s_write' % tag)
sq, kt = load('pmc_%s_sq' % tag)
ff, ww = last(f, 'FETCH_SIZE'), last(w, 'WRITE_SIZE')
L = ['# rocprofv3 PMC passes (%s)\n' % out,
     'Each counter set in its own pass with `--kernel-trace --output-format csv` only: `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`,',
     '`--pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE`,',
     'command `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --precision %s` (B=64, 256x256, cm=1).  Kernel-trace stats of the' % precision,
     'timing run (`--kernel-trace --stats`, `--steps 10 --warmup 3`): `profiles/%s_kernel_stats.csv`.\n' % out,
     'FETCH_SIZE / WRITE_SIZE are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-B requests',
     'as 64 B, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE equals the algorithmic output bytes of every launch exactly.\n',
     '## HBM traffic of one forward (last forward in the trace)\n',
     '| kernel | grid (work-items) | FETCH_SIZE KiB raw | read MB (x2) | WRITE_SIZE KiB | write MB |', '|---|---|---|---|---|---|']
tr = tw = br = bw = 0
for k, (v, _) in ff.items():
    if not any(t in k[0] for t in CONV + ('blur', 'torgb')):
        continue
    wv = ww.get(k, (0, 0))[0]
    L.append('| `%s` | %d | %.0f | %.1f | %.0f | %.1f |' % (k[0], k[1], v, 2 * v * 1024 / 1e6, wv, wv * 1024 / 1e6))
    if any(t in k[0] for t in CONV):
        tr += 2 * v * 1024
        tw += wv * 1024
    if 'blur' in k[0]:
        br += 2 * v * 1024
        bw += wv * 1024
L.append('\nConv kernels per forward (13 launches): read %.2f GB (corrected) + write %.2f GB = %.2f GB, %.1f MB per launch, %.1f MB per image.'
         % (tr / 1e9, tw / 1e9, (tr + tw) / 1e9, (tr + tw) / 13 / 1e6, (tr + tw) / 64 / 1e6))
L.append('Blur launches per forward: read %.2f GB (corrected) + write %.2f GB; read / written = %.2f (1.0 = every parity plane fetched once).'
         % (br / 1e9, bw / 1e9, br / max(bw, 1)))
L.append('\n## SQ counters per conv launch\n')
L.append('GRBM_GUI_ACTIVE is summed over the 8 XCDs: clock = GRBM_GUI_ACTIVE / 8 / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs).\n')
L.append('| kernel | grid | dur us | clock GHz | MFMA busy % | wait_inst / wave_cycles | wait_any / wave_cycles | LDS bank-conflict cycles |')
L.append('|---|---|---|---|---|---|---|---|')
seen = collections.OrderedDict()
for (disp, name, grid), c in sq.items():
    if any(t in name for t in CONV):
        seen[(name, grid)] = (disp, c)
for (name, grid), (disp, c) in seen.items():
    dur = kt.get(disp, 0)
    gui = c.get('GRBM_GUI_ACTIVE', 0) / 8
    wc = max(c.get('SQ_WAVE_CYCLES', 1), 1)
    L.append('| `%s` | %d | %.0f | %.2f | %.1f | %.2f | %.2f | %.3g |' % (
        name, grid, dur / 1e3, gui / dur if dur else 0, 100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui * 1024) if gui else 0,
        c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_LDS_BANK_CONFLICT', 0)))
open('profiles/%s_pmc.md' % out, 'w').write('\n'.join(L) + '\n')
sys.path.insert(0, '.')
import bench
try:
    src_hash = open('gpurun_out/source_hash_%s.txt' % tag).read().strip()      # written on the GPU box by profile_round.sh
except OSError:
    src_hash = bench.kernel_source_hash()
json.dump({'config': {'batch': 64, 'cm': 1, 'size': 256, 'precision': precision}, 'source_hash': src_hash,
           'conv_launches_per_forward': 13, 'read_bytes_per_forward': tr,
           'write_bytes_per_forward': tw, 'bytes_per_launch': (tr + tw) / 13,
           'source': 'profiles/%s_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 gfx950 correction)' % out},
          open('profiles/traffic_latest.json', 'w'), indent=1)
print('\n'.join(L[-25:]))

"""Turns the rocprofv3 outputs of one profiling session (gpurun_out/prof_<tag>, pmc_<tag>_{fetch,write,sq}) into the
committed summaries under profiles/:  python scripts/summarize_pmc.py r1e r01_e [fp16x3|fp32]

Every table lists the launches of the LAST forward of its trace in dispatch order, one row per launch (launches that share a
kernel name and a grid -- the persistent conv kernels, the capped blur grids -- stay separate rows), so the tracked summary
reproduces every per-launch figure of the bench line without gpurun_out/."""
import collections, csv, json, shutil, sys

tag, out = sys.argv[1], sys.argv[2]
CONV = ('modconv_mfma', 'wino_mfma', 'wino2_mfma', 'split_mfma', 'wsplit_kernel', 'wswide_kernel')
HBM = ('blur', 'torgb')
precision = sys.argv[3] if len(sys.argv) > 3 else 'fp16x3'
HBM_PEAK, HBM_COPY = 8.0e12, 6.29e12      # MI355X_MICROARCH.md: spec / measured float4 copy


def short(name):
    return name.split('(')[0].replace('void ', '').replace('sgdfr::', '')


def load(dirn):
    rows = list(csv.DictReader(open('gpurun_out/%s/pmc_counter_collection.csv' % dirn)))
    d = collections.OrderedDict()
    for r in rows:
        key = (int(r['Dispatch_Id']), short(r['Kernel_Name']), int(r['Grid_Size']))
        d.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
    kt = {int(r['Dispatch_Id']): int(r['End_Timestamp']) - int(r['Start_Timestamp'])
          for r in csv.DictReader(open('gpurun_out/%s/pmc_kernel_trace.csv' % dirn))}
    return d, kt


def last_forward(keys):
    """Index of the first launch of the last forward in a dispatch-ordered key list (its first style launch)."""
    starts = [i for i, k in enumerate(keys) if k[1].startswith('styles_batched_kernel<0>')]
    return starts[-1] if starts else 0


def last(d):
    """[(name, grid, dispatch id, counters)] of the last forward, in launch order."""
    keys = list(d.keys())
    return [(name, grid, disp, d[(disp, name, grid)]) for disp, name, grid in keys[last_forward(keys):]]


def timing_last_forward():
    """[(name, duration ns)] of the last forward of the un-instrumented timing run (--kernel-trace --stats)."""
    try:
        rows = sorted(csv.DictReader(open('gpurun_out/prof_%s/bench_kernel_trace.csv' % tag)), key=lambda r: int(r['Start_Timestamp']))
    except OSError:
        return []
    keys = [(0, short(r['Kernel_Name']), 0) for r in rows]
    first = last_forward(keys)
    return [(short(r['Kernel_Name']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows[first:]]


shutil.copy('gpurun_out/prof_%s/bench_kernel_stats.csv' % tag, 'profiles/%s_kernel_stats.csv' % out)
f, fkt = load('pmc_%s_fetch' % tag)
w, _ = load('pmc_%s_write' % tag)
sq, kt = load('pmc_%s_sq' % tag)
ff, ww, ss = last(f), last(w), last(sq)
tl = timing_last_forward()
same_seq = [n for n, _ in tl] == [r[0] for r in ff]      # the timing run launched the same sequence: its durations line up by index
L = ['# rocprofv3 PMC passes (%s)\n' % out,
     'Each counter set in its own pass with `--kernel-trace --output-format csv` only: `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`,',
     '`--pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE`,',
     'command `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --precision %s` (B=64, 256x256, cm=1).  Kernel-trace stats of the' % precision,
     'timing run (`--kernel-trace --stats`, `--steps 10 --warmup 3`): `profiles/%s_kernel_stats.csv`.\n' % out,
     'FETCH_SIZE / WRITE_SIZE are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 128-B requests',
     'as 64 B, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE equals the algorithmic output bytes of every launch exactly.',
     'Every table: the launches of the LAST forward of its trace, one row per launch, in dispatch order (# = position in the forward).\n',
     '## HBM traffic of one forward\n',
     '| # | kernel | grid (work-items) | FETCH_SIZE KiB raw | read MB (x2) | WRITE_SIZE KiB | write MB | dur us (%s) | GB/s | of 8.0 TB/s | of 6.29 TB/s |' %
     ('timing run' if same_seq else 'FETCH pass'),
     '|---|---|---|---|---|---|---|---|---|---|---|']
tr = tw = br = bw = 0
hbm_kernels = []
bt = 0.0
for n, (name, grid, disp, c) in enumerate(ff):
    if not any(t in name for t in CONV + HBM) or 'FETCH_SIZE' not in c:
        continue
    v = c['FETCH_SIZE']
    wv = ww[n][3].get('WRITE_SIZE', 0) if n < len(ww) and ww[n][0] == name else 0
    rd, wr = 2 * v * 1024, wv * 1024
    dur = (tl[n][1] if same_seq else fkt.get(disp, 0)) * 1e-9
    gbs = (rd + wr) / dur if dur else 0
    L.append('| %d | `%s` | %d | %.0f | %.1f | %.0f | %.1f | %.1f | %.0f | %.3f | %.3f |' % (
        n, name, grid, v, rd / 1e6, wv, wr / 1e6, dur * 1e6, gbs / 1e9, gbs / HBM_PEAK, gbs / HBM_COPY))
    if any(t in name for t in CONV):
        tr += rd
        tw += wr
    else:
        hbm_kernels.append({'kernel': name, 'grid': grid, 'read': rd, 'write': wr, 'us': round(dur * 1e6, 1)})
    if 'blur' in name:
        br += rd
        bw += wr
        bt += dur
L.append('\nConv kernels per forward (13 launches): read %.2f GB (corrected) + write %.2f GB = %.2f GB, %.1f MB per launch, %.1f MB per image.'
         % (tr / 1e9, tw / 1e9, (tr + tw) / 1e9, (tr + tw) / 13 / 1e6, (tr + tw) / 64 / 1e6))
L.append('Blur launches per forward: read %.2f GB (corrected) + write %.2f GB in %.1f us = %.2f TB/s = %.2f of the 8.0 TB/s spec, %.2f of the '
         '6.29 TB/s copy rate; read / written = %.2f.' % (br / 1e9, bw / 1e9, bt * 1e6, (br + bw) / max(bt, 1e-12) / 1e12,
                                                         (br + bw) / max(bt, 1e-12) / HBM_PEAK, (br + bw) / max(bt, 1e-12) / HBM_COPY, br / max(bw, 1)))
L.append('Whole forward: %.1f MB per image through HBM (conv + blur + ToRGB launches; algorithmic minimum of SURVEY 8d: 145.9).'
         % ((tr + tw + sum(k['read'] + k['write'] for k in hbm_kernels)) / 64 / 1e6))
L.append('\n## SQ counters per conv launch (every launch of the last forward)\n')
L.append('GRBM_GUI_ACTIVE is summed over the 8 XCDs: clock = GRBM_GUI_ACTIVE / 8 / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs).\n')
L.append('| # | kernel | grid | dur us | clock GHz | MFMA busy % | wait_inst / wave_cycles | wait_any / wave_cycles | LDS bank-conflict cycles |')
L.append('|---|---|---|---|---|---|---|---|---|')
for n, (name, grid, disp, c) in enumerate(ss):
    if not any(t in name for t in CONV):
        continue
    dur = kt.get(disp, 0)
    gui = c.get('GRBM_GUI_ACTIVE', 0) / 8
    wc = max(c.get('SQ_WAVE_CYCLES', 1), 1)
    L.append('| %d | `%s` | %d | %.0f | %.2f | %.1f | %.2f | %.2f | %.3g |' % (
        n, name, grid, dur / 1e3, gui / dur if dur else 0, 100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui * 1024) if gui else 0,
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
           'write_bytes_per_forward': tw, 'bytes_per_launch': (tr + tw) / 13, 'hbm_kernels': hbm_kernels,
           'source': 'profiles/%s_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2 gfx950 correction)' % out},
          open('profiles/traffic_latest.json', 'w'), indent=1)
print('\n'.join(L[-40:]))

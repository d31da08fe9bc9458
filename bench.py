#!/usr/bin/env python
"""Headline benchmark: reenacted frames/s at 256x256 (BASELINE.json), MI355X-native generator path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config synthesis|inference|trainer] [--batch B] [--cm 1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no torchrun environment starts the N ranks ITSELF (re-executes under torch.distributed.run on
127.0.0.1, one process per GPU, backend nccl = RCCL); it refuses to run when fewer than N devices are visible, and every
rank checks WORLD_SIZE == --gpus, so a line that says n_gpus = N was measured on N RCCL ranks.

--config (BASELINE.json `configs`):
  synthesis (default; configs[1], and configs[3] at N > 1)  StyleGAN2 synthesis network, Generator(256, 512, 8, cm=1), random
            W+ codes [64, 14, 512] PER GPU resident in HBM, fixed noise buffers, psi = 1.  A step = one forward of the batch
            -> [64, 3, 256, 256] fp32 images.
  inference (configs[2])  the run_inference.py flow at B = 32: the source W+ code comes from the e4e encoder once (outside the
            timed loop, reported as `e4e_source_ms`); a step = 3DMM parameters of 32 target frames (resident) -> shift vectors
            (device kernel) -> DirectionMatrix -> shift + truncation psi = 0.7 -> generator -> uint8 source|target|reenacted
            video frames [32, 256, 768, 3].
  trainer   (configs[4], per-rank shape B = 16)  one direction-learning step of libs/trainer.py:155-189: two no-grad forwards
            (source, target) + shape-model stand-in, make_shift_vector_50 on the device, grad forward + backward to A through
            the HIP generator, loss heads = IR-SE-50 id loss + LPIPS-shaped stack + DECA stand-in (scripts/loss_heads.py, stock
            PyTorch-ROCm, random weights), one flat all-reduce of A's gradients, Adam step.  value = samples/s.
Synthetic deterministic weights everywhere (no checkpoints exist offline).  Weak scaling: every rank works on its own shard
of the global batch; rank 0's weights reach the others in ONE flat RCCL broadcast before the timed region and inference
has no collective inside it.

Arithmetic of the 3x3 convs (--precision): fp16x3 (default; fp32 operands as fp16 hi+lo, three MFMA products, fp32
accumulation), fp32 (fp32 MFMA + Winograd kernels), bf16x3.  The synthesis run also times the other arithmetic
(`alt_arithmetic`, with its own roofline).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      the 3x3 conv launches of a step: ALGORITHMIC FLOPs (2*9*Cin*Cout per input pixel, whatever multiplies the
                kernel really issues) / HIP-event time on the launch stream; peak = 2500/3 TFLOP/s for the split arithmetics
                (dense 16-bit MFMA peak / 3 products per fp32 product), 157.3 TFLOP/s (fp32 MFMA) for --precision fp32;
                `per_layer` lists every conv instantiation (us per launch, algorithmic TFLOP/s, fraction of that peak)
  cpu_baseline  the oracle (CPU PyTorch restatement of the reference generator, kind "port") timed on this host's cores on
                a bounded sample of the same workload (N=1, synthesis only): thread sweep, then B=2 and B=8
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC for RCCL; must be in place before the HIP runtime starts

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from stylegan_directions_face_reenactment_amd import distributed as D          # noqa: E402
from stylegan_directions_face_reenactment_amd import functional as F_          # noqa: E402
from stylegan_directions_face_reenactment_amd import synthetic as S            # noqa: E402
from stylegan_directions_face_reenactment_amd import timing                 # noqa: E402
from stylegan_directions_face_reenactment_amd.model import Generator           # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
SPLIT_PEAK_TFLOPS = 2500.0 / 3     # dense fp16/bf16 MFMA peak (same guide) / 3 MFMA products per fp32 product
F8_CROSS_PEAK_TFLOPS = 1.0 / (1 / 2500.0 + 1 / 5000.0)      # opt-in fp8 cross terms: one fp16 + one fp8 (5 PFLOP/s dense) MFMA per product
HBM_PEAK_GBS = 8000.0              # same guide, "HBM3E peak BW": 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0        # same guide: 6.29 TB/s measured (float4 copy)
SEED = 7
AFFINITY = None                    # per-rank CPU binding of an N > 1 run (distributed.bind_rank), echoed into the line
DEFAULT_BATCH = {'synthesis': 64, 'inference': 32, 'trainer': 16, 'pti': 1}
DTYPE = {
    'fp32': 'f32',
    'fp16x3': 'f32 (conv operands as fp16 hi+lo after an exact power-of-two range shift: 22 significant bits while '
              '|x*s| >= 2^-9 of the layer scale, absolute floor below; 3 MFMA products, f32 accumulate)',
    'bf16x3': 'f32 (conv operands as bf16 hi+lo = 16 significant bits, 3 MFMA products, f32 accumulate)'}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', choices=('synthesis', 'inference', 'trainer', 'pti'), default='synthesis')
    ap.add_argument('--batch', type=int, default=None, help='per GPU (default: 64 synthesis / 32 inference / 16 trainer)')
    ap.add_argument('--cm', type=int, default=1, help='channel_multiplier (1 = voxceleb-256, the headline config)')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--layers', action='store_true', help='also print the per-layer conv table to stderr')
    ap.add_argument('--precision', choices=('fp16x3', 'fp32', 'bf16x3'), default='fp16x3')
    ap.add_argument('--no-alt', action='store_true', help='skip the extra leg that times the other arithmetic')
    ap.add_argument('--streams', type=int, default=2,
                    help='synthesis config: consecutive (independent) batches alternate between this many HIP streams '
                         '(functional.StreamPipeline; 1 = every step on one stream, also reported as `single_stream`)')
    ap.add_argument('--sustain', type=float, default=6.0,
                    help='seconds of back-to-back steps after the K-step region (synthesis; 0 = skip): the `sustained` figure')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='N=1 synthesis only: skip the short inference (configs[2]) and trainer (configs[4] per-rank shape) legs')
    ap.add_argument('--no-oracle-delta', action='store_true', help='skip max_abs_vs_oracle (rank 0 runs the CPU oracle on 2 rows)')
    ap.add_argument('--cpu-worker', default=None, help=argparse.SUPPRESS)      # size,cm,threads,B,seconds,max_reps,bind (cpu_baseline's subprocess)
    ap.add_argument('--host-check', action='store_true',
                    help='run only the multi-rank host flow (launch, process group, weight broadcast, sharding) on CPU '
                         'tensors over gloo and print what each rank saw -- no generator launch, no GPU needed')
    args = ap.parse_args(argv)
    if args.batch is None:
        args.batch = DEFAULT_BATCH[args.config]
    return args


# ------------------------------------------------------------------------------------------------ N-rank launch

def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """--gpus N without a torchrun environment: start the N ranks here, one per GPU (the single-device lines this
    replaces: run_inference.py:31 / libs/trainer.py:25 `device = 'cuda'`)."""
    if not args.host_check:
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus and os.environ.get('SGDFR_ALLOW_GPU_SHARING') != '1':
            raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible -- refusing to stack ranks on one device '
                             '(SGDFR_ALLOW_GPU_SHARING=1 + SGDFR_DIST_BACKEND=gloo allows it for flow tests)' % (args.gpus, n_dev))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, min(16, (os.cpu_count() or 1) // args.gpus))))     # each rank then pins itself (main)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def host_check(args, rank, world):
    """The N-rank host flow on CPU tensors: every rank builds the generator, rank 0 fills it, ONE flat broadcast, then each
    rank reports its shard and a checksum of what it received."""
    import torch.distributed as dist
    G = Generator(args.size, 512, 8, channel_multiplier=args.cm)
    if rank == 0:
        G.load_state_dict(S.synthetic_state_dict({k: v for k, v in G.state_dict().items()}, seed=SEED))
    t0 = time.perf_counter()
    nbytes = D.broadcast_state(G, src=0)
    t_b = time.perf_counter() - t0
    lo, hi = D.shard_range(args.batch * world, rank, world)
    mine = torch.tensor([float(sum(v.double().sum() for v in G.state_dict().values())), float(lo), float(hi)], dtype=torch.float64)
    seen = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(seen, mine)
    else:
        seen = [mine]
    if rank == 0:
        sums = [float(s[0]) for s in seen]
        # the same assembly as a real line (base_line + finalize_line): the host flow fills what it can measure on CPU, every
        # device-side field is null WITH the reason -- the JSON contract of an N-rank line is checked without a GPU
        why = 'host check: the multi-rank host flow on CPU tensors, no kernel was launched'
        line = base_line(args, world, 'host_check', 'frames/s', 0.0, 0.0, why, {'weight_broadcast_bytes': nbytes})
        line.update({'value': None, 'ms_per_step': None, 'backend': dist.get_backend() if world > 1 else None,
                     'weight_broadcast_bytes': nbytes, 'broadcast_ms': round(t_b * 1e3, 2),
                     'shards': [[int(s[1]), int(s[2])] for s in seen], 'rank_affinity': AFFINITY,
                     'weights_identical_on_all_ranks': all(x == sums[0] for x in sums),
                     'roofline': {'bound': 'mfma', 'achieved': None, 'peak': round(FP32_MFMA_PEAK_TFLOPS if args.precision == 'fp32' else SPLIT_PEAK_TFLOPS, 1), 'unit': 'TFLOP/s', 'frac': None,
                                  'traffic': None, 'reason': why},
                     'max_abs_vs_oracle': {'last_rank_shard': None, 'reason': why}})
        emit(line, args, world)


# ------------------------------------------------------------------------------------------------ shared pieces

def generator_state_template(size, cm):
    """Key -> zero tensor with FIR buffers at constructor values (no oracle import: product-side helper)."""
    G = Generator(size, 512, 8, channel_multiplier=cm)
    return G, {k: v for k, v in G.state_dict().items()}


def kernel_source_hash():
    """Content hash of everything that defines the kernels (csrc/*, include/*): a PMC traffic figure is only echoed into
    the bench line when it was measured on exactly these sources (the GPU box has no .git to ask for HEAD)."""
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, 'stylegan_directions_face_reenactment_amd')
    files = []
    for d, ext in ((os.path.join(pkg, 'csrc'), ('.hip', '.h')), (os.path.join(ROOT, 'include'), ('.h',)), (pkg, ('.py',))):
        files += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(ext)]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def pmc_traffic(args, B):
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes (profiles/traffic_latest.json; PMC counters cannot be
    collected from inside the timed process).  None unless the profile is for this shape AND these kernel sources."""
    path = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None
    if t.get('config') != {'batch': B, 'cm': args.cm, 'size': args.size, 'precision': args.precision}:
        return None
    if t.get('source_hash') != kernel_source_hash():
        return None
    return t


def attach_pmc(roof, t):
    """PMC figures of the committed profile (same shape, same kernel sources) into a roofline dict: `traffic` = HBM bytes per conv
    launch; per HBM-bound launch the measured bytes (read x2 correction applied by scripts/summarize_pmc.py) and the rate they imply."""
    if t is None:
        return
    roof['traffic'] = round(t['bytes_per_launch'])
    roof['traffic_source'] = t.get('source')
    ker = t.get('hbm_kernels') or []
    rows = roof.get('hbm_bound') or []
    if len(ker) == len(rows):          # both are the blur / ToRGB launches of one forward in launch order
        for r, k in zip(rows, ker):
            r['pmc_mb'] = round((k['read'] + k['write']) / 1e6, 2)
            r['pmc_gbs'] = round((k['read'] + k['write']) / (r['us'] * 1e-6) / 1e9, 1)
            r['kernel'] = k['kernel']


def timed_region(step, args, dev, finish=None):
    """W warm-up steps, then EXACTLY K steps between (barrier + synchronize) pairs; returns (max over ranks, this rank).
    finish: called after the K-th step inside the timed region (a pipelined step checks the PREVIOUS step's result: the last
    step's own check belongs to the K steps too)."""
    import gc
    for _ in range(max(args.warmup, 1)):
        out = step()
    # (the K-step region is a fraction of a second -- 0.11 s at the driver's K=20: one cyclic-GC pause of the interpreter inside it
    #  costs several per cent, so the collector runs BEFORE the region and is held off inside it)
    gc.collect()
    gc_was = gc.isenabled()
    gc.disable()
    try:
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        if finish is not None:
            finish()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        D.barrier()
        elapsed = time.perf_counter() - t0
    finally:
        if gc_was:
            gc.enable()
    return D.max_over_ranks(elapsed, dev), mine, out


def conv_roofline(step, steps, peak, kernel_desc, units_per_step):
    """HIP events around every MFMA conv launch of `steps` steps, on the launch stream."""
    with timing.collect() as t:
        for _ in range(steps):
            step()
    torch.cuda.synchronize()
    rec, hbm_rec = t.conv, t.hbm
    per_layer = {}
    for e0, e1, flops, desc in rec:
        a = per_layer.setdefault(desc, [0.0, 0.0, 0])
        a[0] += e0.elapsed_time(e1) * 1e-3
        a[1] += flops
        a[2] += 1
    conv_s = sum(a[0] for a in per_layer.values())
    conv_flops = sum(a[1] for a in per_layer.values())
    n_launch = sum(a[2] for a in per_layer.values())
    achieved = conv_flops / conv_s / 1e12
    # launches whose cross terms are one fp8 MFMA have their own peak (row_peak): the leg's `peak` is the FLOP-weighted harmonic mean of
    # its rows' peaks = the rate at which the same launch mix would run with every MFMA pipe at its dense peak
    base_peak = peak
    peak = conv_flops / sum(a[1] / row_peak(desc, base_peak) for desc, a in per_layer.items())
    return {'bound': 'mfma', 'kernel': kernel_desc, 'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
            'frac': round(achieved / peak, 4), 'traffic': None,
            'avg_launch_us': round(conv_s / n_launch * 1e6, 2), 'launches_per_step': n_launch // steps,
            'conv_ms_per_step': round(conv_s / steps * 1e3, 3),
            'alg_gflop_per_unit': round(conv_flops / (units_per_step * steps) / 1e9, 3),
            'per_layer': [_layer_entry(desc, sec, fl, n, base_peak) for desc, (sec, fl, n) in per_layer.items()],
            'kernel_families': kernel_families(per_layer, steps, base_peak, conv_s, units_per_step),
            'hbm_bound': hbm_rows(hbm_rec, steps)}


def _family(desc, batch):
    """Kernel family of a conv row: the first two words of its description (three for 'bwd ...' rows); the F(4,3) rows are told
    apart by the kernel that really runs them -- wswide_kernel (128 x 128 tiles) where the library's own query says so, else the
    64-tile wsplit_kernel."""
    import re
    words = desc.split()
    n = next((i for i, w in enumerate(words) if '->' in w), 2)          # everything in front of the 'Cin->Cout' word
    key = ' '.join(words[:n])
    m = re.search(r'F\(4,3\) (\d+)->(\d+) @(\d+)x(\d+)', desc)
    if m and batch:
        cin, cout, h, w = (int(x) for x in m.groups())
        try:
            if F_._shape_query('sgdfr_modconv2d_wsplit_wide', int(batch), cin, cout, h, w):
                key += '/wide'
        except Exception:       # noqa: BLE001
            pass
    return key + (' [f8 cross]' if desc.endswith('[f8 cross]') else '')


def kernel_families(per_layer, steps, peak, conv_s, batch=None):
    """The conv launches of a step grouped by kernel family (the first two words of a row's description: 'split mode1' = the transposed
    conv of split_kernel.h, 'wsplit F(4,3)' = the Winograd form, 'split mode0' = the direct plain conv, 'bwd ...' = the adjoints), largest
    share of conv time first: launches and us per step, algorithmic TFLOP/s, fraction of the row peak, share of the conv time.  The
    first entry is the DOMINANT kernel of the line (roofline.dominant_kernel)."""
    fam = {}
    for desc, (sec, fl, n) in per_layer.items():
        key = _family(desc, batch)
        a = fam.setdefault(key, [0.0, 0.0, 0, 0.0])
        a[0] += sec
        a[1] += fl
        a[2] += n
        a[3] += fl / row_peak(desc, peak)
    rows = [{'name': k, 'launches_per_step': a[2] // steps, 'us_per_step': round(a[0] / steps * 1e6, 1),
             'avg_launch_us': round(a[0] / a[2] * 1e6, 1), 'tflops': round(a[1] / a[0] / 1e12, 1), 'frac': round(a[3] / a[0] / 1e12, 3),
             'share_of_conv_time': round(a[0] / conv_s, 3)} for k, a in fam.items()]
    return sorted(rows, key=lambda r: -r['us_per_step'])


def hbm_rows(rec, steps):
    """The HBM-bound launches of a step (the FIR blur that finishes every transposed conv, ToRGB / its finish), one row per launch
    of a step in launch order: ALGORITHMIC bytes (every input read once, every output written once; timing.timed_hbm)
    / HIP-event time, against the 8.0 TB/s HBM3E spec (`frac`) and the 6.29 TB/s a float4 copy reaches on this part
    (`frac_of_achievable`).  `pmc_gbs` is filled from the committed rocprofv3 passes when they match these sources."""
    if not rec:
        return []
    per_step = len(rec) // steps
    rows = []
    for i in range(per_step):
        sec = sum(rec[k * per_step + i][0].elapsed_time(rec[k * per_step + i][1]) for k in range(steps)) * 1e-3 / steps
        nbytes, desc = rec[i][2], rec[i][3]
        if any(rec[k * per_step + i][3] != desc for k in range(steps)):
            return []          # the steps did not launch the same sequence: no per-launch table
        gbs = nbytes / sec / 1e9
        rows.append({'launch': desc, 'us': round(sec * 1e6, 1), 'alg_mb': round(nbytes / 1e6, 2), 'gbs': round(gbs, 1),
                     'frac': round(gbs / HBM_PEAK_GBS, 3), 'frac_of_achievable': round(gbs / HBM_ACHIEVABLE_GBS, 3), 'pmc_mb': None,
                     'pmc_gbs': None})
    return rows


def row_peak(desc, peak):
    """Peak a conv launch is priced against: `peak` (the arithmetic's), except launches tagged '[f8 cross]' by functional._f8_tag -- one
    fp16 MFMA + one fp8 MFMA per fp32 product: 1 / (1/2500 + 1/5000) TFLOP/s."""
    return F8_CROSS_PEAK_TFLOPS if desc.endswith('[f8 cross]') else peak


def _layer_entry(desc, sec, fl, n, peak):
    """One row of `per_layer`.  `frac` is ALGORITHMIC FLOPs / time / peak; a Winograd F(2x2,3x3) launch issues
    16 of the direct conv's 36 multiplies per output tile, so its algorithmic `frac` can exceed 1 -- `mfma_frac` is the share of
    the MFMA pipe it really occupies (frac * 16/36)."""
    peak = row_peak(desc, peak)
    e = {'layer': desc, 'us': round(sec / n * 1e6, 1), 'tflops': round(fl / sec / 1e12, 1), 'frac': round(fl / sec / 1e12 / peak, 3)}
    if desc.endswith('[f8 cross]'):
        e['peak'] = round(peak, 1)
    if desc.startswith('wino'):
        e['mfma_frac'] = round(e['frac'] * 16 / 36, 3)
    if desc.startswith('wsplit'):      # 1-D Winograd form of the split conv: F(2,3) issues 12 of the direct conv's 18 MFMA columns per
        e['mfma_frac'] = round(e['frac'] * (0.5 if 'F(4,3)' in desc else 2 / 3), 3)     # output pair, F(4,3) 18 of 36 per output quad
    return e


def roofline_for(precision, step, steps, units_per_step):
    if precision == 'fp32':
        return conv_roofline(step, steps, FP32_MFMA_PEAK_TFLOPS,
                             'wino_mfma_kernel (large plain 3x3 layers) + modconv_mfma_kernel (small plain, transposed)',
                             units_per_step)
    r = conv_roofline(step, steps, SPLIT_PEAK_TFLOPS, 'split_mfma_kernel (plain + transposed 3x3 conv launches, %s) + wsplit_kernel '
                      '(its 1-D Winograd F(4,3) form on the plain layers with Cin >= 128: half the MFMA work, see mfma_frac)' % precision,
                      units_per_step)
    r['kernel_short'] = 'all 3x3 conv launches of a step: split_mfma_kernel + wswide_kernel / wsplit_kernel (%s)' % precision
    r['peak_note'] = ('dense 16-bit MFMA peak 2500 TFLOP/s / 3 products per fp32 product; the same achieved figure is %.2fx '
                      'the 157.3 TFLOP/s fp32-MFMA peak' % (r['achieved'] / FP32_MFMA_PEAK_TFLOPS))
    r['measured_mfma_ceiling'] = measured_ceiling(precision, r['achieved'])
    return r


def measured_ceiling(precision, achieved):
    """What the 16-bit MFMA sustains on THIS device (sgdfr_mfma_ceiling_probe, ~50 ms): the chip clocks to its power budget and
    MFMA power follows operand toggling, so a bare MFMA loop on random operands runs well below the nominal 2.5 PFLOP/s that
    `peak` is derived from.  All figures / 3 = fp32-product TFLOP/s, comparable with `achieved`."""
    z = F_.mfma_ceiling(precision, lds_fragments=False, random_operands=False)
    rr = F_.mfma_ceiling(precision, lds_fragments=False, random_operands=True)
    rl = F_.mfma_ceiling(precision, lds_fragments=True, random_operands=True)
    return {'unit': 'TFLOP/s (16-bit MFMA rate / 3 products)', 'zero_operands_register_loop': round(z / 3, 1),
            'random_operands_register_loop': round(rr / 3, 1), 'random_operands_lds_fed_loop': round(rl / 3, 1),
            'achieved_over_random_register_loop': round(achieved / (rr / 3), 3),
            'achieved_over_random_lds_fed_loop': round(achieved / (rl / 3), 3),
            'note': 'bare v_mfma_f32_32x32x16 loops, 8 waves per CU on 256 CUs: power-limited rates of this device; the '
                    'lds-fed loop reads its fragments at the conv kernel\'s ratio (8 ds_read_b128 per 12 MFMAs)'}


def oracle_delta(size, cm, w2, images):
    """max-abs difference between the HIP images and the oracle (checker side; CPU PyTorch restatement of the reference generator,
    pinned to the real reference by tests/golden) on THE SAME latents: w2 = the first rows of the timed batch, images =
    {arithmetic: the HIP images of those rows}.  fp32 oracle for all rows, fp64 oracle for the first one."""
    from oracle import sg2_oracle as O      # allowed here: checker only
    P = S.synthetic_state_dict(O.template_state(size, 512, 8, cm), seed=SEED)
    w2 = w2.detach().float().cpu()
    with torch.no_grad():
        ref32, _ = O.generator_forward(P, [w2], input_is_latent=True)
        ref64, _ = O.generator_forward(O.cast_state(P, torch.float64), [w2[:1].double()], input_is_latent=True)
    out = {'rows': int(w2.shape[0]), 'oracle': 'oracle/sg2_oracle.py generator_forward (fp32 torch-CPU; fp64 for row 0)',
           'max_abs_image': round(float(ref32.abs().max()), 4), 'bar': 1e-3}
    for name, img in images.items():
        img = img.detach().cpu()
        out[name] = float((img.double() - ref32.double()).abs().max())
        out[name + '_vs_fp64_oracle_row0'] = float((img[:1].double() - ref64).abs().max())
    out['fp32_oracle_vs_fp64_oracle_row0'] = float((ref32[:1].double() - ref64).abs().max())
    out['within_bar'] = all(out[k] <= 1e-3 for k in images)
    return out


def cpu_worker(size, cm, threads, B, seconds, max_reps, bind=0):
    """One leg of the CPU baseline in its OWN process (`bench.py --cpu-worker ...`): the oracle at a fixed thread count on the
    first B rows of the timed batch, one warm-up forward, then forwards until `seconds` or `max_reps`.  A fresh process per
    thread count: inside one process the idle workers of a larger OpenMP team kept disturbing the legs that followed it (round 3:
    39 frames/s in a 2-repetition sweep sample against 13.7 in the 20-forward leg at the same thread count).  bind=1: the process is
    confined to the first `threads` CPUs it may run on (distinct physical cores of one socket on the GPU boxes' EPYC hosts, where
    SMT siblings are numbered +128) and the parent sets OMP_PROC_BIND / OMP_PLACES -- unbound, a 16-thread team wanders over 256
    logical CPUs and two NUMA nodes (the fresh-process legs measured 2.5-7x below the in-process ones)."""
    if bind:
        cpus = sorted(os.sched_getaffinity(0))[:max(1, threads)]
        os.sched_setaffinity(0, cpus)
    from oracle import sg2_oracle as O      # allowed here: the cpu_baseline leg only
    torch.set_num_threads(threads)
    P = S.synthetic_state_dict(O.template_state(size, 512, 8, cm), seed=SEED)
    w = S.synthetic_latents(SEED, 8, key='bench.w')[:B]          # rows 0..B-1 of the timed batch
    with torch.no_grad():
        O.generator_forward(P, [w], input_is_latent=True)          # warm-up
        t0, reps = time.perf_counter(), 0
        while True:
            O.generator_forward(P, [w], input_is_latent=True)
            reps += 1
            el = time.perf_counter() - t0
            if el >= seconds or reps >= max_reps:
                break
    print(json.dumps({'frames_per_s': B * reps / el, 'reps': reps, 'seconds': el, 'threads': threads, 'batch': B, 'bound': bool(bind)}),
          flush=True)


def cpu_baseline(size, cm, budget_s=26.0):
    """Times the oracle (checker side) on the host CPU, every leg in a fresh subprocess (cpu_worker) while this process -- the
    one that owns the GPU context -- sleeps in subprocess.run.  A short sweep (1.5 s per leg, B=2) over thread counts, each
    unbound and bound to as many cores of one socket, PICKS the configuration; two sustained legs then run at it (B=2, and B=8)
    and `value` is the BEST SUSTAINED rate of the two -- every leg is listed beside it."""
    host = os.cpu_count() or 1

    def leg(threads, B, seconds, max_reps, bind, timeout=120):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES='')
        if bind:
            env.update(OMP_PROC_BIND='close', OMP_PLACES='cores')
        cmd = [sys.executable, os.path.abspath(__file__), '--cpu-worker', '%d,%d,%d,%d,%g,%d,%d' % (size, cm, threads, B, seconds, max_reps, int(bind))]
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
            return json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:      # noqa: BLE001  (a failed leg must not take the bench line down)
            return {'frames_per_s': 0.0, 'reps': 0, 'seconds': 0.0, 'threads': threads, 'batch': B, 'bound': bool(bind), 'error': str(e)[:200]}
    t_start = time.perf_counter()
    sweep = {}
    for bind in (1, 0):
        peak = 0.0
        for thr in sorted({min(t, host) for t in (8, 16, 32, 64)}):
            if time.perf_counter() - t_start > budget_s * 0.45:
                break
            r = sweep[(thr, bind)] = round(leg(thr, 2, 1.5, 6, bind)['frames_per_s'], 2)
            if r < 0.7 * peak:
                break
            peak = max(peak, r)
    best_thr, best_bind = max(sweep, key=sweep.get)
    left = max(10.0, budget_s - (time.perf_counter() - t_start))
    main = leg(best_thr, 2, left * 0.5, 40, best_bind)
    b8 = leg(best_thr, 8, left * 0.35, 6, best_bind)
    top = max((main, b8), key=lambda r: r['frames_per_s'])
    # SURVEY.md 8d's protocol as written -- torch.set_num_threads(os.cpu_count()), unbound -- beside the best-of-sweep value (the
    # oracle's grouped convs do not scale past one socket's worth of threads; both figures belong in the line)
    # (bounded: on the 256-thread GPU hosts an unbound all-cores team did not finish ONE batch-2 forward in 120 s -- r06_a -- so the leg
    #  runs batch 1 under a 15 s limit and reports the bound it proves when it times out)
    allc = leg(host, 1, 2.0, 2, 0, timeout=15) if host != best_thr or best_bind else main
    # one thread per PHYSICAL core (SMT siblings are numbered + host/2 on the GPU hosts): the largest team that still finishes -- measured
    # outside the bench on such a host (round 6): 64 threads 3.5, 128 threads 2.0 frames/s, 256 threads 58.7 s per batch-1 forward = 0.017
    phys = leg(max(1, host // 2), 1, 2.0, 4, 0, timeout=20) if host >= 16 else None
    if allc.get('error') and 'timed out' in allc['error']:
        allc['upper_bound_frames_per_s'] = round(1 / 15.0, 3)
    return {'value': round(top['frames_per_s'], 3), 'unit': 'frames/s', 'cores': best_thr, 'host_cores': host, 'kind': 'port',
            'bound_to_cores': bool(best_bind),
            'all_cores_value': round(allc['frames_per_s'], 3) if allc['reps'] else None,
            'all_cores_note': ('%d threads, unbound: no forward finished in 15 s (< %.3f frames/s)' % (host, 1 / 15.0)) if not allc['reps']
            else '%d threads, unbound, batch %d' % (host, allc['batch']),
            'all_cores_leg': {k: (round(v, 3) if isinstance(v, float) else v) for k, v in allc.items()},
            'physical_cores_value': round(phys['frames_per_s'], 3) if (phys and phys['reps']) else None,
            'physical_cores': max(1, host // 2) if phys else None,
            'sustained_legs': [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in (main, b8)],
            'thread_sweep_batch2_short': {'%d threads, %s' % (t, 'bound' if b else 'unbound'): v for (t, b), v in sweep.items()},
            'sample': '%d forwards of batch %d in %.1f s (`value` = the better of two sustained legs: batch 2 and batch 8, first rows of '
                      'the timed batch), Generator(%d, cm=%d) synthesis-only, torch-CPU fp32 oracle (oracle/sg2_oracle.py) at %d threads%s -- '
                      'the best of a short sweep (1.5 s per configuration, picks the configuration only); every leg in a fresh process'
                      % (top['reps'], top['batch'], top['seconds'], size, cm, best_thr,
                         ' bound to %d cores of one socket' % best_thr if best_bind else ', unbound')}


def sustained_leg(step, units_per_step, ms_per_step, dev, seconds, probe=100, join=None):
    """The same step back to back for >= `seconds` of wall time (the K-step region above is a sub-second burst on a chip that
    clocks to its power budget): frames/s over the whole leg and HIP-event times of its first and last `probe` steps."""
    n = max(int(seconds * 1e3 / max(ms_per_step, 1e-3)) + 1, 3 * probe)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        if i in (probe, n - probe) and join is not None:
            join()                       # (steps on slot streams: the probing stream waits for them before it stamps)
        if i == probe:
            ev[1].record()
        if i == n - probe:
            ev[2].record()
        step()
    if join is not None:
        join()
    ev[3].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    D.barrier()
    wall = D.max_over_ranks(wall, dev)
    first, last = ev[0].elapsed_time(ev[1]) / probe, ev[2].elapsed_time(ev[3]) / probe
    return n, wall, {'steps': n, 'seconds': round(wall, 3), 'ms_per_step': round(wall / n * 1e3, 4),
                     'first_%d_ms_per_step' % probe: round(first, 4), 'last_%d_ms_per_step' % probe: round(last, 4),
                     'last_over_first': round(last / first, 4)}


def rank_spread(frames_local, mine, dev, world):
    """frames/s of the slowest and fastest rank (each rank's own K steps, before the closing barrier)."""
    import torch.distributed as dist
    v = torch.tensor([frames_local / mine], dtype=torch.float64, device=dev)
    if world == 1:
        return [round(float(v), 2)] * 2
    allv = [torch.zeros_like(v) for _ in range(world)]
    dist.all_gather(allv, v)
    vals = [float(x) for x in allv]
    return [round(min(vals), 2), round(max(vals), 2)]


def base_line(args, world, metric, unit, value, elapsed, workload, extra_cfg):
    return {'metric': metric, 'value': round(value, 2), 'unit': unit, 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': DTYPE[args.precision], 'data': 'synthetic',
            'config': dict({'workload': workload, 'per_gpu_batch': args.batch, 'global_batch': args.batch * world,
                            'resolution': args.size, 'channel_multiplier': args.cm,
                            'parallelism': 'batch-sharded x%d, no data-path collective' % world,
                            'rank_affinity': AFFINITY}, **extra_cfg)}


CONTRACT_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def finalize_line(out, args, world):
    """The JSON contract of the driver's BENCH / SCALE steps, enforced on every line before it is printed (and on the CPU
    host-check line of tests/test_distributed.py, so the first real N > 1 run cannot fail on the format): every contract key
    present; `roofline` with bound / achieved / peak / unit / frac / traffic; `cpu_baseline` an object -- measured on rank 0 at
    N = 1, otherwise value null WITH the reason; for N > 1 synthesis lines the last rank's oracle check beside rank 0's."""
    if not isinstance(out.get('cpu_baseline'), dict):
        out['cpu_baseline'] = {'value': None, 'unit': out.get('unit'), 'cores': None, 'kind': 'port', 'sample': None,
                               'reason': ('the CPU baseline is timed on rank 0 at N=1 only (the N=1 line of the same commit carries it): '
                                          'with %d ranks the host cores are divided between the ranks' % world) if world > 1
                               else ('--no-cpu-baseline' if args.no_cpu_baseline else 'this --config times it in the synthesis line only')}
    missing = [k for k in CONTRACT_KEYS if k not in out]
    roof = out.get('roofline') or {}
    missing += ['roofline.' + k for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic') if k not in roof]
    if world > 1 and args.config == 'synthesis' and not args.no_oracle_delta and not args.host_check:
        if 'last_rank_shard' not in (out.get('max_abs_vs_oracle') or {}):
            missing.append('max_abs_vs_oracle.last_rank_shard')
    if out.get('n_gpus') != world:
        missing.append('n_gpus == WORLD_SIZE')
    if not args.host_check and F_.config().cross_terms != 'fp16':
        # the opt-in fp8 cross terms are narrower than the default arithmetic: they are a leg beside `value`, never `value` itself
        raise SystemExit("bench.py: `value` must be measured with cross_terms == 'fp16' (got %r)" % F_.config().cross_terms)
    if missing:
        raise SystemExit('bench.py: the line misses contract keys: %s' % ', '.join(missing))
    return out


LINE_LIMIT = 4096          # bytes of the LAST stdout line (round 5's 21 KB line was not parsed by the driver: BENCH_r05.parsed = null)
SHORT_DTYPE = {'fp32': 'f32 (fp32 MFMA + Winograd)', 'fp16x3': 'f32 (operands as fp16 hi+lo, 3 MFMA products, f32 accumulate)',
               'bf16x3': 'f32 (operands as bf16 hi+lo, 3 MFMA products, f32 accumulate)'}
DETAIL_FILE = 'bench_detail.json'


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _leg_triple(leg):
    """{value, unit, ms_per_step, frac(+ peak)} of a secondary leg (everything else about it is in the detail object)."""
    if not isinstance(leg, dict):
        return None
    if 'error' in leg:
        return {'error': str(leg['error'])[:160]}
    roof = leg.get('conv_roofline') or leg.get('roofline') or {}
    t = _pick(leg, 'value', 'unit', 'ms_per_step')
    t.update({k: roof[k] for k in ('frac', 'peak') if k in roof and not (k == 'peak' and abs(roof[k] - SPLIT_PEAK_TFLOPS) < 0.1)})
    for k in ('max_abs_vs_oracle', 'max_abs_between_the_two_paths', 'launches_per_step', 'tflops', 'hbm_gbs', 'hbm_frac', 'precision',
              'batches', 'rerendered_batches', 'plan_widenings', 'max_abs_vs_fp32_kernels'):
        if isinstance(leg.get(k), (int, float, str)):
            t[k] = leg[k]
    for k in ('generator_only_ms_per_step', 'e4e_source_ms'):
        if isinstance((leg.get('detail') or {}).get(k), (int, float)):
            t[k] = leg['detail'][k]
    if isinstance(leg.get('one_batch_at_a_time'), dict):
        t['one_batch_at_a_time'] = leg['one_batch_at_a_time'].get('value')
    return t


def compact_line(full, args):
    """The line the driver parses: the contract keys, `roofline` and `cpu_baseline` as SCALARS, one {value, ms_per_step, frac} triple per
    other leg -- <= LINE_LIMIT bytes whatever legs are added later.  Tables (per-layer rows, HBM rows, CPU sweep, ...) live in the
    detail object only (emit)."""
    out = {k: full[k] for k in CONTRACT_KEYS[:10]}
    out['dtype'] = SHORT_DTYPE.get(args.precision, full['dtype']) if full.get('dtype') in DTYPE.values() else str(full['dtype'])[:200]
    out['data'] = full['data']
    cfg = full['config']
    out['config'] = _pick(cfg, 'workload', 'per_gpu_batch', 'global_batch', 'resolution', 'channel_multiplier', 'parallelism',
                          'weight_broadcast_bytes', 'weight_broadcast_ms', 'per_rank_frames_per_s_min_max', 'per_rank_samples_per_s_min_max',
                          'generator_only_ms_per_step', 'e4e_source_ms', 'ranks')
    out['config']['workload'] = str(out['config'].get('workload', ''))[:240]
    out['config']['parallelism'] = str(out['config'].get('parallelism', ''))[:120]
    mo = full.get('max_abs_vs_oracle')
    if isinstance(mo, dict):
        out['max_abs_vs_oracle'] = mo.get(args.precision)
        out['max_abs_bar'] = mo.get('bar', 1e-3)
        if isinstance(mo.get('last_rank_shard'), dict):
            out['max_abs_vs_oracle_last_rank'] = mo['last_rank_shard'].get(args.precision)
    roof = full['roofline']
    r = {k: roof.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')}
    r['kernel'] = str(roof.get('kernel_short') or roof.get('kernel', ''))[:120]
    r.update(_pick(roof, 'avg_launch_us', 'launches_per_step', 'conv_ms_per_step', 'alg_gflop_per_unit', 'traffic_source', 'reason'))
    if isinstance(roof.get('end_to_end'), dict):
        r['end_to_end'] = _pick(roof['end_to_end'], 'achieved', 'frac', 'single_stream_frac')
    fam = roof.get('kernel_families') or []
    if fam:
        r['dominant_kernel'] = _pick(fam[0], 'name', 'launches_per_step', 'avg_launch_us', 'tflops', 'frac', 'share_of_conv_time')
    hb = roof.get('hbm_bound') or []
    blur = [x for x in hb if x['launch'].startswith('blur')]
    if blur:
        us, mb = sum(x['us'] for x in blur), sum(x['alg_mb'] for x in blur)
        r['blur'] = {'launches_per_step': len(blur), 'us_per_step': round(us, 1), 'alg_gbs': round(mb / us * 1e3, 1),
                     'frac_of_hbm_peak': round(mb / us * 1e3 / HBM_PEAK_GBS, 3)}
    out['roofline'] = r
    cb = full['cpu_baseline']
    out['cpu_baseline'] = {k: cb.get(k) for k in ('value', 'unit', 'cores', 'kind')}
    out['cpu_baseline'].update(_pick(cb, 'host_cores', 'all_cores_value', 'all_cores_note', 'physical_cores', 'physical_cores_value',
                                     'bound_to_cores', 'reason'))
    out['cpu_baseline']['sample'] = None if cb.get('sample') is None else str(cb['sample'])[:90]
    for k in ('verified', 'rerendered_batches', 'fp16_saturated_pairs', 'fp16_range_mode'):
        if k in full:
            out[k] = full[k]
    legs = {}
    for k in ('single_stream', 'default_call', 'unverified', 'one_batch_at_a_time'):
        if isinstance(full.get(k), dict):
            legs[k] = _pick(full[k], 'value', 'ms_per_step')
    if isinstance(full.get('sustained'), dict):
        legs['sustained'] = _pick(full['sustained'], 'frames_per_s', 'seconds', 'vs_value')
    for k in ('alt_arithmetic', 'fallback_arithmetic'):
        if isinstance(full.get(k), dict):
            legs[k] = _leg_triple(full[k])
            legs[k].pop('unit', None)
    for k, v in (full.get('other_configs') or {}).items():
        legs[k] = _leg_triple(v)
    if legs:
        out['legs'] = legs
    out['detail'] = DETAIL_FILE + ' / stderr line BENCH_DETAIL: every table of every leg'
    return out


def _sig(v, digits=5):
    """Floats of the compact line at `digits` significant digits (1.430511474609375e-05 -> 1.4305e-05); containers recursively."""
    if isinstance(v, float):
        return float('%.*g' % (digits, v))
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items()}
    if isinstance(v, list):
        return [_sig(x, digits) for x in v]
    return v


def emit(full, args, world):
    """Rank 0's output.  The full record goes to bench_detail.json beside this file and to ONE stderr line ('BENCH_DETAIL {...}');
    stdout gets exactly one line, the compact one, LAST."""
    full = finalize_line(full, args, world)
    compact = compact_line(full, args)
    exact = {k: compact[k] for k in ('value', 'ms_per_step') if k in compact}       # (the contract's own figures stay as computed)
    compact = _sig(compact)
    compact.update(exact)
    line = json.dumps(compact)
    if len(line) > LINE_LIMIT:
        raise SystemExit('bench.py: the compact line is %d bytes (> %d): move the new keys into the detail object' % (len(line), LINE_LIMIT))
    detail = json.dumps(full)
    try:
        with open(os.path.join(ROOT, DETAIL_FILE), 'w') as f:
            f.write(detail + '\n')
    except OSError:
        pass
    sys.stderr.write('BENCH_DETAIL ' + detail + '\n')
    sys.stderr.flush()
    print(line, flush=True)


def build_generator(args, rank, dev):
    """Rank 0 generates the weights; everyone receives ONE flat RCCL broadcast (timed)."""
    G, template = generator_state_template(args.size, args.cm)
    if rank == 0:
        G.load_state_dict(S.synthetic_state_dict(template, seed=SEED))
    G = G.eval().to(dev)
    return G


def broadcast_timed(*objs):
    torch.cuda.synchronize()
    D.barrier()
    t0 = time.perf_counter()
    nbytes = D.broadcast_state(*objs, src=0)
    torch.cuda.synchronize()
    return nbytes, (time.perf_counter() - t0) * 1e3


# ------------------------------------------------------------------------------------------------ configs

def run_synthesis(args, rank, world, dev):
    G = build_generator(args, rank, dev)
    bcast_bytes, bcast_ms = broadcast_timed(G)
    B = args.batch
    lo, hi = D.shard_range(B * world, rank, world)
    w = S.synthetic_latents(SEED, B * world, n_latent=G.n_latent, key='bench.w')[lo:hi].contiguous().to(dev)

    def step1():                                   # one unverified forward on the caller's stream (the roofline pass; `single_stream`)
        img, _ = G([w], input_is_latent=True, verify_range=False)
        return img

    def step_default():                            # the reference-shaped call: no extra keyword -> verified before it returns
        img, _ = G([w], input_is_latent=True)
        return img

    # Steps are independent batches: they alternate between `--streams` HIP streams, so the latency-bound head of a forward
    # (K-sliced 4x4 ... 16x16 layers, ~30 dependent launches) runs beside the big layers of the previous one.  Every kernel
    # and every image is the same as on one stream (tests/test_gpu_generator.py); `single_stream` below is the K-step figure
    # without it, and the per-kernel roofline pass always runs on one stream.
    # (forwards small enough for the generator's own hipGraph replay get one capture per stream: Generator._graph_key)
    pipe = F_.StreamPipeline(args.streams, dev) if args.streams > 1 else None

    def step_unverified():
        if pipe is None:
            return step1()
        with pipe.next():
            return step1()

    # `value`: every batch is VERIFIED against the fp16 range plan before it counts -- the check of batch i (its RangeToken: one
    # event wait + one pinned word) happens after batch i+1 has been queued, so the device never idles for it; a batch that
    # clamped operands would be rendered again in the fallback arithmetic (counted in `rerendered_batches`: 0 on this data).
    pending, rerendered = [None], [0]

    def settle(item):
        if item is None:
            return
        img, tok, stream = item
        if not G.range_ok(tok):
            rerendered[0] += 1
            if stream is not None:
                pipe.join(stream=stream)
            G([w], input_is_latent=True, verify_range=True)

    def step():
        if pipe is None:
            img = step1()
            cur = (img, G.take_range_token(), None)
        else:
            with pipe.next():
                img = step1()
                cur = (img, G.take_range_token(), pipe.last)
        prev, pending[0] = pending[0], cur
        settle(prev)
        return img

    def finish():
        prev, pending[0] = pending[0], None
        settle(prev)

    F_.set_precision(args.precision)
    sustained = single = unverified = default_call = fallback = None
    with torch.no_grad():
        if pipe is not None:
            e1, _, _ = timed_region(step1, args, dev)
            single = {'value': round(B * world * args.steps / e1, 2), 'unit': 'frames/s', 'ms_per_step': round(e1 / args.steps * 1e3, 3),
                      'what': 'unverified forwards back to back on one HIP stream'}
        eu, _, _ = timed_region(step_unverified, args, dev)
        unverified = {'value': round(B * world * args.steps / eu, 2), 'unit': 'frames/s', 'ms_per_step': round(eu / args.steps * 1e3, 3),
                      'what': 'verify_range=False and nobody checks the tokens (round 3\'s `value`)'}
        ed, _, _ = timed_region(step_default, args, dev)
        default_call = {'value': round(B * world * args.steps / ed, 2), 'unit': 'frames/s', 'ms_per_step': round(ed / args.steps * 1e3, 3),
                        'what': 'G([w], input_is_latent=True) as the reference scripts call it: each call waits for its own batch and '
                                'returns verified frames (one stream, hipGraph replay)'}
        elapsed, mine, img = timed_region(step, args, dev, finish=finish)
        assert img.shape == (hi - lo, 3, args.size, args.size) and bool(torch.isfinite(img).all())
        head = {args.precision: img[:2].clone()}            # rows 0, 1 of the timed batch -> max_abs_vs_oracle
        spread = rank_spread((hi - lo) * args.steps, mine, dev, world)
        if args.sustain > 0:
            n_s, wall_s, sustained = sustained_leg(step, B, elapsed / args.steps * 1e3, dev, args.sustain,
                                                   join=pipe.join if pipe is not None else None)
            finish()
            sustained['frames_per_s'] = round(B * world * n_s / wall_s, 2)
        roof = roofline_for(args.precision, step1, args.steps, B)
        attach_pmc(roof, pmc_traffic(args, B))
        alt = None
        alt_mode = 'fp32' if args.precision != 'fp32' else 'fp16x3'
        if not args.no_alt:
            exact = step1()
            F_.set_precision(alt_mode)
            try:
                alt_elapsed, _, fast = timed_region(step_unverified, args, dev)
                alt_roof = roofline_for(alt_mode, step1, args.steps, B)
            finally:
                F_.set_precision(args.precision)
            head[alt_mode] = fast[:2].clone()
            alt = {'precision': alt_mode, 'value': round(B * world * args.steps / alt_elapsed, 2), 'unit': 'frames/s',
                   'ms_per_step': round(alt_elapsed / args.steps * 1e3, 3), 'dtype': DTYPE[alt_mode],
                   'max_abs_between_the_two_paths': float((fast - exact).abs().max()), 'roofline': alt_roof}
            if args.precision == 'fp16x3':
                # what a batch that leaves the fp16 range plan is re-rendered in: the same kernels on bf16 hi+lo terms (fp32 range)
                F_.set_precision('bf16x3')
                try:
                    fb_elapsed, _, fb = timed_region(step_unverified, args, dev)
                finally:
                    F_.set_precision(args.precision)
                head['bf16x3'] = fb[:2].clone()
                fallback = {'precision': 'bf16x3', 'value': round(B * world * args.steps / fb_elapsed, 2), 'unit': 'frames/s',
                            'ms_per_step': round(fb_elapsed / args.steps * 1e3, 3), 'dtype': DTYPE['bf16x3'],
                            'max_abs_vs_default_arithmetic': float((fb - exact).abs().max())}
    # the oracle check on a shard that is NOT rank 0's: the last rank runs it on its first two rows (own range plan, own latents)
    far = None
    if world > 1 and not args.no_oracle_delta:
        mine_delta = oracle_delta(args.size, args.cm, w[:2], {args.precision: head[args.precision]}) if rank == world - 1 else None
        far = D.gather_objects(mine_delta)[world - 1]
    if rank != 0:
        return None
    value = B * world * args.steps / elapsed
    out = base_line(args, world, 'reenacted frames/sec @%dx%d' % (args.size, args.size), 'frames/s', value, elapsed,
                    '%dxMI355X HIP synthesis-only: Generator(%d,512,8,cm=%d), random w+ [%d,14,512] per GPU, fixed noise, psi=1; every '
                    'batch verified against the fp16 range plan (token check one batch behind the launches)'
                    % (world, args.size, args.cm, B),
                    {'weight_broadcast_bytes': bcast_bytes, 'weight_broadcast_ms': round(bcast_ms, 2),
                     'per_rank_frames_per_s_min_max': spread})
    out['verified'] = True
    out['rerendered_batches'] = rerendered[0]
    out['unverified'] = unverified
    out['default_call'] = default_call
    if single is not None:
        out['config']['parallelism'] += '; consecutive batches alternate between %d HIP streams' % args.streams
        out['single_stream'] = single
    if sustained is not None:
        sustained['vs_value'] = round(sustained['frames_per_s'] / value, 4)
        if abs(sustained['vs_value'] - 1) > 0.03:
            sustained['note'] = ('the sustained rate differs from `value` (the contract\'s K-step burst) by %+.1f %%: the chip '
                                 'clocks to its power budget, quote the sustained figure for long runs'
                                 % ((sustained['vs_value'] - 1) * 100))
        out['sustained'] = sustained
    if not args.no_oracle_delta and w.shape[0] >= 2:
        # the second half of BASELINE.json's metric: max-abs delta vs the reference (through its pinned restatement) on the
        # same latents as the timed batch
        out['max_abs_vs_oracle'] = oracle_delta(args.size, args.cm, w[:2], head)
        if far is not None:
            out['max_abs_vs_oracle']['last_rank_shard'] = far
    # end to end: ALL algorithmic FLOPs of a step (the 3x3 convs are 98.8 % of the path) over the timed step, the
    # figure the driver can recompute from `ms_per_step`; `frac` above is the conv launches alone
    step_flops = roof['alg_gflop_per_unit'] * 1e9 * B
    roof['end_to_end'] = {'achieved': round(step_flops / (elapsed / args.steps) / 1e12, 2), 'unit': 'TFLOP/s',
                          'frac': round(step_flops / (elapsed / args.steps) / 1e12 / roof['peak'], 4),
                          'what': 'conv FLOPs of a step / ms_per_step of `value` / peak (everything that is not a conv launch counts as lost time)'}
    if single is not None:
        roof['end_to_end']['single_stream_frac'] = round(step_flops / (single['ms_per_step'] * 1e-3) / 1e12 / roof['peak'], 4)
    out['roofline'] = roof
    if args.precision == 'fp16x3':
        # operand pairs this generator's fp16-split launches had to clamp / found non-finite during the whole run (its own
        # saturation word): 0 = the fp32-grade claim holds
        out['fp16_saturated_pairs'] = G.saturated_pairs()
        out['fp16_range_mode'] = G.range_mode()
    if alt is not None:
        out['alt_arithmetic'] = alt
    if fallback is not None:
        if 'max_abs_vs_oracle' in out:
            fallback['max_abs_vs_oracle'] = out['max_abs_vs_oracle'].get('bf16x3')
        out['fallback_arithmetic'] = fallback
    if args.layers:
        for e in roof['per_layer']:
            sys.stderr.write('%-34s %8.1f us/launch %7.1f TFLOP/s  %.3f\n' % (e['layer'], e['us'], e['tflops'], e['frac']))
        for e in roof['hbm_bound']:
            sys.stderr.write('%-44s %8.1f us/launch %7.1f GB/s  %.3f\n' % (e['launch'], e['us'], e['gbs'], e['frac']))
    if world == 1 and not args.no_other_configs:
        del G, w, img
        torch.cuda.empty_cache()
        out['other_configs'] = other_config_legs(args, rank, world, dev)
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(args.size, args.cm)
    return out


def other_config_legs(args, rank, world, dev):
    """BASELINE.json configs[2] (run_inference flow, B=32) and configs[4]'s per-rank shape (trainer step, B=16) as short legs of
    the default N=1 run, so the driver observes them too; each is the same code as `--config inference|trainer`."""
    legs = {}
    if args.cm != 2 and args.config == 'synthesis':
        # SURVEY's secondary shape set: channel_multiplier 2 (ffhq-256, config_models.py:10-20: 512@64^2, 256@128^2, 128@256^2),
        # the same synthesis-only workload -- frames/s, every conv row, max-abs vs the oracle on the timed batch
        sub = argparse.Namespace(**vars(args))
        sub.cm, sub.no_alt, sub.sustain, sub.no_other_configs, sub.no_cpu_baseline, sub.layers = 2, True, 0, True, True, False
        sub.steps, sub.warmup = max(5, min(args.steps, 20)), max(3, min(args.warmup, 5))
        t0 = time.perf_counter()
        try:
            line = run_synthesis(sub, rank, world, dev)
            legs['synthesis_cm2'] = {
                'metric': line['metric'], 'value': line['value'], 'unit': line['unit'], 'steps': sub.steps, 'warmup': sub.warmup,
                'ms_per_step': line['ms_per_step'], 'per_gpu_batch': sub.batch, 'workload': line['config']['workload'],
                'single_stream': line.get('single_stream', {}).get('value'),
                'max_abs_vs_oracle': (line.get('max_abs_vs_oracle') or {}).get(sub.precision),
                'fp16_saturated_pairs': line.get('fp16_saturated_pairs'),
                'conv_roofline': {k: line['roofline'][k] for k in ('achieved', 'peak', 'unit', 'frac', 'conv_ms_per_step', 'alg_gflop_per_unit')},
                'conv_per_layer': line['roofline']['per_layer'], 'leg_wall_s': round(time.perf_counter() - t0, 1)}
        except Exception as e:
            legs['synthesis_cm2'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        torch.cuda.empty_cache()
    if args.config == 'synthesis' and args.precision == 'fp16x3' and F_.config().cross_terms == 'fp16':
        # The same workload with Config.cross_terms='fp8' (opt-in, include/sgdfr.h SGDFR_SPLIT_FP16F8): the F(4,3) layers that take the
        # wide-tile kernel, the transposed convs they feed (deep plan) and the direct 64 -> 64 layer keep both cross terms in e4m3 -- 2 MFMA
        # units per product instead of 3.  Reported
        # beside `value`, never as it: the default stays three fp16 products (1.4e-5 against the oracle); this leg's max-abs is measured
        # on its own timed batch against the same oracle and the same 1e-3 bar.
        sub = argparse.Namespace(**vars(args))
        sub.no_alt, sub.sustain, sub.no_other_configs, sub.no_cpu_baseline, sub.layers = True, 0, True, True, False
        sub.steps, sub.warmup = max(5, min(args.steps, 20)), max(3, min(args.warmup, 5))
        t0 = time.perf_counter()
        base_cfg = F_.config()
        try:
            F_.set_default(base_cfg.replace(cross_terms='fp8'))
            line = run_synthesis(sub, rank, world, dev)
            legs['synthesis_fp8_cross_terms'] = {
                'metric': line['metric'], 'value': line['value'], 'unit': line['unit'], 'steps': sub.steps, 'warmup': sub.warmup,
                'ms_per_step': line['ms_per_step'], 'per_gpu_batch': sub.batch, 'workload': line['config']['workload'],
                'dtype': 'f32 (fp16 hi+lo operands; on the wide-tile F(4,3) layers, the transposed convs after them and the last direct layer the two cross terms as e4m3 pairs in one fp8 MFMA)',
                'single_stream': line.get('single_stream', {}).get('value'),
                'max_abs_vs_oracle': (line.get('max_abs_vs_oracle') or {}).get(sub.precision), 'bar': 1e-3,
                'fp16_saturated_pairs': line.get('fp16_saturated_pairs'),
                'conv_roofline': {k: line['roofline'][k] for k in ('achieved', 'peak', 'unit', 'frac', 'conv_ms_per_step', 'alg_gflop_per_unit')},
                'conv_per_layer': line['roofline']['per_layer'], 'leg_wall_s': round(time.perf_counter() - t0, 1)}
        except Exception as e:
            legs['synthesis_fp8_cross_terms'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        finally:
            F_.set_default(F_.config().replace(cross_terms=base_cfg.cross_terms))
        torch.cuda.empty_cache()
    if args.config == 'synthesis' and args.precision == 'fp16x3':
        t0 = time.perf_counter()
        try:
            legs['range_plan_stress'] = range_plan_stress(args, dev)
            legs['range_plan_stress']['leg_wall_s'] = round(time.perf_counter() - t0, 1)
        except Exception as e:
            legs['range_plan_stress'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        torch.cuda.empty_cache()
    for name, fn in (('inference', run_inference), ('trainer', run_trainer), ('pti', run_pti)):
        sub = argparse.Namespace(**vars(args))
        sub.config, sub.batch = name, DEFAULT_BATCH[name]
        sub.skip_e4e_batch = True
        sub.steps, sub.warmup = max(5, min(args.steps, 20)), max(3, min(args.warmup, 5))
        t0 = time.perf_counter()
        try:
            line = fn(sub, rank, world, dev)
            legs[name] = {'metric': line['metric'], 'value': line['value'], 'unit': line['unit'], 'steps': sub.steps,
                          'warmup': sub.warmup, 'ms_per_step': line['ms_per_step'], 'per_gpu_batch': sub.batch,
                          'workload': line['config']['workload'],
                          'detail': {k: v for k, v in line['config'].items()
                                     if k in ('e4e_source_ms', 'e4e_batch_images_per_s', 'generator_only_ms_per_step',
                                              'loss_heads_and_optimizer_ms_per_step', 'losses_finite', 'backward_arithmetic')},
                          'conv_roofline': {k: line['roofline'].get(k) for k in ('achieved', 'peak', 'unit', 'frac', 'conv_ms_per_step')},
                          # every conv instantiation of the leg (forward rows, and for the trainer the `bwd ...` dL/dx rows)
                          'conv_per_layer': line['roofline'].get('per_layer'),
                          'leg_wall_s': round(time.perf_counter() - t0, 1)}
            if 'fp16_saturated_pairs' in line:
                legs[name]['fp16_saturated_pairs'] = line['fp16_saturated_pairs']
            if 'one_batch_at_a_time' in line:
                legs[name]['one_batch_at_a_time'] = line['one_batch_at_a_time']
            for k in ('launches_per_step', 'tflops', 'hbm_gbs', 'hbm_frac', 'eager', 'loss_first_last', 'needed_gradients_only'):
                if k in line:
                    legs[name][k] = line[k]
        except Exception as e:          # a neighbour leg must never take the headline line down with it
            legs[name] = {'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        torch.cuda.empty_cache()
    return legs


def range_plan_stress(args, dev, n_batches=200, B=8):
    """VERDICT r5 item 7 as a counter of the default run: the fp16x3 range plan on trained-LIKE weights (heavy-tailed conv weights,
    log-normal modulation biases: synthetic.trained_like_state_dict) with W+ codes from the e4e stand-in on `n_batches` batches of random
    images (every tenth batch 4x louder); each batch goes through the reference-shaped VERIFIED call.  Reports how many batches had to
    be rendered twice and the largest difference of the first / last five batches against the fp32-MFMA kernels of the same generator.
    (tests/test_gpu_generator.py::test_range_plan_on_trained_like_weights holds every frame to the bar and checks rows against the oracle.)"""
    from stylegan_directions_face_reenactment_amd.encoder import Encoder4Editing
    G, template = generator_state_template(args.size, args.cm)
    G.load_state_dict(S.trained_like_state_dict(template, seed=SEED))
    G = G.eval().to(dev)
    enc = Encoder4Editing(50, 'ir_se', args.size).eval()
    enc.load_state_dict(S.synthetic_encoder_state(enc.state_dict(), seed=SEED + 1))
    enc = enc.to(dev)
    tr = S.counter_tensor(SEED, 'rp.t', (1, 512)).to(dev)
    worst = scale = 0.0
    with torch.no_grad():
        for i in range(n_batches):
            x = S.counter_tensor(SEED, 'rp.x.%d' % (i % 16), (B, 3, args.size, args.size), 0.0, 0.5).clamp_(-1, 1).to(dev)
            w = enc(x) * ((4.0 if i % 10 == 9 else 1.0) * (1.0 + 0.01 * (i // 16)))
            img, _ = G([w], input_is_latent=True, truncation=0.7, truncation_latent=tr)
            if i < 5 or i >= n_batches - 5:
                with F_.precision('fp32'):
                    ref, _ = G([w], input_is_latent=True, truncation=0.7, truncation_latent=tr)
                worst, scale = max(worst, float((img - ref).abs().max())), max(scale, float(ref.abs().max()))
    st = G.range_stats()
    return {'value': round(st['rerendered'] / n_batches, 4), 'unit': 'fraction of verified batches rendered twice', 'batches': n_batches,
            'per_gpu_batch': B, 'rerendered_batches': st['rerendered'], 'plan_widenings': st['widenings'], 'mode_after': st['mode'],
            'fp16_saturated_pairs': G.saturated_pairs(), 'max_abs_vs_fp32_kernels': worst, 'max_abs_image': round(scale, 2),
            'workload': 'Generator(%d,cm=%d) with trained-like weights (1 %% of the input channels x30, log-normal modulation biases), W+ from '
                        'the synthetic e4e encoder on random images, every tenth batch x4, psi=0.7, verified G([w]) calls' % (args.size, args.cm)}


def _direction_ranges():
    """[54,2] (min, max) ranges of the 3DMM parameters: the reference's data file libs/configs/ranges_voxceleb.npy as carried
    by the kat8 fixture (the reference checkout does not exist on the GPU box)."""
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'kat8_shift.npz'))['ranges_voxceleb']


def run_inference(args, rank, world, dev):
    from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
    from stylegan_directions_face_reenactment_amd.encoder import Encoder4Editing
    from stylegan_directions_face_reenactment_amd.reenact import ReenactmentSession, grid_frames_uint8
    from stylegan_directions_face_reenactment_amd.shift import ShiftVectors
    G = build_generator(args, rank, dev)
    enc = Encoder4Editing(50, 'ir_se', args.size).eval()
    A = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, verbose=False)
    if rank == 0:
        enc.load_state_dict(S.synthetic_encoder_state(enc.state_dict(), seed=SEED + 1))
        A.load_state_dict(S.synthetic_direction_state(SEED + 2))
    enc, A = enc.to(dev), A.to(dev).eval()
    F_.set_precision(args.precision)
    with torch.no_grad():
        trunc = G.style(S.synthetic_z(SEED, 4096, key='trunc.z').to(dev)).mean(0, keepdim=True) if rank == 0 else \
            torch.empty(1, 512, device=dev)
    bcast_bytes, bcast_ms = broadcast_timed(G, A, [trunc], {k: v for k, v in enc.state_dict().items() if v.dtype == torch.float32})
    B = args.batch
    lo, hi = D.shard_range(B * world, rank, world)
    src_img = S.counter_tensor(SEED, 'c3.src', (1, 3, args.size, args.size), 0.0, 0.5).clamp_(-1, 1).to(dev)
    tgt_img = S.counter_tensor(SEED, 'c3.tgt', (B * world, 3, args.size, args.size), 0.0, 0.5)[lo:hi].clamp_(-1, 1).contiguous().to(dev)
    ang_s, par_s = S.synthetic_shape_params(SEED, 'c3.src', 1)
    ang_t, par_t = S.synthetic_shape_params(SEED, 'c3.tgt', B * world)
    ang_s, par_s = ang_s.to(dev), {k: v.to(dev) for k, v in par_s.items()}
    ang_t, par_t = ang_t[lo:hi].contiguous().to(dev), {k: v[lo:hi].contiguous().to(dev) for k, v in par_t.items()}
    shifts = ShiftVectors('voxceleb', 15, 6.0, ranges=_direction_ranges())

    with torch.no_grad():
        for _ in range(3):                 # warm-up: MIOpen picks its kernels on the first calls
            source_code = enc(src_img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            source_code = enc(src_img)
        torch.cuda.synchronize()
        e4e_ms = (time.perf_counter() - t0) / 5 * 1e3
        sess = ReenactmentSession(G, A, source_code, 0.7, trunc, batch=B, shifts=shifts, streams=args.streams)

        def step1():                       # one batch, start to finish (the roofline pass: kernels of one undisturbed step)
            frames = []
            for img in sess.frames_for_targets(ang_s, par_s, ang_t, par_t):
                frames.append(grid_frames_uint8([src_img, tgt_img[:img.shape[0]], img], swap_rb=True))
            return frames[0]

        # The frames of a video arrive batch after batch: a step queues ITS batch (shift vectors -> DirectionMatrix -> generator,
        # on the session's other HIP stream) and finishes the previous one (range check, uint8 source|target|reenacted frames)
        # -- ReenactmentSession.streaming(), the one-batch look-ahead `frames()` uses inside a longer video.  One batch of B
        # frames enters and one leaves per step.
        live = sess.streaming()
        live.push(sess.shift_vectors_for(ang_s, par_s, ang_t, par_t))

        def step():
            img = live.push(sess.shift_vectors_for(ang_s, par_s, ang_t, par_t))
            return grid_frames_uint8([src_img, tgt_img[:img.shape[0]], img], swap_rb=True)

        elapsed, mine, frames = timed_region(step, args, dev)
        live.flush()
        e1, _, _ = timed_region(step1, args, dev)
        single = {'value': round(B * world * args.steps / e1, 2), 'unit': 'frames/s', 'ms_per_step': round(e1 / args.steps * 1e3, 3),
                  'what': 'every step renders its one batch start to finish (no look-ahead across steps)'}
        # the same with the session's whole step (DirectionMatrix -> shift -> generator) replayed as ONE hipGraph (graph=True)
        sess_g = ReenactmentSession(G, A, source_code, 0.7, trunc, batch=B, shifts=shifts, streams=1, graph=True)

        def step1g():
            for img in sess_g.frames_for_targets(ang_s, par_s, ang_t, par_t):
                out = grid_frames_uint8([src_img, tgt_img[:img.shape[0]], img], swap_rb=True)
            return out
        eg, _, _ = timed_region(step1g, args, dev)
        single['session_graph'] = {'value': round(B * world * args.steps / eg, 2), 'ms_per_step': round(eg / args.steps * 1e3, 3),
                                   'what': 'ReenactmentSession(graph=True): the step as one hipGraph replay'}
        del sess_g
        assert frames.shape == (hi - lo, args.size, 3 * args.size, 3) and frames.dtype == torch.uint8
        spread = rank_spread((hi - lo) * args.steps, mine, dev, world)
        roof = roofline_for(args.precision, step1, args.steps, B)
        e4e_batch = None
        if not getattr(args, 'skip_e4e_batch', False):     # (the leg of the default run skips it: MIOpen tunes the B=32 shapes for ~20 s)
            for _ in range(2):
                enc(tgt_img)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(3):
                enc(tgt_img)
            torch.cuda.synchronize()
            e4e_batch = (hi - lo) * 3 / (time.perf_counter() - tb)
    if rank != 0:
        return None
    out = base_line(args, world, 'reenacted frames/sec @%dx%d' % (args.size, args.size), 'frames/s', B * world * args.steps / elapsed, elapsed,
                    '%dxMI355X run_inference.py flow: e4e source W+ (once) + per batch of %d target frames: shift vectors from 3DMM '
                    'parameters -> DirectionMatrix -> shift + truncation psi=0.7 -> HIP Generator(%d,cm=%d) -> uint8 '
                    'source|target|reenacted frames; batches stream through ReenactmentSession.streaming() (a step queues its '
                    'batch and finishes the previous one)' % (world, B, args.size, args.cm),
                    {'weight_broadcast_bytes': bcast_bytes, 'weight_broadcast_ms': round(bcast_ms, 2),
                     'per_rank_frames_per_s_min_max': spread, 'e4e_source_ms': round(e4e_ms, 2),
                     'e4e_batch_images_per_s': round(e4e_batch, 1) if e4e_batch is not None else None,
                     'not_in_the_timed_step': 'DECA / face detection of the targets (out of scope, SURVEY.md §2); e4e of the one source image'})
    out['roofline'] = roof
    out['one_batch_at_a_time'] = single
    if args.precision == 'fp16x3':
        out['fp16_saturated_pairs'] = G.saturated_pairs()
    return out


def run_trainer(args, rank, world, dev):
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import loss_heads as LH
    from stylegan_directions_face_reenactment_amd.direction_matrix import DirectionMatrix
    from stylegan_directions_face_reenactment_amd.generic import generate_image
    from stylegan_directions_face_reenactment_amd.shift import ShiftVectors
    B = args.batch
    if B % 2:
        raise SystemExit('--config trainer needs an even per-GPU batch (make_shift_vector_50 halves it, utils_train.py:179-184)')
    G = build_generator(args, rank, dev)
    for p in G.parameters():            # only A is optimised (trainer.py:144); the reference leaves requires_grad on and discards
        p.requires_grad_(False)         # the generator's weight gradients
    torch.manual_seed(SEED)
    A = DirectionMatrix(512, input_dim=15, out_dim=512, w_plus=True, num_layers=8, verbose=False).to(dev)
    id_loss, lpips, deca = LH.IdLoss().to(dev).eval(), LH.LpipsShaped().to(dev).eval(), LH.ShapeModelStandIn().to(dev).eval()
    for m in (id_loss, lpips, deca):
        for p in m.parameters():
            p.requires_grad_(False)
    F_.set_precision(args.precision)
    with torch.no_grad():
        trunc = G.style(S.synthetic_z(SEED, 4096, key='trunc.z').to(dev)).mean(0, keepdim=True) if rank == 0 else \
            torch.empty(1, 512, device=dev)
    bcast_bytes, bcast_ms = broadcast_timed(G, A, [trunc], id_loss, lpips, deca)
    opt = torch.optim.Adam(A.parameters(), lr=1e-4, weight_decay=5e-4)             # trainer.py:145
    shifts = ShiftVectors('voxceleb', 15, 6.0, ranges=_direction_ranges())
    lo, hi = D.shard_range(B * world, rank, world)
    zs = S.synthetic_z(SEED, B * world, key='train.zs')[lo:hi].contiguous().to(dev)
    zt = S.synthetic_z(SEED, B * world, key='train.zt')[lo:hi].contiguous().to(dev)
    zst = torch.cat([zs, zt])
    losses = []

    def step():                                                                       # trainer.py:155-189
        with torch.no_grad():
            # trainer.py:157-166 renders source and target with two generator calls; the images of a batch do not depend on their
            # neighbours (tests/test_gpu_generator.py::test_batch_independence_at_bench_size), so both go through ONE forward of
            # 2B rows -- the same frames, fuller tiles
            both = generate_image(G, zst, 0.7, trunc, input_is_latent=False, return_latents=False)
            imgs_source, imgs_target = both[:B], both[B:]
            params_source, angles_source = deca(imgs_source)
            params_target, angles_target = deca(imgs_target)
            shift_vector, which = shifts.make_shift_vector_50(params_source, params_target, angles_source, angles_target)
        shift = A(shift_vector)
        imgs_shifted, _ = generate_image(G, zs, 0.7, trunc, shift_code=shift, input_is_latent=False, return_latents=True)
        params_shifted, _ = deca(imgs_shifted)
        gt = {'pose': torch.cat([params_target['pose'][:B // 2], params_source['pose'][B // 2:]]),     # utils_train.py:288-300
              'alpha_exp': torch.cat([params_target['alpha_exp'][:B // 2], params_source['alpha_exp'][B // 2:]]),
              'alpha_shp': params_source['alpha_shp']}
        loss = 1.0 * deca.landmark_loss(gt, params_shifted) + 10.0 * id_loss(imgs_shifted, imgs_source) + \
            10.0 * lpips(imgs_shifted, imgs_source)                                  # lambdas: config_arguments.py:15-20
        A.zero_grad()
        loss.backward()
        D.allreduce_grads(A)                                                          # the one collective of a step (262 KB)
        opt.step()
        losses.append(loss.detach())
        return imgs_shifted

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        elapsed, mine, img = timed_region(step, args, dev)
    assert img.shape == (B, 3, args.size, args.size)
    spread = rank_spread(B * args.steps, mine, dev, world)
    # where the step goes: generator-only legs timed on their own (same shapes), the rest is the loss heads + optimizer
    def gen_only():
        with torch.no_grad():
            generate_image(G, zst, 0.7, trunc, input_is_latent=False)
        sv = S.counter_tensor(SEED, 'train.sv', (B, 15), 0.0, 3.0).to(dev)
        im = generate_image(G, zs, 0.7, trunc, shift_code=A(sv), input_is_latent=False)
        A.zero_grad()
        im.backward(torch.ones_like(im) * 1e-3)
    for _ in range(2):
        gen_only()
    torch.cuda.synchronize()
    tg = time.perf_counter()
    for _ in range(args.steps):
        gen_only()
    torch.cuda.synchronize()
    gen_ms = (time.perf_counter() - tg) / args.steps * 1e3
    roof = roofline_for(args.precision, gen_only, max(2, args.steps // 4), B)
    finite = bool(torch.isfinite(torch.stack(losses)).all())
    if rank != 0:
        return None
    out = base_line(args, world, 'direction-learning samples/sec @%dx%d' % (args.size, args.size), 'samples/s', B * world * args.steps / elapsed, elapsed,
                    '%dxMI355X libs/trainer.py step: source + target frames in one no-grad forward of 2B rows + shape-model stand-in, make_shift_vector_50 on device, grad '
                    'forward + backward to A through the HIP Generator(%d,cm=%d) (frozen), IR-SE-50 id loss + LPIPS-shaped stack + '
                    'DECA stand-in (PyTorch-ROCm, random weights), all-reduce of dA, Adam; B=%d per GPU'
                    % (world, args.size, args.cm, B),
                    {'weight_broadcast_bytes': bcast_bytes, 'weight_broadcast_ms': round(bcast_ms, 2),
                     'per_rank_samples_per_s_min_max': spread, 'generator_only_ms_per_step': round(gen_ms, 3),
                     'loss_heads_and_optimizer_ms_per_step': round(elapsed / args.steps * 1e3 - gen_ms, 3),
                     'losses_finite': finite, 'loss_first_last': [round(float(losses[0]), 5), round(float(losses[-1]), 5)],
                     'fp16_saturated_pairs_fwd_and_bwd': G.saturated_pairs() if args.precision == 'fp16x3' else None,
                     'forward_arithmetic': args.precision,
                     'backward_arithmetic': 'fp32' if args.precision == 'fp32' else F_.config().backward_arith})
    out['dtype'] = DTYPE[args.precision] + ('; backward: dL/dx convs in the same fp16 hi+lo arithmetic, range-planned per image from max|g| (bf16 hi+lo with '
                                            'SGDFR_BWD_ARITH=bf16x3), no weight gradients (G frozen), everything else f32' if args.precision != 'fp32' else '')
    out['roofline'] = roof
    out['roofline']['note'] = ('conv launches of the generator legs of a step (the no-grad forward of 2B rows + grad forward + the dL/dx convs of '
                               'the backward), loss heads excluded')
    return out


def count_device_launches(fn):
    """Kernel launches (+ device memcpys / memsets) one call of fn() puts on the device, counted by torch.profiler (roctracer); None
    when the profiler is not usable on this box."""
    try:
        from torch.profiler import ProfilerActivity, profile
        fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fn()
            torch.cuda.synchronize()
        n = sum(1 for e in prof.events() if str(getattr(e, 'device_type', '')).endswith('CUDA'))
        return n or None
    except Exception:       # noqa: BLE001
        return None


def run_pti(args, rank, world, dev):
    """the generator fine-tuning step of libs/optimization.py:47-68 (PTI; default-ON in run_inference.py:309) at its
    own shape -- ONE source image per step: forward with grad, L2 stand-in loss, backward (dL/dx, dL/ds AND dL/dW of every layer:
    the reference leaves requires_grad on all generator parameters), Adam over convs[4..11] -- as finetune.optimize_g runs it: the
    whole step replayed as one hipGraph.  `eager` = the same step launched from Python."""
    from stylegan_directions_face_reenactment_amd import finetune as FT
    B = 1
    G = build_generator(args, rank, dev).train()
    params, lam = FT.pti_parameters(G)
    trunc = S.counter_tensor(SEED, 'pti.trunc', (1, 512)).to(dev)
    latent = S.synthetic_latents(SEED, B, n_latent=G.n_latent, key='pti.w').to(dev)
    target = torch.tanh(S.counter_tensor(SEED, 'pti.t', (B, 3, args.size, args.size))).to(dev)
    F_.set_precision(args.precision)
    import warnings
    warnings.simplefilter('ignore')

    def make_step(opt):
        def step():
            img, _ = G([latent], input_is_latent=True, return_latents=False, truncation=0.7, truncation_latent=trunc)
            loss = FT.l2_loss_fn(img, target, lam)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            return loss.detach()
        return step
    eager = make_step(FT.FusedAdam(params, lr=3e-3))
    e_el, _, l0 = timed_region(eager, args, dev)
    launches = count_device_launches(eager)
    with timing.collect() as t:                  # the conv launches of one eager step (forward rows + `bwd ...` rows)
        eager()
    torch.cuda.synchronize()
    conv_s = sum(e0.elapsed_time(e1) for e0, e1, _, _ in t.conv) * 1e-3
    conv_fl = sum(fl for _, _, fl, _ in t.conv)
    runner = FT.GraphedStep(make_step(FT.FusedAdam(params, lr=3e-3)), warmup=3, clear_grads_of=list(G.parameters()))
    elapsed, mine, loss = timed_region(runner, args, dev)
    # the same step computing only the gradients the optimizer reads (finetune.optimize_g(freeze_unused=True)): the weights it
    # produces are the same, the never-read .grad of the other 5 layers / mapping network is not formed
    ids = {id(p) for p in params}
    flags = [(p, p.requires_grad) for p in G.parameters()]
    for p in G.parameters():
        p.requires_grad_(id(p) in ids)
    try:
        runner2 = FT.GraphedStep(make_step(FT.FusedAdam(params, lr=3e-3)), warmup=3, clear_grads_of=list(G.parameters()))
        needed_el, _, _ = timed_region(runner2, args, dev)
    finally:
        for p, rg in flags:
            p.requires_grad_(rg)
    if rank != 0:
        return None
    # algorithmic work of a step: forward convs + dL/dx convs + dL/dW convs = 3 x the forward's conv FLOPs; bytes:
    # every weight read by forward and by dL/dx, every weight gradient written once (3 x 4 B x all conv weights), Adam reads p, g, m, v
    # and writes p, m, v of the optimised ones, the activations are written, re-read by the backward, and their gradients written + read
    w_all = sum(p.numel() for p in G.parameters())
    w_opt = sum(p.numel() for p in params)
    fwd_gflop = sum(2 * 9 * c.conv.in_channel * c.conv.out_channel * (r if not c.conv.upsample else r // 2) ** 2
                    for c, r in zip([G.conv1] + list(G.convs), [4] + [2 ** (3 + i // 2) for i in range(len(G.convs))])) / 1e9
    step_gflop = 3 * fwd_gflop * B
    act_bytes = 4 * 145.9e6 * B * (args.size / 256.0) ** 2
    step_bytes = 3 * 4 * w_all + 7 * 4 * w_opt + act_bytes
    ms = elapsed / args.steps * 1e3
    out = base_line(args, world, 'PTI fine-tuning steps/sec @%dx%d' % (args.size, args.size), 'steps/s', world * args.steps / elapsed, elapsed,
                    '%dxMI355X optimization.py:47-68 step at B=1: grad forward + L2 stand-in + backward (dx, ds, dW of every layer) + Adam over '
                    'convs[4..11] of Generator(%d,cm=%d), the step replayed as ONE hipGraph (finetune.GraphedStep)' % (world, args.size, args.cm),
                    {'optimised_parameters': w_opt, 'generator_parameters': w_all})
    out['config']['per_gpu_batch'] = out['config']['global_batch'] = B
    out['eager'] = {'ms_per_step': round(e_el / args.steps * 1e3, 3), 'device_launches_per_step': launches,
                    'conv_launches_per_step': len(t.conv), 'conv_ms_per_step': round(conv_s * 1e3, 3),
                    'conv_tflops': round(conv_fl / conv_s / 1e12, 2) if conv_s else None}
    out['needed_gradients_only'] = {'ms_per_step': round(needed_el / args.steps * 1e3, 3),
                                    'what': 'optimize_g(freeze_unused=True): only the gradients Adam reads (convs[4..11]); same updated weights'}
    out['launches_per_step'] = launches
    out['tflops'] = round(step_gflop / ms, 2)          # GFLOP / ms = TFLOP/s
    out['hbm_gbs'] = round(step_bytes / (ms * 1e-3) / 1e9, 1)
    out['hbm_frac'] = round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    out['loss_first_last'] = [round(float(l0), 6), round(float(loss), 6)]
    peak = FP32_MFMA_PEAK_TFLOPS
    out['roofline'] = {'bound': 'mfma', 'kernel': 'whole PTI step (forward + dL/dx + dL/dW convs; fp32-grade arithmetic)', 'achieved': out['tflops'],
                       'peak': peak, 'unit': 'TFLOP/s', 'frac': round(out['tflops'] / peak, 4), 'traffic': None,
                       'alg_gflop_per_step': round(step_gflop, 2), 'alg_mb_per_step': round(step_bytes / 1e6, 1),
                       'hbm': {'achieved_gbs': out['hbm_gbs'], 'peak_gbs': HBM_PEAK_GBS, 'frac': out['hbm_frac']},
                       'note': 'B=1: 3 x %.1f GFLOP against %.0f MB of weights, gradients and Adam state -- the step is launch- and HBM-latency-'
                               'bound, not MFMA-bound; both fractions are reported' % (fwd_gflop, step_bytes / 1e6)}
    return out


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.cpu_worker:
        f = args.cpu_worker.split(',')
        return cpu_worker(int(f[0]), int(f[1]), int(f[2]), int(f[3]), float(f[4]), int(f[5]), int(f[6]) if len(f) > 6 else 0)
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))
    if args.host_check:
        os.environ.setdefault('SGDFR_DIST_BACKEND', 'gloo')
    if D.env_world()[2] != args.gpus:       # before any rendezvous: a mismatched launch must fail, not hang or mis-report
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, D.env_world()[2]))
    rank, local_rank, world = D.init_from_env(backend='gloo' if args.host_check else None, use_gpu=not args.host_check)
    global AFFINITY
    if world > 1:       # one rank = one GPU = one slice of that GPU's NUMA node (a 1-rank run keeps the whole host: CPU baseline)
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', world))
        AFFINITY = D.gather_objects(D.bind_rank(local_rank, local_world, use_gpu=not args.host_check))
    if args.host_check:
        host_check(args, rank, world)
        # (leave through the same door as a real run: a rank that lets the interpreter tear a live gloo group down races its
        #  worker threads -- "terminate called without an active exception", SIGABRT about once in five 8-rank launches)
        D.shutdown()
        return
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    dev = torch.device('cuda', torch.cuda.current_device() if world > 1 else 0)
    torch.cuda.set_device(dev)
    out = {'synthesis': run_synthesis, 'inference': run_inference, 'trainer': run_trainer, 'pti': run_pti}[args.config](args, rank, world, dev)
    if rank == 0:
        emit(out, args, world)
    D.shutdown()


if __name__ == '__main__':
    main()

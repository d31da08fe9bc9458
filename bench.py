#!/usr/bin/env python
"""Headline benchmark: reenacted frames/s at 256x256 (BASELINE.json), MI355X-native generator path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64] [--cm 1] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] / configs[3]): StyleGAN2 synthesis network, Generator(256, 512, 8,
channel_multiplier=1), random W+ codes [64, 14, 512] PER GPU already resident in HBM, fixed noise buffers,
psi=1 (synthesis only), fp32 tensors end to end, synthetic deterministic weights (no checkpoints exist offline).
Arithmetic of the 3x3 convs (--precision): fp16x3 (default; fp32 operands split into fp16 hi+lo, three MFMA products,
fp32 accumulation -- held to the same error bound as the fp32 kernels by tests/test_gpu_split.py), fp32 (fp32 MFMA +
Winograd kernels), bf16x3.  The default run also times the other arithmetic (`alt_arithmetic`).
A step = one forward of the whole batch -> [64, 3, 256, 256] fp32 images.  Weak scaling: every rank
generates its own 64-latent shard of the global batch; rank 0's weights are broadcast once over RCCL
before the timed region and there is no collective inside it.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      the 13 3x3 conv launches of a forward: ALGORITHMIC FLOPs (2*9*Cin*Cout per input pixel, whatever
                multiplies the kernel really issues) / HIP-event time on the launch stream.  peak = 2500/3 TFLOP/s for the
                split arithmetics (dense 16-bit MFMA peak / 3 products), 157.3 TFLOP/s (fp32 MFMA) for --precision fp32
  cpu_baseline  the oracle (CPU PyTorch restatement of the reference generator, kind "port") timed on this
                host's cores on a bounded sample of the same workload (N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from stylegan_directions_face_reenactment_amd import distributed as D          # noqa: E402
from stylegan_directions_face_reenactment_amd import functional as F_          # noqa: E402
from stylegan_directions_face_reenactment_amd import synthetic as S            # noqa: E402
from stylegan_directions_face_reenactment_amd.model import Generator           # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
SPLIT_PEAK_TFLOPS = 2500.0 / 3     # dense fp16/bf16 MFMA peak (same guide) / 3 MFMA products per fp32 product
SEED = 7


def generator_state_template(size, cm):
    """Key -> zero tensor with FIR buffers at constructor values (no oracle import: product-side helper)."""
    G = Generator(size, 512, 8, channel_multiplier=cm)
    return G, {k: v for k, v in G.state_dict().items()}


def cpu_baseline(size, cm, budget_s=20.0):
    """Times the oracle (checker side) on the host CPU: B=2 forwards of the same synthesis workload."""
    from oracle import sg2_oracle as O      # allowed here: the cpu_baseline leg only
    # 16 threads is this workload's sweet spot on the GPU box's 2x64-core EPYC 9575F (measured: 8 thr 5.4,
    # 16 thr 5.65, 32 thr 4.1, 64 thr 2.5, 256 thr 0.03 frames/s): oneDNN's small grouped convs do not scale further
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    P = S.synthetic_state_dict(O.template_state(size, 512, 8, cm), seed=SEED)
    B = 2
    w = S.synthetic_latents(SEED, B, key='cpu.w')
    with torch.no_grad():
        O.generator_forward(P, [w], input_is_latent=True)          # warm-up
        t0 = time.perf_counter()
        reps = 0
        while True:
            O.generator_forward(P, [w], input_is_latent=True)
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget_s or reps >= 20:
                break
    return {'value': round(B * reps / el, 3), 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'sample': '%d forwards of batch %d, Generator(%d, cm=%d) synthesis-only, torch-CPU fp32 oracle '
                      '(oracle/sg2_oracle.py), %.1f s' % (reps, B, size, cm, el)}


def pmc_traffic(args, B):
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes (profiles/traffic_latest.json; PMC
    counters cannot be collected from inside the timed process).  None when the profile is for another shape."""
    path = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None
    if t.get('config') != {'batch': B, 'cm': args.cm, 'size': args.size, 'precision': args.precision}:
        return None
    return round(t['bytes_per_launch'])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='latents per GPU')
    ap.add_argument('--cm', type=int, default=1, help='channel_multiplier (1 = voxceleb-256, the headline config)')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--layers', action='store_true', help='also print the per-layer conv table to stderr')
    ap.add_argument('--precision', choices=('fp16x3', 'fp32', 'bf16x3'), default='fp16x3',
                    help="arithmetic of the 3x3 convs: fp16x3 (default) = fp32 operands as fp16 hi+lo (22 mantissa bits), "
                         "hi*hi+hi*lo+lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation -- measured as accurate as "
                         "the fp32 kernels; fp32 = fp32 MFMA / Winograd kernels; bf16x3 = bf16 hi+lo (fp32 range, ~1e-4)")
    ap.add_argument('--no-alt', action='store_true', help='skip the extra leg that times the other arithmetic')
    args = ap.parse_args()

    rank, local_rank, world = D.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)

    # ---- weights: rank 0 generates, everyone receives one flat RCCL broadcast
    G, template = generator_state_template(args.size, args.cm)
    if rank == 0:
        G.load_state_dict(S.synthetic_state_dict(template, seed=SEED))
    G = G.eval().to(dev)
    bcast_bytes = D.broadcast_state(G, src=0)

    # ---- this rank's shard of the global latent batch (contiguous split), resident in HBM
    B = args.batch
    lo, hi = D.shard_range(B * world, rank, world)
    w = S.synthetic_latents(SEED, B * world, n_latent=G.n_latent, key='bench.w')[lo:hi].contiguous().to(dev)

    def step():
        img, _ = G([w], input_is_latent=True)
        return img

    F_.set_precision(args.precision)
    with torch.no_grad():
        for _ in range(max(args.warmup, 1)):
            img = step()
        assert img.shape == (hi - lo, 3, args.size, args.size) and bool(torch.isfinite(img).all())
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        D.barrier()
        elapsed = time.perf_counter() - t0
        elapsed = D.max_over_ranks(elapsed, dev)

        # ---- roofline leg: HIP events around every MFMA conv launch, on the launch stream
        F_.CONV_TIMING = []
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        rec, F_.CONV_TIMING = F_.CONV_TIMING, None

        # ---- extra leg: the same steps in the other arithmetic (fp32 MFMA kernels <-> fp16x3) + the deviation between
        # the two sets of images
        alt = None
        alt_mode = 'fp32' if args.precision != 'fp32' else 'fp16x3'
        if not args.no_alt:
            exact = step()
            F_.set_precision(alt_mode)
            try:
                for _ in range(max(args.warmup, 1)):
                    fast = step()
                D.barrier()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                D.barrier()
                alt_elapsed = D.max_over_ranks(time.perf_counter() - t1, dev)
            finally:
                F_.set_precision(args.precision)
            alt = {'precision': alt_mode, 'value': round(B * world * args.steps / alt_elapsed, 2), 'unit': 'frames/s',
                   'ms_per_step': round(alt_elapsed / args.steps * 1e3, 3),
                   'max_abs_between_the_two_paths': float((fast - exact).abs().max()),
                   'note': 'same workload with --precision %s (SGDFR_PRECISION); tests/test_gpu_split.py holds fp16x3 and the '
                           'fp32 kernels to the same bound vs the fp64 oracle (measured 7.7e-6 / 9.5e-6 on 256x256 images)'
                           % alt_mode}
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

    if rank != 0:
        return
    frames = (B * world) * args.steps
    out = {
        'metric': 'reenacted frames/sec @256x256',
        'value': round(frames / elapsed, 2),
        'unit': 'frames/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': {'fp32': 'f32', 'fp16x3': 'f32 (operands as fp16 hi+lo = 22 mantissa bits, 3 MFMA products, f32 accumulate)',
                  'bf16x3': 'f32 (operands as bf16 hi+lo = 16 mantissa bits, 3 MFMA products, f32 accumulate)'}[args.precision],
        'data': 'synthetic',
        'config': {'workload': '%dxMI355X HIP synthesis-only: Generator(%d,512,8,cm=%d), random w+ [%d,14,512] per GPU, '
                               'fixed noise, psi=1' % (world, args.size, args.cm, B),
                   'per_gpu_batch': B, 'global_batch': B * world, 'resolution': args.size,
                   'channel_multiplier': args.cm, 'parallelism': 'batch-sharded x%d, no data-path collective' % world,
                   'weight_broadcast_bytes': bcast_bytes},
        'roofline': {'bound': 'mfma', 'kernel': 'wino_mfma_kernel (5 plain 3x3 layers) + modconv_mfma_kernel (2 plain, 6 transposed): 13 conv launches/forward',
                     'achieved': round(achieved, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': pmc_traffic(args, B),
                     'avg_launch_us': round(conv_s / n_launch * 1e6, 2),
                     'conv_ms_per_step': round(conv_s / args.steps * 1e3, 3),
                     'alg_gflop_per_frame': round(conv_flops / (B * args.steps) / 1e9, 3)},
    }
    if args.precision == 'fp16x3':
        # operand pairs the fp16-split kernels had to clamp (|x*s| > 1.04e6) during this whole run: 0 = the fp32-grade claim holds
        out['fp16_saturated_pairs'] = F_.split_saturation_count(reset=False)
    if alt is not None:
        out['alt_arithmetic'] = alt
    if args.precision != 'fp32':
        # SURVEY.md §8d rule for split arithmetic: ALGORITHMIC fp32 FLOPs over time, denominator stated
        out['roofline'].update({'kernel': 'split_mfma_kernel (7 plain + 6 transposed 3x3 conv launches/forward, %s)' % args.precision,
                                'peak': round(SPLIT_PEAK_TFLOPS, 1), 'frac': round(achieved / SPLIT_PEAK_TFLOPS, 4),
                                'peak_note': 'dense 16-bit MFMA peak 2500 TFLOP/s / 3 products per fp32 product; the same '
                                             'achieved figure is %.2fx the 157.3 TFLOP/s fp32-MFMA peak'
                                             % (achieved / FP32_MFMA_PEAK_TFLOPS)})
    if args.layers:
        for desc, (sec, fl, n) in per_layer.items():
            sys.stderr.write('%-28s %8.1f us/launch %7.1f TFLOP/s\n' % (desc, sec / n * 1e6, fl / sec / 1e12))
    if world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(args.size, args.cm)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()

"""profiles/<out>_trainer_*: the generator legs of the direction-learning step (scripts/train_step_bench.py, B=16: one no-grad forward of 2B rows
+ grad forward + backward to A) from rocprofv3 --kernel-trace --stats (gpurun_out/prof_<tag>_train) and one SQ PMC pass
(gpurun_out/pmc_<tag>_train_sq), aggregated per kernel:  python scripts/summarize_trainer.py r4d r04_d"""
import collections, csv, shutil, sys

tag, out = sys.argv[1], sys.argv[2]
shutil.copy('gpurun_out/prof_%s_train/train_kernel_stats.csv' % tag, 'profiles/%s_trainer_step_kernel_stats.csv' % out)
rows = list(csv.DictReader(open('gpurun_out/pmc_%s_train_sq/pmc_counter_collection.csv' % tag)))
kt = {int(r['Dispatch_Id']): int(r['End_Timestamp']) - int(r['Start_Timestamp'])
      for r in csv.DictReader(open('gpurun_out/pmc_%s_train_sq/pmc_kernel_trace.csv' % tag))}
d = collections.OrderedDict()
for r in rows:
    d.setdefault((int(r['Dispatch_Id']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('sgdfr::', ''), int(r['Grid_Size'])), {})[
        r['Counter_Name']] = float(r['Counter_Value'])
agg = collections.OrderedDict()
for (disp, name, grid), c in d.items():
    if not any(t in name for t in ('split_mfma', 'wsplit_kernel', 'wswide', 'modconv_mfma', 'wino', 'wgrad', 'act_grad', 'grad_join', 'scale_reduce', 'blur', 'torgb', 'styles_batched', 'absmax', 'split_range', 'splitk_reduce', 'linear_skinny')):
        continue
    a = agg.setdefault((name, grid), [0, 0.0, 0.0, 0.0, 0.0, 0.0])
    dur = kt.get(disp, 0)
    gui = c.get('GRBM_GUI_ACTIVE', 0) / 8
    a[0] += 1
    a[1] += dur
    a[2] += c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
    a[3] += gui * 1024
    a[4] += c.get('SQ_WAIT_ANY', 0)
    a[5] += c.get('SQ_WAVE_CYCLES', 0)
tot = sum(a[1] for a in agg.values())
L = ['# Direction-learning step, generator legs (%s)\n' % out,
     '`python scripts/train_step_bench.py 16` (B=16, 256x256, cm=1, G frozen: source + target rows in one no-grad forward of 32 + grad forward + backward to A through autograd.SynthesisFn) under',
     '`rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY',
     'SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE`; kernel stats of the un-instrumented run: `profiles/%s_trainer_step_kernel_stats.csv`.' % out,
     'Launches aggregated per (kernel, grid) over the whole run (13 steps), sorted by total time; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES /',
     '(GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).  `<1, ...>` / `<2, ...>` of split_mfma_kernel are the transposed conv and its adjoint (DOWN3),',
     '`..., false>` instantiations take fp32 input (autograd forward and the plain layers\' dL/dx), `true` the pre-split chain.\n',
     '| kernel | grid | launches | avg us | share of the listed time | MFMA busy % | wait_any / wave_cycles |', '|---|---|---|---|---|---|---|']
for (name, grid), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    L.append('| `%s` | %d | %d | %.1f | %.1f %% | %.1f | %.2f |' % (name, grid, a[0], a[1] / a[0] / 1e3, 100 * a[1] / tot,
                                                                100 * a[2] / a[3] if a[3] else 0, a[4] / a[5] if a[5] else 0))
open('profiles/%s_trainer_pmc.md' % out, 'w').write('\n'.join(L) + '\n')
print('\n'.join(L[:40]))

#!/bin/bash
# rocprofv3 evidence of one state of the round, on the GPU box:  bash scripts/profile_round.sh r1h
# (timing run with --kernel-trace --stats, then one PMC pass per counter set -- never combined with other traces), then
# locally: python scripts/summarize_pmc.py r1h r01_h fp16x3
set -u
tag=$1
mkdir -p gpurun_out
python -c "import bench; print(bench.kernel_source_hash())" > gpurun_out/source_hash_$tag.txt   # what the counters were measured on
B="python bench.py --no-cpu-baseline --no-alt --no-other-configs --no-oracle-delta --sustain 0 --streams 1"   # (kernels of one forward at a time: the durations bench.py's roofline pass sees)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- $B --steps 10 --warmup 3 > gpurun_out/bench_prof_$tag.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d gpurun_out/pmc_${tag}_$n -o pmc -- $B --steps 2 --warmup 1 > gpurun_out/pmc_${tag}_$n.log 2>&1
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  -d gpurun_out/pmc_${tag}_sq -o pmc -- $B --steps 2 --warmup 1 > gpurun_out/pmc_${tag}_sq.log 2>&1
grep -h metric gpurun_out/bench_prof_$tag.log | cut -c1-200
python bench.py > gpurun_out/bench_line_$tag.json 2> gpurun_out/bench_line_$tag.err; tail -1 gpurun_out/bench_line_$tag.json | cut -c1-200

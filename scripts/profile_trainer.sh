#!/bin/bash
# rocprofv3 evidence of the trainer / PTI legs on the GPU box:  bash scripts/profile_trainer.sh r6a
#   kernel stats of scripts/train_step_bench.py 16 (B=16 direction-learning step, generator legs) and scripts/pti_step_bench.py 1 graph,
#   then one SQ PMC pass of the trainer step (never combined with other traces).  Locally: python scripts/summarize_trainer.py r6a r06_a
set -u
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_train -o train -- python scripts/train_step_bench.py 16 > gpurun_out/train_prof_$tag.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_pti -o pti -- python scripts/pti_step_bench.py 1 graph fused > gpurun_out/pti_prof_$tag.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  -d gpurun_out/pmc_${tag}_train_sq -o pmc -- python scripts/train_step_bench.py 16 > gpurun_out/pmc_${tag}_train_sq.log 2>&1
tail -2 gpurun_out/train_prof_$tag.log gpurun_out/pti_prof_$tag.log
find gpurun_out/prof_${tag}_train gpurun_out/prof_${tag}_pti gpurun_out/pmc_${tag}_train_sq -name '*.csv' | head -20

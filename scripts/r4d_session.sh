python -m pytest tests/test_gpu_split.py tests/test_gpu_wsplit.py -q -k "two_generators or blur_winograd" 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r4d_train -o train -- python scripts/train_step_bench.py 16 > gpurun_out/train_prof_r4d.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d gpurun_out/pmc_r4d_train_sq -o pmc -- python scripts/train_step_bench.py 16 > gpurun_out/train_pmc_r4d.log 2>&1
grep "train-shaped" gpurun_out/train_prof_r4d.log gpurun_out/train_pmc_r4d.log
python scripts/train_step_bench.py 16; python scripts/train_step_bench.py 64
AB_GREP="mode1 512->512 @16|blur|mode[01] 512->512 @(4|8)" bash scripts/env_ab.sh "SGDFR_X=0" "SGDFR_SPLIT_UP_NARROW=1" "SGDFR_BLUR_SEGMENTS=1" "SGDFR_BLUR_SEGMENTS=2" "SGDFR_BLUR_SEGMENTS=4" "SGDFR_SPLIT_KTARGET=512" "SGDFR_SPLIT_DESYNC=100" "SGDFR_X=0"

for cfg in "SGDFR_UPPP=0" "SGDFR_UPPP=1"; do
  for b in 16 32; do
    echo "=== $cfg B=$b"
    env $cfg python bench.py --batch $b --no-cpu-baseline --no-alt --no-other-configs --no-oracle-delta --sustain 0 --layers --steps 30 --warmup 10 2> gpurun_out/ab_layers.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',d['value'],'single',d['single_stream']['value'],'conv_ms',d['roofline']['conv_ms_per_step'])"
    grep -E "mode1 (512->256|256|128)|up-pp (512->256|256|128)" gpurun_out/ab_layers.txt
  done
done

# same-box A/B of the fused up-conv + blur level(s):  bash scripts/upfir_ab.sh "SGDFR_UPFIR=0" "SGDFR_UPFIR=1" ...
B="python bench.py --no-cpu-baseline --no-alt --no-other-configs --no-oracle-delta --sustain 0 --layers --steps 30 --warmup 10"
for cfg in "$@"; do
  echo "=== $cfg"
  env $cfg $B 2> gpurun_out/ab_layers.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',d['value'],'unverified',d['unverified']['value'],'single',d['single_stream']['value'],'conv_ms',d['roofline']['conv_ms_per_step'])
"
  grep -E "upfir|mode1 (256|128)|blur (128|64) ch" gpurun_out/ab_layers.txt
done

# same-box A/B of environment switches on the headline bench:  bash scripts/env_ab.sh "SGDFR_X=0" "SGDFR_X=1" ...
# (prints value / unverified / single-stream frames/s, conv ms, and the per-launch rows matching $AB_GREP)
B="python bench.py --no-cpu-baseline --no-alt --no-other-configs --no-oracle-delta --sustain 0 --layers --steps 30 --warmup 10"
for cfg in "$@"; do
  echo "=== $cfg"
  env $cfg $B 2> gpurun_out/ab_layers.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',d['value'],'unverified',d['unverified']['value'],'single',d['single_stream']['value'],'conv_ms',d['roofline']['conv_ms_per_step'])
"
  grep -E "${AB_GREP:-mode1|blur}" gpurun_out/ab_layers.txt
done

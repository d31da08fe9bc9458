#!/bin/bash
# Is the split conv's K loop bound by its LDS fragment reads?  Rebuilds split.hip with -DSGDFR_PROBE_NOFETCH (one fragment
# fetch per kernel row instead of three: a third of the ds_read_b128 traffic, wrong results) and prints the per-layer times
# next to the real build's.   bash scripts/lds_probe.sh   (on the GPU box; restores the real build afterwards)
set -u
python bench.py --layers --no-cpu-baseline --no-alt --steps 10 2>&1 >/dev/null | grep "split mode0" > gpurun_out/lds_probe_real.txt
touch stylegan_directions_face_reenactment_amd/csrc/split.hip
python -c "from stylegan_directions_face_reenactment_amd import build_native as b; b.build(extra=['-DSGDFR_PROBE_NOFETCH'])"
python bench.py --layers --no-cpu-baseline --no-alt --steps 10 2>&1 >/dev/null | grep "split mode0" > gpurun_out/lds_probe_nofetch.txt
touch stylegan_directions_face_reenactment_amd/csrc/split.hip
python -c "from stylegan_directions_face_reenactment_amd import build_native as b; b.build()"
paste gpurun_out/lds_probe_real.txt gpurun_out/lds_probe_nofetch.txt | awk '{printf "%-8s %-12s %-10s real %8s us   nofetch %8s us\n", $2, $3, $4, $5, $13}'

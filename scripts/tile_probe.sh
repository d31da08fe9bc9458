#!/bin/bash
# Rebuilds split.hip with the probe switch and times the hot layers under SGDFR_SPLIT_DBG combinations
# (1 one channel block, 2 no epilogue, 4 no K loop, 8 no activation DMA, 16 no weight DMA); restores the real build.
touch stylegan_directions_face_reenactment_amd/csrc/split.hip
python -c "from stylegan_directions_face_reenactment_amd import build_native as b; b.build(extra=['-DSGDFR_SPLIT_PROBE'])"
for d in 0 8 16 24 2 26 4; do SGDFR_SPLIT_DBG=$d python scripts/tile_probe.py 2>&1 | grep dbg; done
touch stylegan_directions_face_reenactment_amd/csrc/split.hip
python -c "from stylegan_directions_face_reenactment_amd import build_native as b; b.build()"

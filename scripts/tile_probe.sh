#!/bin/bash
# Times the hot layers under SGDFR_SPLIT_DBG combinations on the probe build (scripts/build_probe.py -> build/libsgdfr_hip_probe.so;
# 1 one channel block, 2 no epilogue, 4 no K loop, 8 no activation DMA, 16 no weight DMA).  The product library is not touched.
export SGDFR_LIB=build/libsgdfr_hip_probe.so SGDFR_ALLOW_LIB_OVERRIDE=1
for d in ${DBGS:-0 8 16 24 2 26 4}; do SGDFR_SPLIT_DBG=$d python scripts/tile_probe.py 2>&1 | grep dbg; done

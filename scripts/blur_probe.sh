#!/bin/bash
# What bounds the split blur (planes -> next conv's input)?  Times the four big levels with the real kernel, without its
# stores (-DSGDFR_BLUR_PROBE=1) and without its plane loads (=2); restores the real build.  bash scripts/blur_probe.sh
for probe in 0 1 2; do
  touch stylegan_directions_face_reenactment_amd/csrc/upfirdn2d.hip
  python -c "from stylegan_directions_face_reenactment_amd import build_native as b; b.build(extra=['-DSGDFR_BLUR_PROBE=$probe'])"
  echo -n "probe $probe: "; python scripts/blur_time.py 2>&1 | tail -1
  for sg in 1 2; do echo -n "probe $probe segments=$sg: "; SGDFR_BLUR_SEGMENTS=$sg python scripts/blur_time.py 2>&1 | tail -1; done
done
touch stylegan_directions_face_reenactment_amd/csrc/upfirdn2d.hip
python -c "from stylegan_directions_face_reenactment_amd import build_native as b; b.build()"

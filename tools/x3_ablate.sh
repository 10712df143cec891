#!/bin/bash
# ablation timings of conv3d_x3_kernel: make -C instanceseg-without-voxelwise-labeling_amd/csrc x3_ablate (builds libm3d_x3abl*.so, X3_EXP bits in
# conv3d_x3.hip), then on the GPU box: bash tools/x3_ablate.sh
cd /root/repo
echo "== release"; python tools/bench_x3.py 2>/dev/null | head -2
for e in 1 2 4 8 16 6 14 30; do
  echo "== X3_EXP=$e"; M3D_LIB_PATH=/root/repo/instanceseg-without-voxelwise-labeling_amd/csrc/libm3d_x3abl$e.so python tools/bench_x3.py 2>/dev/null | head -2
done

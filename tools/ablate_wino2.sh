#!/bin/bash
# usage: [SG=r,v,w,m] [TAG=name] ablate_wino2.sh [EXP codes]   (SG: interleave pattern per MFMA slot; EXP: ablation bits)
# Timing-only ablation builds of the 2-D Winograd kernel (results are wrong by construction): build/abl/libm3d_expN.so
set -e
cd "$(dirname "$0")/../instanceseg-without-voxelwise-labeling_amd/csrc"
mkdir -p ../../build/abl
OBJS="m3d_core.o roi_align3d.o box_ops.o conv3d.o otsu2d.o pool_bn.o prm.o binarize.o cc3d.o conv3d_wgrad.o conv3d_wino.o conv3d_stem_wino.o"
for e in ${@:-1 2 3 4 7}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fvisibility=hidden -I../../include -DM3D_EXP=$((e % 10)) ${SG:+-DM3D_SG=$SG} -c conv3d_wino2.hip -o ../../build/abl/wino2_${TAG:-}$e.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../../build/abl/wino2_${TAG:-}$e.o -o ../../build/abl/libm3d_${TAG:-exp}$e.so
done

#!/bin/bash
cd /root/repo
O=gpurun_out/r3_run9; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "stem" > $O/pytest_stem.txt 2>&1; echo "pytest stem rc=$?"
tail -5 $O/pytest_stem.txt | cut -c1-200
for v in 8 4; do
echo "== stamps tune_stem $v"
ROWS=$((2*v)) M3D_TUNE_STEM=$v M3D_LIB_PATH=/root/repo/instanceseg-without-voxelwise-labeling_amd/csrc/libm3d_w2stamps.so timeout -k 10 200 python tools/stem_stamps.py 2>&1 | grep -v amdgpu.ids
done
for rep in 1 2; do
for v in 1 4 8; do
  echo "== tune_stem $v"
  M3D_TUNE_STEM=$v LAYERS=conv1a BATCH=4 timeout -k 10 200 python tools/bench_layers.py 128 20 2>&1 | grep conv1a | sed -E 's/.*F\(2,5\)x/F(2,5)x/'
  M3D_TUNE_STEM=$v LAYERS=conv1a FUSED=0 BATCH=4 timeout -k 10 200 python tools/bench_layers.py 128 20 2>&1 | grep conv1a | sed -E 's/.*F\(2,5\)x/nopool F(2,5)x/'
done
done

#!/bin/bash
cd /root/repo
O=gpurun_out/r3_run4; mkdir -p $O
CS=$PWD/instanceseg-without-voxelwise-labeling_amd/csrc
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv3d" > $O/pytest_conv.txt 2>&1; echo "pytest conv rc=$?"
tail -5 $O/pytest_conv.txt
for rep in 1 2; do
for v in r2 new; do
  if [ $v = r2 ]; then export M3D_LIB_PATH=$CS/libm3d_r2.so; else unset M3D_LIB_PATH; fi
  BATCH=4 timeout -k 10 200 python tools/bench_layers.py 128 20 2>&1 | grep -E "conv2|conv3|conv4|rpn_conv" | sed -E 's/.*(conv[0-9a-z+_]+|rpn_conv) .*F\(2x2,3x3\)/\1/' > $O/layers_$v.txt
  echo "== $v"; cat $O/layers_$v.txt
done
done
unset M3D_LIB_PATH
echo "== stamps" > $O/stamps.txt
M3D_LIB_PATH=$CS/libm3d_w2stamps.so timeout -k 10 120 python tools/w2_stamps.py conv2b conv2a conv3b conv4b 2>&1 | grep -E "batch|median" >> $O/stamps.txt
cat $O/stamps.txt

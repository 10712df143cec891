#!/bin/bash
# family 5 (wide F(2x4)) against the default family: conv parity tests + layer timings
cd /root/repo
O=gpurun_out/r3_fam5; mkdir -p $O
M3D_TUNE_WINO2=599 timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv3d" > $O/pytest_conv_599.txt 2>&1; echo "pytest conv (family 5) rc=$?"
tail -6 $O/pytest_conv_599.txt | cut -c1-220
for rep in 1 2; do
for v in 499 599; do
  echo "== family $v"
  M3D_TUNE_WINO2=$v BATCH=4 timeout -k 10 200 python tools/bench_layers.py 128 20 2>&1 | grep -E "conv[2-4]|rpn_conv" | sed -E 's/.*(conv[0-9a-z+_]+|rpn_conv) .*(F\(2x[24],3x3\))/\1 \2/' | cut -c1-120
done
done

#!/bin/bash
cd /root/repo
O=gpurun_out/r3_run6; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv3d" > $O/pytest_conv.txt 2>&1; echo "pytest conv rc=$?"
tail -12 $O/pytest_conv.txt | cut -c1-200
for rep in 1 2; do
for v in 299 499; do
  M3D_TUNE_WINO2=$v BATCH=4 timeout -k 10 200 python tools/bench_layers.py 128 20 2>&1 | grep -E "conv2|conv3|conv4|rpn_conv" | sed -E 's/.*(conv[0-9a-z+_]+|rpn_conv) .*F\(2x2,3x3\)/\1/' > $O/layers_$v.txt
  echo "== family $v"; cat $O/layers_$v.txt
done
done

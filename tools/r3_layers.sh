#!/bin/bash
# conv parity tests + layer timings of the default family (two repeats)
cd /root/repo
O=gpurun_out/r3_layers; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv3d" > $O/pytest_conv.txt 2>&1; echo "pytest conv rc=$?"
tail -4 $O/pytest_conv.txt | cut -c1-220
for rep in 1 2; do
  BATCH=4 timeout -k 10 200 python tools/bench_layers.py 128 20 2>&1 | grep -E "conv|rpn_conv|TOTAL" | sed -E 's/.*(conv[0-9a-z+_]+|rpn_conv) .*(F\(2x[24],3x3\)|F\(2,5\)x)/\1 \2/' | cut -c1-150
done

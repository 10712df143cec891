#!/bin/bash
cd /root/repo
O=gpurun_out/r3_run10; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "stem" > $O/pytest_stem.txt 2>&1; echo "pytest stem rc=$?"
tail -5 $O/pytest_stem.txt | cut -c1-200
for rep in 1 2; do
for v in 1 8; do
  echo "== tune_stem $v"
  M3D_TUNE_STEM=$v LAYERS=conv1a BATCH=4 timeout -k 10 200 python tools/bench_layers.py 128 20 2>&1 | grep conv1a | sed -E 's/.*F\(2,5\)x/F(2,5)x/'
  M3D_TUNE_STEM=$v LAYERS=conv1a FUSED=0 BATCH=4 timeout -k 10 200 python tools/bench_layers.py 128 20 2>&1 | grep conv1a | sed -E 's/.*F\(2,5\)x/nopool F(2,5)x/'
done
done

#!/bin/bash
# round-3 GPU call 1: correctness of the re-scheduled eta-split Winograd kernel, A/B against the round-2 build, stamps, bench
cd /root/repo
O=gpurun_out/r3_run1; mkdir -p $O
CS=instanceseg-without-voxelwise-labeling_amd/csrc
set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv3d" > $O/pytest_conv.txt 2>&1; echo "pytest conv rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_conv.txt
for rep in 1 2; do
  for v in r2 new; do
    if [ $v = r2 ]; then export M3D_LIB_PATH=$PWD/$CS/libm3d_r2.so; else unset M3D_LIB_PATH; fi
    BATCH=4 timeout -k 10 200 python tools/bench_layers.py 128 20 > $O/layers_${v}_$rep.txt 2>&1 || echo "bench_layers $v failed"
  done
done
unset M3D_LIB_PATH
grep -h -E "conv2a|conv2b|conv3a|conv3b|conv4a|conv4b|rpn_conv|TOTAL" $O/layers_r2_2.txt | cut -c1-40,95-200 > $O/ab.txt; echo ---- >> $O/ab.txt
grep -h -E "conv2a|conv2b|conv3a|conv3b|conv4a|conv4b|rpn_conv|TOTAL" $O/layers_new_2.txt | cut -c1-40,95-200 >> $O/ab.txt
cat $O/ab.txt
M3D_LIB_PATH=$PWD/$CS/libm3d_w2stamps.so timeout -k 10 200 python tools/w2_stamps.py conv2b conv2a conv3b conv4b > $O/stamps.txt 2>&1; cat $O/stamps.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json

#!/bin/bash
cd /root/repo
O=gpurun_out/r3_run5; mkdir -p $O
CS=$PWD/instanceseg-without-voxelwise-labeling_amd/csrc
: > $O/stamps.txt
for e in stamps abl1 abl2 abl3 abl64 abl128 abl256 abl320 abl512 abl59; do
  echo "== $e" >> $O/stamps.txt
  M3D_LIB_PATH=$CS/libm3d_w2$e.so timeout -k 10 120 python tools/w2_stamps.py conv2b conv3b 2>&1 | grep -E "batch|median" >> $O/stamps.txt
done
cat $O/stamps.txt

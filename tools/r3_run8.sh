#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r3_run8; mkdir -p $O
for v in 0 1; do
  export M3D_TUNE_STEM=$v
  bash tools/pmc_mfma_busy.sh > $O/mfma_busy_stem$v.txt 2>&1
  echo "== tune_stem $v"; grep -E "stem|kernel " $O/mfma_busy_stem$v.txt | cut -c1-260
done

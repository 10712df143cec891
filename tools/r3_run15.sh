#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r3_run15; mkdir -p $O
L=/root/repo/instanceseg-without-voxelwise-labeling_amd/csrc
for rep in 1 2; do
for lib in libm3d_w2old.so libm3d.so; do
  for wl in prm prm-nuclei; do
    M3D_LIB_PATH=$L/$lib timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 > $O/x.json
    python3 - $O/x.json $lib $wl <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("%-18s %-11s ms/tile %.3f" % (sys.argv[2], sys.argv[3], d["ms_per_step"]))
PY
  done
done
done

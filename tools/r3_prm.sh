#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r3_prm; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_prm.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest_prm.txt 2>&1; echo "pytest prm rc=$?"
tail -3 $O/pytest_prm.txt | cut -c1-200
for i in 1 2; do
for wl in prm prm-nuclei; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 > $O/x.json
  python3 - $O/x.json $wl <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("%-11s ms/tile %.3f" % (sys.argv[2], d["ms_per_step"]))
PY
done
done

#!/bin/bash
# full GPU suite + default bench (what the driver runs at round end)
cd /root/repo
O=gpurun_out/r3_full; mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?"
tail -4 $O/pytest_gpu.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_full/bench.json"))
print("ms_per_step", d["ms_per_step"], "backbone_ms_per_volume", d["config"].get("backbone_ms_per_volume"))
print(d["config"]["kernel_ms_per_launch"])
PY

#!/bin/bash
# full GPU suite + default bench (what the driver runs at round end) + the same bench on the round-2/3 F(2x2) family for comparison
cd /root/repo
O=gpurun_out/r3_full; mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?"
tail -4 $O/pytest_gpu.txt
for v in default 299 default 299; do
  if [ $v = default ]; then unset M3D_TUNE_WINO2; else export M3D_TUNE_WINO2=$v; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?"
  python - $v <<'PY'
import json,sys
d=json.load(open("gpurun_out/r3_full/bench_%s.json"%sys.argv[1]))
print("  ms_per_step %.3f backbone_ms_per_volume %.3f conv-family frac %.3f alg %.1f TF" % (d["ms_per_step"], d["config"].get("backbone_ms_per_volume"), d["roofline"]["frac"], d["roofline"]["algorithmic_tflops"]))
print("  ", d["config"]["kernel_ms_per_launch"])
PY
done
timeout -k 10 300 python bench.py --workload prm --no-cpu-baseline > $O/prm_soma.json 2> $O/prm_soma.err; echo "prm soma rc=$?"
timeout -k 10 300 python bench.py --workload prm-nuclei --no-cpu-baseline > $O/prm_nuclei.json 2> $O/prm_nuclei.err; echo "prm nuclei rc=$?"
python - <<'PY'
import json
for f in ("prm_soma", "prm_nuclei"):
    try:
        d = json.loads(open("gpurun_out/r3_full/%s.json" % f).read().strip().splitlines()[-1])
        print("==", f, "ms/step %.3f" % d["ms_per_step"], {k: round(v, 3) for k, v in d.get("config", {}).get("stage_ms", {}).items()} if isinstance(d.get("config", {}).get("stage_ms"), dict) else "")
    except Exception as e:
        print(f, "FAILED", e)
PY

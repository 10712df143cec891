#!/bin/bash
cd /root/repo
O=gpurun_out/r3_bench; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_compat.py tests/test_gpu_fullsize.py -x -q -m gpu -k "fused or linear_interception or six_batches or begin_finish" > $O/pytest_new.txt 2>&1; echo "pytest new rc=$?"; tail -3 $O/pytest_new.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/detect.json 2> $O/detect.err; echo "detect rc=$?"
timeout -k 10 300 python bench.py --workload prm > $O/prm_soma.json 2> $O/prm_soma.err; echo "prm rc=$?"
timeout -k 10 300 python bench.py --stress-rois --steps 10 --warmup 3 > $O/stress.json 2> $O/stress.err; echo "stress rc=$?"
timeout -k 10 400 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 > $O/g2.json 2> $O/g2.err; echo "g2 rc=$?"
python - <<'PY'
import json
for f in ("detect","prm_soma","stress","g2"):
    try:
        d=json.loads(open("gpurun_out/r3_bench/%s.json"%f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "FAILED", e); print(open("gpurun_out/r3_bench/%s.err"%f).read()[-1500:]); continue
    print("==", f, "ms/step %.3f value %.3e" % (d["ms_per_step"], d["value"]))
    r=d.get("roofline") or {}
    print("   roofline: achieved %.1f frac %.3f alg %.1f" % (r.get("achieved",0), r.get("frac",0), r.get("algorithmic_tflops",0)))
    for k in ("pipelined","e2e_host_to_host","sustained","without_exchange","single_gpu_same_batch","exchange","configs4_shape","cpu_baseline"):
        if k in d: print("   ", k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in d[k].items() if a not in ("what","includes","sample","clock")})
    if f=="detect":
        print("   layers:", {k:(round(v["frac"],3), round(v["kernel_ms"],3)) for k,v in d["rooflines"].items()})
    if "config" in d: print("   rois/vol", d["config"].get("rois_per_volume"), "kern", d["config"].get("kernel_ms_per_launch"))
PY

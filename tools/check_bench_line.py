#!/usr/bin/env python3
"""Check a bench.py stdout capture the way the driver reads it: the LAST line must be JSON, under 8000 bytes, and carry the
contract's keys + roofline + cpu_baseline.   python tools/check_bench_line.py gpurun_out/bench_default.log"""
import json
import sys

last = open(sys.argv[1]).read().rstrip("\n").splitlines()[-1]
d = json.loads(last)
assert len(last) < 8000, len(last)
for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
    assert k in d, k
assert "workload" in d["config"]
assert not d.get("accounting_errors"), d["accounting_errors"]
if d["n_gpus"] == 1 and not d.get("dry"):
    assert d.get("roofline") and d.get("cpu_baseline"), "roofline / cpu_baseline missing"
    for sub in ("configs1_backbone", "stress_rois", "configs3_prm_soma", "prm_nuclei_tile", "volume_pipeline"):
        if sub in d:
            assert "error" not in d[sub], (sub, d[sub])
            assert d[sub].get("roofline") is not None, sub + ": roofline null"
print("bench line ok: %d bytes, value %.4g %s, %.3f ms/step, roofline.frac %s" %
      (len(last), d["value"], d["unit"], d["ms_per_step"], (d.get("roofline") or {}).get("frac")))

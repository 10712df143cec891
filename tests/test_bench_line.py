"""bench.py's final stdout line must stay parseable by the driver, which keeps ~9 KB of stdout tail (round 4's 25.9 KB line left
BENCH_r04.parsed null).  CPU only: the compaction on a committed full record, on a bloated synthetic one, and on the --dry path."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config")


def test_compact_line_of_a_real_full_record_is_short_and_keeps_what_is_judged():
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_default_line_run2.json")))      # 25.9 KB: the line that did not parse
    assert len(json.dumps(full)) > 20000
    line = b.compact_line(full, "gpurun_out/bench_full_detect_n1.json")
    assert len(line) < b.LINE_MAX <= 8000 and "\n" not in line
    d = json.loads(line)
    for k in CONTRACT:
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"] and d["steps"] == 20 and d["warmup"] == 5
    assert d["config"]["workload"].startswith("detection-mode infer_simple") and d["config"]["volumes_per_step"] == 4
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_algorithmic", "kernel_ms", "traffic"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5
    assert set(d["cpu_baseline"]) == {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    for sub in ("configs1_backbone", "stress_rois", "configs3_prm_soma", "prm_nuclei_tile", "volume_pipeline"):
        r = d[sub]
        assert r["value"] > 0 and r["ms_per_step"] > 0 and "workload" in r["config"]
        assert "roofline" in r and (r["roofline"] is None or "frac" in r["roofline"])
    assert "rooflines" not in d and "step_ms_timed" not in json.dumps(d) and "frac_definition" not in line


def test_compact_line_survives_a_bloated_record_and_non_finite_numbers():
    b = _bench()
    blob = "x" * 5000
    rec = {"metric": b.METRIC, "value": 1.0, "unit": "voxels/s", "n_gpus": 8, "steps": 3, "warmup": 1, "ms_per_step": float("nan"),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": blob, "volumes_per_step": 64, "backend": "nccl", "junk": [1] * 4000},
           "roofline": {"bound": "mfma", "kernel": blob, "achieved": float("inf"), "peak": 157.3, "unit": "TFLOP/s", "frac": 0.5,
                        "frac_definition": blob, "traffic": None},
           "cpu_baseline": {"value": 2.0, "unit": "voxels/s", "cores": 16, "kind": "port", "sample": blob, "more": blob},
           "exchange": {"us": 31.5, "ranks": 8, "backend": "nccl", "what": blob},
           "without_exchange": {"value": 3.0, "ms_per_step": 4.0, "what": blob}}
    for k in ("configs1_backbone", "stress_rois", "configs3_prm_soma", "prm_nuclei_tile", "volume_pipeline"):
        rec[k] = {"value": 5.0, "ms_per_step": 6.0, "roofline": {"bound": "hbm", "kernel": blob, "frac": 0.1, "traffic": 7.0, "note": blob},
                  "cpu_baseline": {"value": 1.0, "cores": 16, "kind": "port", "sample": blob}, "config": {"workload": blob, "phase_ms": {"a": 1.0}},
                  "volumes": {"nuclei": {"value": 1.0, "seconds_per_volume": 2.0, "peaks": 3, "cpu_baseline": {"sample": blob}}}}
    rec["prm_nuclei_tile"] = {"error": blob}
    line = b.compact_line(rec)
    assert len(line) < 8000
    d = json.loads(line)                                   # strict JSON: no NaN / Infinity tokens
    assert "NaN" not in line and "Infinity" not in line
    assert d["ms_per_step"] is None and d["roofline"]["achieved"] is None
    assert d["exchange"] == {"us": 31.5, "ranks": 8, "backend": "nccl"} and d["n_gpus"] == 8
    assert d["without_exchange"] == {"value": 3.0, "ms_per_step": 4.0}
    assert len(d["prm_nuclei_tile"]["error"]) <= 200


def test_dry_run_prints_one_short_json_line_last():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert len(last) < 8000
    d = json.loads(last)
    for k in CONTRACT:
        assert k in d, k
    assert d["dry"] is True and d["steps"] == 2 and d["warmup"] == 1 and d["n_gpus"] == 1


def test_a_hardware_fraction_above_one_is_an_accounting_error():
    """Round 5's soma volume counted peaks the engine had skipped (frac 1.89): the checker must flag that record, pass the newest
    committed one, and emit() must carry the finding on the line."""
    b = _bench()
    r05 = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default_full_record.json")))
    bad = b.hardware_fracs_above_one(r05)
    assert any("volume_pipeline" in p and "soma" in p and v > 1.5 for p, v in bad), bad
    assert b.hardware_fracs_above_one({"roofline": {"frac": 0.63, "frac_algorithmic": 1.76}, "rooflines": [{"frac": 1.0}]}) == []
    assert b.hardware_fracs_above_one({"a": [{"roofline": {"frac": 1.2}}]}) == [("a[0].roofline.frac", 1.2)]
    import glob
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_bench_default_full_record*.json")))
    for f in newest:                                       # this round's judged copies: no fraction above 1 anywhere
        rec = json.load(open(f))
        assert b.hardware_fracs_above_one(rec) == [], f
        soma = rec["configs3_prm_soma"]["config"]
        assert soma["peaks_back_propagated"] == soma["peaks_per_tile"], soma
    line = json.loads(b.compact_line(dict(r05, accounting_errors=["%s = %.3f > 1" % x for x in bad])))
    assert line["accounting_errors"]

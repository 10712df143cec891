#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r3_run14; mkdir -p $O
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?"
tail -4 $O/pytest_gpu.txt | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/rp_gap -o gap -- python3 /root/repo/bench.py --no-cpu-baseline > $O/bench_rocprof.json 2> $O/rp.err || { tail -5 $O/rp.err; exit 1; }
python3 /root/repo/tools/step_gaps.py $(find /tmp/rp_gap -name "*_results.db" | head -1) 40 > $O/step_gaps.txt
grep -E "steps:|<<<|next step" $O/step_gaps.txt
cd /root/repo
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --interleaved > $O/bench_$i.json 2> $O/bench_$i.err; python3 - $O/bench_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms_per_step %.3f sustained %.3f interleaved %s" % (d["ms_per_step"], d["sustained"]["ms_per_step"], d.get("interleaved", {}).get("ms_per_step_runs")))
PY
done

#!/bin/bash
cd /root/repo
O=/root/repo/gpurun_out/r3_fc; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_compat.py -q -m gpu -k "linear" > $O/pytest_linear.txt 2>&1; echo "pytest linear rc=$?"
tail -3 $O/pytest_linear.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/rp_gap -o gap -- python3 /root/repo/bench.py --no-cpu-baseline > $O/bench_rocprof.json 2> $O/rp.err || { tail -5 $O/rp.err; exit 1; }
python3 /root/repo/tools/step_gaps.py $(find /tmp/rp_gap -name "*_results.db" | head -1) 40 > $O/step_gaps.txt
grep -E "steps:|fc_|<<<" $O/step_gaps.txt

#!/bin/bash
O=/root/repo/gpurun_out/r3_run13; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/rp_gap -o gap -- python3 /root/repo/bench.py --no-cpu-baseline > $O/bench.json 2> $O/rp.err || { tail -5 $O/rp.err; exit 1; }
python3 /root/repo/tools/step_gaps.py $(find /tmp/rp_gap -name "*_results.db" | head -1) 40 > $O/step_gaps.txt
cat $O/step_gaps.txt

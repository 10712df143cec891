#!/bin/bash
# final evidence of the round: profile round (bench lines, rocprof, PMC, stamps, timeline) + differential fuzz against the oracle
cd /root/repo
bash tools/profile_round.sh r03 > gpurun_out/profile_round.log 2>&1; echo "profile_round rc=$?"
timeout -k 10 500 python tools/fuzz_parity.py 25 3 > gpurun_out/prof/r03_fuzz_parity.txt 2>&1; echo "fuzz rc=$?"
tail -16 gpurun_out/prof/r03_fuzz_parity.txt

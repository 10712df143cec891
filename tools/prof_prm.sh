#!/bin/bash
# rocprofv3 kernel statistics of the two PRM tile workloads (per-kernel durations, launches per tile).  usage: bash tools/prof_prm.sh TAG
T=${1:-r05}
P=/root/repo/gpurun_out/prof; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
for w in prm prm-nuclei; do
  n=$([ $w = prm ] && echo soma || echo nuclei)
  rm -rf /tmp/rp_$n
  rocprofv3 --kernel-trace -d /tmp/rp_$n -o $n -- python3 /root/repo/bench.py --workload $w --no-cpu-baseline --steps 20 --warmup 5 > $P/${T}_bench_prm_${n}_under_rocprof.json 2>/tmp/rp_$n.err
  python3 /root/repo/tools/rocpd_stats.py $(find /tmp/rp_$n -name "*_results.db" | head -1) $P/${T}_prm_${n}_kernel_stats.csv > $P/${T}_${n}_stats.txt
done

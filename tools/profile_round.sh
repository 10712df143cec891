set -e
cd /root/repo
mkdir -p gpurun_out/prof
python bench.py > gpurun_out/prof/r02_bench_detect.json 2> gpurun_out/prof/bench_detect.err
python bench.py --workload prm --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/prof/r02_bench_prm_soma.json
python bench.py --workload prm-nuclei --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/prof/r02_bench_prm_nuclei.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/rp_det -o det -- python3 /root/repo/bench.py --no-cpu-baseline > /root/repo/gpurun_out/prof/r02_bench_detect_under_rocprof.json 2>/tmp/rp_det.err
python3 /root/repo/tools/rocpd_stats.py $(find /tmp/rp_det -name "*_results.db" | head -1) /root/repo/gpurun_out/prof/r02_bench_detect_kernel_stats.csv > /root/repo/gpurun_out/prof/det_stats.txt
rocprofv3 --kernel-trace -d /tmp/rp_soma -o soma -- python3 /root/repo/bench.py --workload prm --no-cpu-baseline > /root/repo/gpurun_out/prof/r02_bench_prm_soma_under_rocprof.json 2>/tmp/rp_soma.err
python3 /root/repo/tools/rocpd_stats.py $(find /tmp/rp_soma -name "*_results.db" | head -1) /root/repo/gpurun_out/prof/r02_prm_soma_kernel_stats.csv > /root/repo/gpurun_out/prof/soma_stats.txt
rocprofv3 --kernel-trace -d /tmp/rp_nuc -o nuc -- python3 /root/repo/bench.py --workload prm-nuclei --no-cpu-baseline > /root/repo/gpurun_out/prof/r02_bench_prm_nuclei_under_rocprof.json 2>/tmp/rp_nuc.err
python3 /root/repo/tools/rocpd_stats.py $(find /tmp/rp_nuc -name "*_results.db" | head -1) /root/repo/gpurun_out/prof/r02_prm_nuclei_kernel_stats.csv > /root/repo/gpurun_out/prof/nuc_stats.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -- python3 /root/repo/tools/pmc_probe.py > /tmp/pF.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -- python3 /root/repo/tools/pmc_probe.py > /tmp/pW.log 2>&1
cp $(find /tmp/pmcF -name "*counter_collection.csv" | head -1) /root/repo/gpurun_out/prof/fetch_size_counter_collection.csv
cp $(find /tmp/pmcW -name "*counter_collection.csv" | head -1) /root/repo/gpurun_out/prof/write_size_counter_collection.csv
cd /root/repo && python3 tools/pmc_traffic.py gpurun_out/prof/fetch_size_counter_collection.csv gpurun_out/prof/write_size_counter_collection.csv 1342177280 gpurun_out/prof/r02_pmc_traffic.json > /dev/null
ls -la gpurun_out/prof; tail -1 gpurun_out/prof/r02_bench_detect.json | cut -c1-300

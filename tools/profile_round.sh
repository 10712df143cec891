#!/bin/bash
# Round profiles: bench lines, rocprofv3 kernel statistics of the same commands, PMC traffic and MFMA-busy passes.
#   usage (on the GPU box): bash tools/profile_round.sh r03      -> gpurun_out/prof/ ; copy what should be judged into profiles/
R=${1:-r04}
PHASE=${2:-AB}          # A: bench lines + kernel traces;  B: PMC passes, layer benches, stamps (two gpurun calls fit the 20-minute limit)
cd /root/repo
P=/root/repo/gpurun_out/prof; mkdir -p $P
if [[ $PHASE == *A* ]]; then
python bench.py > $P/${R}_bench_default_line.json 2> $P/bench_detect.err          # the driver's command: headline + every sub-record
cp gpurun_out/bench_full_detect_n1.json $P/${R}_bench_default_full_record.json     # (the later --no-subrecords runs overwrite that file)
python bench.py --workload prm 2>/dev/null | tail -1 > $P/${R}_bench_prm_soma.json
python bench.py --workload prm-nuclei 2>/dev/null | tail -1 > $P/${R}_bench_prm_nuclei.json
python bench.py --workload prm-nuclei --prm-rpn-logit-scale 1.0 --no-cpu-baseline 2>/dev/null | tail -1 > $P/${R}_bench_prm_nuclei_saturated_init.json   # rounds 1-3 workload
python bench.py --workload volume 2>/dev/null | tail -1 > $P/${R}_bench_volume.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/rp_det -o det -- python3 /root/repo/bench.py --no-cpu-baseline --no-subrecords > $P/${R}_bench_detect_under_rocprof.json 2>/tmp/rp_det.err
python3 /root/repo/tools/rocpd_stats.py $(find /tmp/rp_det -name "*_results.db" | head -1) $P/${R}_bench_detect_kernel_stats.csv > $P/det_stats.txt
python3 /root/repo/tools/step_gaps.py $(find /tmp/rp_det -name "*_results.db" | head -1) 40 > $P/${R}_step_timeline.txt
rocprofv3 --kernel-trace -d /tmp/rp_soma -o soma -- python3 /root/repo/bench.py --workload prm --no-cpu-baseline > $P/${R}_bench_prm_soma_under_rocprof.json 2>/tmp/rp_soma.err
python3 /root/repo/tools/rocpd_stats.py $(find /tmp/rp_soma -name "*_results.db" | head -1) $P/${R}_prm_soma_kernel_stats.csv > $P/soma_stats.txt
rocprofv3 --kernel-trace -d /tmp/rp_nuc -o nuc -- python3 /root/repo/bench.py --workload prm-nuclei --no-cpu-baseline > $P/${R}_bench_prm_nuclei_under_rocprof.json 2>/tmp/rp_nuc.err
python3 /root/repo/tools/rocpd_stats.py $(find /tmp/rp_nuc -name "*_results.db" | head -1) $P/${R}_prm_nuclei_kernel_stats.csv > $P/nuc_stats.txt
rocprofv3 --kernel-trace -d /tmp/rp_vol -o vol -- python3 /root/repo/bench.py --workload volume --no-cpu-baseline > $P/${R}_bench_volume_under_rocprof.json 2>/tmp/rp_vol.err
python3 /root/repo/tools/volume_gaps.py $(find /tmp/rp_vol -name "*_results.db" | head -1) > $P/${R}_volume_gaps.txt
fi
if [[ $PHASE == *B* ]]; then
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF -- python3 /root/repo/tools/pmc_probe.py > /tmp/pF.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW -- python3 /root/repo/tools/pmc_probe.py > /tmp/pW.log 2>&1
mkdir -p $P/${R}_pmc
cp $(find /tmp/pmcF -name "*counter_collection.csv" | head -1) $P/${R}_pmc/fetch_size_counter_collection.csv
cp $(find /tmp/pmcW -name "*counter_collection.csv" | head -1) $P/${R}_pmc/write_size_counter_collection.csv
python3 /root/repo/tools/pmc_traffic.py $P/${R}_pmc/fetch_size_counter_collection.csv $P/${R}_pmc/write_size_counter_collection.csv 1342177280 $P/${R}_pmc_traffic.json > /dev/null
bash /root/repo/tools/pmc_mfma_busy.sh > $P/${R}_mfma_busy.txt 2>&1
cd /root/repo
BATCH=4 python tools/bench_layers.py 128 20 2>/dev/null | grep -v amdgpu.ids > $P/${R}_bench_layers_batch4.txt
if [ -f instanceseg-without-voxelwise-labeling_amd/csrc/libm3d_w2stamps.so ]; then
  export M3D_LIB_PATH=/root/repo/instanceseg-without-voxelwise-labeling_amd/csrc/libm3d_w2stamps.so
  python tools/w2_stamps.py conv2b conv2a conv3b conv4b 2>/dev/null | grep -v amdgpu.ids > $P/${R}_w24_stamps.txt
  python tools/stem_stamps.py 2>/dev/null | grep -v amdgpu.ids > $P/${R}_stem_stamps.txt
  unset M3D_LIB_PATH
fi
python bench.py --interleaved --pipelined --no-cpu-baseline --no-subrecords 2>/dev/null | tail -1 > $P/${R}_bench_detect_interleaved_and_pipelined.json
fi
ls -la $P | tail -40

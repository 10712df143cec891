#!/bin/bash
# PMC passes over the f16x2 conv kernel (tools/zw_probe.py): MFMA-pipe busy share + clock, shares of wave time, LDS bank conflicts.
# usage (GPU box): bash tools/zw_pmc.sh > gpurun_out/zw_pmc.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/zwpmc_$i -- python3 /root/repo/tools/zw_probe.py > /tmp/zwpmc_$i.log 2>&1 || { echo "counter set refused: $set"; tail -2 /tmp/zwpmc_$i.log; continue; }
  python3 - "$(find /tmp/zwpmc_$i -name '*counter_collection.csv' | head -1)" "$(find /tmp/zwpmc_$i -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "conv3d_zw_kernel" not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
    cnt[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    if "conv3d_zw_kernel" in r["Kernel_Name"]:
        dur[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
names = sorted({c for k in cnt for c in cnt[k]})
print("%-56s " % "kernel, grid" + " ".join("%24s" % n for n in names))
for k in cnt:
    print("%-56s " % (k[0] + " " + k[1]) + " ".join("%24.5g" % (sum(cnt[k][n]) / max(1, len(cnt[k][n]))) for n in names))
print({k: round(sum(v) / len(v) / 1e3, 1) for k, v in dur.items()}, "us (all launches of the template)")
PY
done

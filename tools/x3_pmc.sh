#!/bin/bash
# LDS bank conflicts / MFMA busy of conv3d_x3_kernel.  usage (GPU box): bash tools/x3_pmc.sh
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"; do
  d=/tmp/x3pmc_$(echo $set | tr ' ' '_')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python3 /root/repo/tools/x3_probe.py > /tmp/x3pmc.log 2>&1 || { echo "FAILED: $set"; tail -3 /tmp/x3pmc.log; continue; }
  python3 - "$(find $d -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "conv3d_x3_kernel" in r["Kernel_Name"]:
        cnt[(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in cnt.items():
    print("grid", k, {n: sum(v) / len(v) for n, v in c.items()})
PY
done

#!/bin/bash
# MFMA-pipe occupancy and the clock the chip holds, per kernel of the detection step (tools/pmc_probe.py):
#   busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs);  clock = GRBM_GUI_ACTIVE / 8 / duration
# usage (on the GPU box): bash tools/pmc_mfma_busy.sh > gpurun_out/mfma_busy.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcmb -- python3 /root/repo/tools/pmc_probe.py > /tmp/pmcmb.log 2>&1 || { tail -5 /tmp/pmcmb.log; exit 1; }
python3 - "$(find /tmp/pmcmb -name '*counter_collection.csv' | head -1)" "$(find /tmp/pmcmb -name '*kernel_trace.csv' | head -1)" <<'EOF'
import csv, sys, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("%-70s %6s %9s %8s %8s" % ("kernel", "calls", "avg us", "MFMA busy", "GHz"))
rows = []
for k, c in cnt.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c or not dur.get(k):
        continue
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]) / 8.0
    d = sum(dur[k]) / len(dur[k])
    if busy <= 0 or d < 20000:
        continue
    rows.append((d, k, len(dur[k]), busy / (1024.0 * gui), gui / d))
for d, k, n, b, g in sorted(rows, reverse=True)[:14]:
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    print("%-70s %6d %9.1f %8.2f %8.2f" % (k[:70], n, d / 1e3, b, g))
EOF

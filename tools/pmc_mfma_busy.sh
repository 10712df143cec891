#!/bin/bash
# MFMA-pipe occupancy and the clock the chip holds, per kernel of the detection step (tools/pmc_probe.py):
#   busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs);  clock = GRBM_GUI_ACTIVE / 8 / duration
# second pass: LDS bank-conflict share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE... as available) and VALU / LDS / wait shares of wave time
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

echo
echo "# second pass: shares of wave time (quad-cycle counters / SQ_WAVE_CYCLES) and LDS bank conflicts"
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-48)
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmcmb_$n -- python3 /root/repo/tools/pmc_probe.py > /tmp/pmcmb_$n.log 2>&1 || { echo "counter set refused: $set"; tail -2 /tmp/pmcmb_$n.log; continue; }
  python3 - "$(find /tmp/pmcmb_$n -name '*counter_collection.csv' | head -1)" <<'EOF'
import csv, sys, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in cnt for c in cnt[k]})
keep = [k for k in cnt if any(t in k for t in ("wino2e", "wino24", "conv3d_zw", "stem_wino", "fc_x3", "roi_align3d_fwd_v3"))]
print("%-60s " % "kernel" + " ".join("%22s" % n for n in names))
for k in sorted(keep, key=lambda k: -sum(cnt[k].get("SQ_WAVE_CYCLES", [0]))):
    short = k.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print("%-60s " % short + " ".join("%22.4g" % (sum(cnt[k][n]) / max(1, len(cnt[k][n]))) if n in cnt[k] else "%22s" % "-" for n in names))
EOF
done

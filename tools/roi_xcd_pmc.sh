#!/bin/bash
# RoIAlign3D, the two workgroup maps (tools/roi_xcd_ab.py): time, then FETCH_SIZE / WRITE_SIZE / TCC hit-miss per launch (separate passes).
#   usage (GPU box): bash tools/roi_xcd_pmc.sh > gpurun_out/roi_xcd_ab.txt
cd /root/repo
python3 tools/roi_xcd_ab.py 2>/dev/null | grep tune_roi_xcd
python3 tools/roi_xcd_ab.py --stress 2>/dev/null | grep tune_roi_xcd
cd /tmp && export TMPDIR=/tmp && export ROI_XCD_PMC=1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum"; do
  D=/tmp/roi_pmc_$(echo $C | tr ' ' '_')
  rm -rf $D
  rocprofv3 --pmc $C --output-format csv -d $D -- python3 /root/repo/tools/roi_xcd_ab.py > /tmp/roi_pmc.log 2>&1
  F=$(find $D -name "*counter_collection.csv" | head -1)
  echo "== --pmc $C (launches in order: tune_roi_xcd = 0, 1, 0, 1; FETCH_SIZE / WRITE_SIZE in KiB)"
  python3 - "$F" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "roi_align3d_fwd_v3" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
by = {}
for r in rows:
    by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), int(r["Grid_Size"]), float(r["Counter_Value"])))
for name, v in by.items():
    print("  %-16s" % name, "  ".join("grid %d: %.4g" % (g, c) for _, g, c in v[-4:]))
PY
done

# PMC pass over the PRM bench: LDS bank conflicts and MFMA-pipe occupancy per kernel (usage on the GPU box: bash tools/pmc_prm.sh > gpurun_out/pmc_prm.txt)
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pp_$n -- python3 /root/repo/bench.py --workload prm-nuclei --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pp_$n.log 2>&1 || { tail -3 /tmp/pp_$n.log; continue; }
  python3 - "$(find /tmp/pp_$n -name '*counter_collection.csv' | head -1)" "$(find /tmp/pp_$n -name '*kernel_trace.csv' | head -1)" <<'EOF'
import csv, sys, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
names = sorted({c for k in cnt for c in cnt[k]})
tot = {k: sum(v) for k, v in dur.items()}
print("%-64s %6s %9s " % ("kernel", "calls", "total ms") + " ".join("%22s" % n for n in names))
for k in sorted(tot, key=lambda k: -tot[k])[:14]:
    if k not in cnt: continue
    short = k.replace("(anonymous namespace)::", "").replace("void ", "")[:64]
    print("%-64s %6d %9.2f " % (short, len(dur[k]), tot[k] / 1e6) + " ".join("%22.4g" % (sum(cnt[k][n]) / max(1, len(cnt[k][n]))) if n in cnt[k] else "%22s" % "-" for n in names))
EOF
done

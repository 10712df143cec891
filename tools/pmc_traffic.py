#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes over tools/pmc_probe.py (FETCH_SIZE, WRITE_SIZE counter_collection.csv) into
per-kernel HBM traffic, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
  * both counters are reported in KiB;
  * gfx950's FETCH_SIZE under-counts coalesced reads (the guide measures exactly 1/2 for 16 B/lane); the factor for
    each access width is calibrated here from launches of known byte count in the same pass
    (min_partial_kernel: 4 B/lane dword loads, the conv kernels' width; torch copy: 16 B/lane);
  * WRITE_SIZE is taken as exact (checked against the copy's known byte count).
usage: pmc_traffic.py FETCH.csv WRITE.csv CAL_BYTES out.json"""
import csv, json, sys, collections


def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]) * 1024.0)
    return d


def main():
    fetch, write, cal_bytes, out = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
    def find(d, frag):
        ks = [k for k in d if frag in k[0]]
        return max(ks, key=lambda k: max(d[k])) if ks else None
    k_min = find(fetch, "min_partial_kernel")
    others = [k for k in fetch if "copyBuffer" in k[0] or "direct_copy" in k[0]]
    k_cpy = max(others, key=lambda k: max(fetch[k])) if others else None      # the 1.25 GiB torch copy
    # the calibration launches are the LARGEST of their kernel (the PRM engine's reduce_min runs the same kernel, same grid, on small
    # tensors later in the probe): calibrate on the maximum, not on the mean of the group
    f_dword = cal_bytes / max(fetch[k_min])
    f_vec = cal_bytes / max(fetch[k_cpy]) if k_cpy else None
    w_chk = max(write[k_cpy]) / cal_bytes if k_cpy and k_cpy in write else None
    res = {"units": "bytes per launch", "calibration": {"known_bytes": cal_bytes, "fetch_factor_dword_loads": f_dword,
           "fetch_factor_16B_loads": f_vec, "write_size_over_known": w_chk,
           "kernels": {"dword": k_min[0][:80], "vec": k_cpy[0][:80] if k_cpy else None}}, "kernels": []}
    for k in sorted(fetch, key=lambda k: -sum(fetch[k]) / len(fetch[k])):
        if not any(t in k[0] for t in ("conv3d", "wino24", "fc_gemm", "fc_x3_gemm", "fc_x3b_gemm", "absmax", "fc_reduce", "roi_align3d", "proposals_stage", "nms_mask_tiles", "norm1", "prm_", "window_sums")) or "pack" in k[0]:
            continue
        # A persistent kernel (conv3d_zw_kernel: one workgroup per CU whatever the layer) runs DIFFERENT layers under one (name, grid): its
        # launches are split into groups of similar FETCH_SIZE (within 25 %), one entry per group, the write counters matched by launch index
        idx = sorted(range(len(fetch[k])), key=lambda i: -fetch[k][i])
        groups = []
        for i in idx:
            if "conv3d_zw_kernel" in k[0] and groups and fetch[k][i] < 0.75 * fetch[k][groups[-1][0]]:
                groups.append([i])
            elif groups:
                groups[-1].append(i)
            else:
                groups.append([i])
        for g in groups:
            fr = sum(fetch[k][i] for i in g) / len(g)
            wv = [write[k][i] for i in g if k in write and i < len(write[k])]
            wr = sum(wv) / len(wv) if wv else None
            res["kernels"].append({"kernel": k[0].split("(float")[0].split("((anonymous")[0].split("(ZwArgs")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""), "grid": k[1],
                                   "launches": len(g), "fetch_raw": fr, "fetch_corrected": fr * f_dword, "write": wr,
                                   "traffic": fr * f_dword + (wr or 0)})
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — throughput of the 3D detection hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload backbone|detect]

A "step" is one pass of the hot path over one batch of synthetic 1x128x128x128 volumes already resident in
HBM.  Workloads (BASELINE.json configs):
  backbone  configs[1]: dsn_body forward (7 convs + BN + ReLU + 3 max-pools) on one 128^3 volume per rank
  detect    configs[2]-style: full detection-mode tile (backbone, RPN, on-device proposals, RoIAlign3D,
            box head, decode, NMS) on one 128^3 volume per rank
Rank 0 prints ONE JSON line: metric voxels/s (whole job), plus `roofline` for the dominant kernel (the
conv2b MFMA implicit-GEMM launch, timed live with HIP events on the launch stream) and `cpu_baseline`
(the oracle's torch-CPU restatement of the same workload on the host cores, rank 0, bounded sample).
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); volumes are independent, so ranks share
nothing on the data path ("weak" scaling); detect mode ends with the single all_gather of padded detections
(SURVEY 8e).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

VOL = 128
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz, no xf32 on gfx950


def host_cores():
    """Cores this process may really use: the cgroup CPU quota when there is one (the GPU box shows 256 logical
    CPUs but grants a 16-core share per GPU), else the affinity mask."""
    if "M3D_CPU_THREADS" in os.environ:
        return int(os.environ["M3D_CPU_THREADS"])
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def conv_flops(cin, cout, k, vox):
    return 2.0 * cin * cout * k ** 3 * vox


def backbone_flops(size):
    v = size ** 3
    L = [(1, 32, 5, v), (32, 64, 3, v // 8), (64, 64, 3, v // 8), (64, 128, 3, v // 64), (128, 128, 3, v // 64),
         (128, 256, 3, v // 512), (256, 256, 3, v // 512)]
    return sum(conv_flops(*l) for l in L)


def bench_prm(args, rank, world, dist):
    """configs[3]: PRM_ON soma tile 1x64x160x160: forward (2 convs per layer) + batched peak back-propagation."""
    import m3d
    from m3d.model import DetectorM3D
    from m3d.prm import PRMEngine
    from m3d.config import Cfg
    from m3d.synth import make_params, synth_volume
    from m3d import tiling
    nuclei = args.workload == "prm-nuclei"
    cfg = Cfg.nuclei(score_thresh=0.0) if nuclei else Cfg.soma()
    P = make_params(stride=cfg.stride, num_anchors=cfg.num_anchors, mlp_dim=cfg.mlp_dim, seed=0)
    eng = PRMEngine(DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg))
    S, H, W = cfg.in_size
    vol = torch.from_numpy(tiling.norm1(synth_volume(rank, (S, H, W)), np.float32).astype(np.float32)).reshape(1, 1, S, H, W).cuda()
    npk = []

    def step():
        out = eng.prm_tile(vol, dense=False)
        npk.append(0 if out is None else int(out["peaks"].shape[0]))
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # forward-only and backward-only split
    e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    e[0].record(); eng.forward(vol); e[1].record(); torch.cuda.synchronize()
    fwd_ms = e[0].elapsed_time(e[1])
    if rank == 0:
        print(json.dumps({"metric": "voxels/sec end-to-end infer_simple (PRM_ON soma tile)", "value": world * args.steps * S * H * W / dt,
                          "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic",
                          "config": {"workload": ("PRM tile 1x%dx%dx%d %s net: PRM forward + box head + batched peak back-propagation%s" %
                                                 (S, H, W, "nuclei (stride 8, 35 anchors)" if nuclei else "soma (stride 4, 14 anchors)",
                                                  "" if nuclei else " [configs[3]]")), "peaks_per_tile": npk[-1],
                                     "prm_forward_ms": fwd_ms}}))


WINO_WORK = {0: 1.0, 1: 2.0 / 3.0, 2: 4.0 / 9.0}   # fraction of the algorithmic multiply-adds issued as MFMA work


def pmc_traffic(symbol):
    """HBM-side bytes per launch of the dominant kernel, from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (tools/pmc_probe.py -> tools/pmc_traffic.py -> profiles/rNN_pmc_traffic.json; counters cannot be read from inside
    this process).  Returned as roofline.traffic with the algorithmic bytes beside it."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return {}
    try:
        d = json.load(open(files[-1]))
        k = [e for e in d["kernels"] if e["kernel"] == symbol]
        if not k:
            return {}
        alg = (64 * 64 ** 3 + 64 * 32 ** 3 + 64 * 64 * 27) * 4        # conv2b+pool: input + pooled output + weights, once each
        return {"traffic": k[0]["traffic"], "traffic_unit": "bytes/launch (FETCH_SIZE x%.2f gfx950 correction + WRITE_SIZE)" %
                d["calibration"]["fetch_factor_dword_loads"], "algorithmic_bytes_per_launch": alg,
                "traffic_source": os.path.relpath(files[-1], ROOT)}
    except Exception:
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="backbone", choices=["backbone", "detect", "prm", "prm-nuclei"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import m3d
    from m3d.model import DetectorM3D
    from m3d.config import Cfg
    from m3d.synth import make_params, synth_volume
    from m3d import tiling

    if args.workload in ("prm", "prm-nuclei"):
        return bench_prm(args, rank, world, dist)
    cfg = Cfg.nuclei()
    P = make_params(stride=8, num_anchors=35, mlp_dim=cfg.mlp_dim, seed=0, head=(args.workload == "detect"))
    det = DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg)
    # configs[1]: one volume per step; configs[2] (--workload detect): a batch of 4 volumes per step, one after the other
    # (the reference's tiler handles one tile at a time, core/test.py:91-145)
    nvol = 4 if args.workload == "detect" else 1
    vols = [torch.from_numpy(tiling.norm1(synth_volume(rank * nvol + v, (VOL, VOL, VOL)), np.float32).astype(np.float32))   # blob.py:179-184
            .view(1, 1, VOL, VOL, VOL).cuda() for v in range(nvol)]
    vol = vols[0]

    # dominant kernel = conv2b (64->64, 3^3, 64^3 voxels) with the fused BN+ReLU+MaxPool epilogue: 57.98 GFLOP
    # ALGORITHMIC per launch, 34 % of the backbone FLOPs and the single largest kernel; launched
    # once per step, so rocprofv3 --stats' per-symbol average is this launch.  Symbol: conv3d_wino_kernel<4,32,1,2,2,4,1,true>
    # (Winograd F(2,3) along x: executes 2/3 of the algorithmic multiply-adds on the matrix cores, so achieved/peak can
    # exceed 1) or, with M3D_WINO=0, the direct kernel conv3d_mfma_kernel<3,2,32,4,2,2,2,true,1>.
    dom_layers = (2,)
    dom_flops = conv_flops(64, 64, 3, (VOL // 2) ** 3)
    dom_ev = []

    # backbone workload: the layers before and after the dominant one are replayed as two HIP graphs (fewer launch gaps);
    # the dominant layer stays an eager launch between two events so its duration is measured live in the timed region.
    # Opt-in (M3D_GRAPH=1): measured 0.9247 vs 0.9251 ms per step - the eager loop is already GPU-bound (the host runs
    # ahead of nine ~100 us kernels), so the default stays the plain launches.
    graphs = None
    if args.workload == "backbone" and os.environ.get("M3D_GRAPH", "0") == "1" and det.wino_mode == 2:
        try:
            dl = dom_layers[0]
            pre, mid_in = det.capture_body(vol, 0, dl)
            mid_out = det.body_layer(dl, mid_in).clone()
            post, final = det.capture_body(mid_out, dl + 1, None)
            graphs = (pre, mid_in, mid_out, post, final, dl)
        except Exception as e:                            # noqa: BLE001
            print("bench: graph capture unavailable (%s); eager loop" % e, file=sys.stderr)
            graphs = None

    def step_graph(timed):
        pre, mid_in, mid_out, post, final, dl = graphs
        pre()
        conv, scale, shift, pool = det.body[dl]
        e0 = e1 = None
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        det.body_wino[dl].pooled(mid_in, scale=scale, shift=shift, relu=True, out=mid_out)
        if timed:
            e1.record()
            dom_ev.append((e0, e1))
        post()
        return final

    def step(timed):
        if graphs is not None:
            return step_graph(timed)
        out = None
        for v in vols:
            out = step_volume(timed, v)
        return out

    def step_volume(timed, x):
        for li in range(len(det.body)):
            if timed and li in dom_layers:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                x = det.body_layer(li, x)
                e1.record()
                dom_ev.append((e0, e1))
            else:
                x = det.body_layer(li, x)
        if args.workload == "backbone":
            return x
        prob, deltas = det.rpn(x)
        im_info = np.array([VOL, VOL, VOL, 1.0])
        rois, probs, keep_idx = det.proposals(prob, deltas, im_info)
        cls, bbox = det.box_head(x, rois)
        pred = m3d.bbox_transform3d(rois[:, 1:7].contiguous(), bbox, cfg.bbox_reg_weights, clip_to=im_info[:3])
        sc, bx, _, _ = det.box_results_with_nms_and_limit(cls, pred, keep_idx)
        out = torch.zeros((cfg.detections_per_im, 7), device="cuda")
        n = min(sc.numel(), cfg.detections_per_im)
        out[:n, :6] = bx[:n]
        out[:n, 6] = sc[:n]
        if dist is not None:   # the path's one exchange step: all_gather of padded detections
            gathered = [torch.empty_like(out) for _ in range(world)]
            dist.all_gather(gathered, out)
        return out

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        dom_ms = float(np.mean([a.elapsed_time(b) for a, b in dom_ev]))
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12
        voxels = world * args.steps * nvol * VOL ** 3
        res = {
            "metric": "voxels/sec end-to-end infer_simple (128^3 vol); 3D-conv TFLOPS vs roofline",
            "value": voxels / dt, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("dsn_body forward (7 conv3d + BN + ReLU + 3 maxpool), 1x1x128x128x128 per rank [configs[1]]"
                                    if args.workload == "backbone" else
                                    "detection-mode infer: backbone+RPN+proposals+RoIAlign3D+2mlp head+NMS, batch of 4 x (1x128^3) per rank [configs[2]]"),
                       "volumes_per_step": world * nvol, "net": "nuclei stride-8 dsn_body, 35 anchors",
                       "backbone_gflop_per_volume": backbone_flops(VOL) / 1e9,
                       "backbone_tflops_whole_step": (backbone_flops(VOL) / (dt / args.steps) / 1e12) if args.workload == "backbone" else None,
                       "launch": "2 HIP graphs + 1 eager launch per step" if graphs is not None else "eager launches"},
            "roofline": {"bound": "mfma",
                         "kernel": ("conv3d_wino2_kernel<4,32,2,2,true> (conv2b 64->64 3^3 @64^3, Winograd F(2x2,3x3) on (y,x) + fused BN/ReLU/MaxPool)"
                                    if det.wino_mode == 2 else
                                    "conv3d_wino_kernel<4,32,1,2,2,4,1,true> (conv2b 64->64 3^3 @64^3, Winograd F(2,3) along x + fused BN/ReLU/MaxPool)"
                                    if det.wino_mode == 1 else
                                    "conv3d_mfma_kernel<3,2,32,4,2,2,2,true,1> (conv2b 64->64 3^3 @64^3 + fused BN/ReLU/MaxPool)"),
                         "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                         "traffic": None, "kernel_ms": dom_ms, "algorithmic_gflop_per_launch": dom_flops / 1e9,
                         "mfma_executed_tflops": achieved * WINO_WORK[det.wino_mode],
                         "mfma_executed_frac": achieved * WINO_WORK[det.wino_mode] / FP32_MFMA_PEAK_TFLOPS,
                         "note": ("achieved counts the ALGORITHMIC 2*Cin*Cout*27 FLOP per output voxel; the Winograd kernel issues %s of "
                                  "them as MFMA work (mfma_executed_*), which is why frac can exceed 1" % ("4/9" if det.wino_mode == 2 else "2/3"))
                                 if det.use_wino else None},
        }
        res["roofline"].update(pmc_traffic("conv3d_wino2_kernel<4, 32, 2, 2, true>" if det.wino_mode == 2 else
                                           "conv3d_wino_kernel<4, 32, 1, 2, 2, 4, 1, true>" if det.wino_mode == 1 else
                                           "conv3d_mfma_kernel<3, 2, 32, 4, 2, 2, 2, true, 1>"))
        if not args.no_cpu_baseline and world == 1:      # contract: CPU baseline on rank 0 at N = 1 only
            # CPU baseline leg: the ONLY place bench.py touches oracle/ (the checker's torch-CPU restatement)
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as O
            ncpu = host_cores()
            torch.set_num_threads(ncpu)
            Pc = P
            xv = vol.cpu()
            with torch.no_grad():
                f = O.dsn_body_forward if args.workload == "backbone" else None
                if f is not None:
                    f(Pc, xv, 8)     # warm-up
                    nrep, tcpu = 0, 0.0
                    while tcpu < 10.0 and nrep < 20:
                        c0 = time.perf_counter()
                        f(Pc, xv, 8)
                        tcpu += time.perf_counter() - c0
                        nrep += 1
                else:
                    ocfg = O.Cfg()
                    O.detect_tile(Pc, ocfg, xv)
                    nrep, tcpu = 0, 0.0
                    while tcpu < 10.0 and nrep < 10:
                        c0 = time.perf_counter()
                        O.detect_tile(Pc, ocfg, xv)
                        tcpu += time.perf_counter() - c0
                        nrep += 1
            res["cpu_baseline"] = {"value": nrep * VOL ** 3 / tcpu, "unit": "voxels/s", "cores": ncpu, "kind": "port",
                                   "sample": "%d x the same 1x128^3 workload (oracle: torch-CPU fp32 convs + oracle C ops), "
                                             "torch.set_num_threads(%d), %.1f s" % (nrep, ncpu, tcpu)}
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

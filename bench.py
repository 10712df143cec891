#!/usr/bin/env python3
"""bench.py — throughput of the 3D detection hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload detect|backbone|prm|prm-nuclei] [--stress-rois]

Default workload = BASELINE.json configs[2]: the full detection-mode pipeline of tools/infer_simple.py:249-265 /
lib/core/test.py:54-177 on a batch of 4 synthetic 1x128x128x128 volumes per rank - at every N, so that the driver's 1 -> 8
scaling ratio compares equal per-GPU work (weak scaling; `--vols-per-rank 8` at N = 8 is configs[4]'s 64 volumes, which the N > 1
line also carries as `configs4_shape`).  One "step" = every volume of the batch through
    raw uint16 volume -> norm1 (blob.py:179-184, on device) -> dsn_body -> RPN -> proposals (on device) -> RoIAlign3D ->
    2-MLP head -> decode/clip -> per-class NMS + cap -> cross-tile NMS (core/test.py:159)
followed by the path's ONE exchange: a single all_gather of the padded detections [vols, 301, 7] (m3d.shard; a no-op at N = 1).
`value` (voxels/s, whole job) is measured with the raw volumes resident in HBM when the timed region starts (the bench contract);
`e2e_host_to_host` on the same line is SURVEY 8d's end-to-end definition: the same steps with the raw volumes coming from pinned
host memory (H2D on a copy stream, double-buffered) and the gathered detections copied back to the host; `sustained` repeats the
resident loop for at least two seconds.
`roofline` is the metric's named quantity, the 3D-convolution family (dsn_body + RPN convs): MFMA FLOPs ISSUED by all its launches
over their summed live duration (HIP events on the launch stream inside the timed region) against the fp32 MFMA peak, with the
algorithmic (direct-convolution) TFLOP/s beside it and one entry per layer; `rooflines` also holds fc1 (bf16 matrix cores at fp32
accuracy: 6 MFMAs per product, i.e. a 2500 / 6 = 416.7 TF fp32-equivalent ceiling).
`cpu_baseline` is the oracle's restatement of the same per-volume pipeline (torch-CPU convs + oracle C ops) on a bounded sample.

N > 1: `python bench.py --gpus N` starts N fresh child processes itself (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in
their environment, created before this process touches the GPU) unless a launcher (torch.distributed.run) already did.  The N > 1
line also carries `without_exchange`, `single_gpu_same_batch` (rank 0's batch timed while the other ranks idle) and `exchange`
(the all_gather alone, microseconds, ranks, backend).
`--dry --backend gloo` runs the same launcher + exchange with a stub step on CPU (tests/test_host_logic.py).
Other workloads: backbone (configs[1]: dsn_body forward only), prm (configs[3]: soma PRM tile), prm-nuclei.
`--stress-rois`: RPN NMS threshold 1.0, so every volume hands the reference's RPN_POST_NMS_TOP_N = 1000 RoIs to the box head.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))

VOL = 128
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz, no xf32 on gfx950
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA = 16 x the fp32-input MFMA rate (nominal clock; random data holds less)
HBM_PEAK_GBS = 8000.0
WINO_WORK = {0: 1.0, 1: 2.0 / 3.0, 2: 4.0 / 9.0}   # fraction of the algorithmic multiply-adds issued as MFMA work
METRIC = "voxels/sec end-to-end infer_simple (128^3 vol); 3D-conv TFLOPS vs roofline"


# ------------------------------------------------------------------------------------------------ launcher (N > 1)
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """Start args.gpus child ranks of this script.  The parent never initialises the GPU (importing torch does not) and
    never exec()s; it waits for the children and returns the worst exit code.  Rank 0's JSON line goes to our stdout."""
    port = int(os.environ.get("MASTER_PORT", "0")) or free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    # wait for all; if one rank dies the others would sit in a collective for ever: stop them (exact PIDs) and report its code
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:
                    q.terminate()
        time.sleep(0.2)
    return rc


# ------------------------------------------------------------------------------------------------ helpers
def host_cores():
    """Cores this process may really use: the cgroup CPU quota when there is one (the GPU box shows 256 logical CPUs but
    grants a 16-core share per GPU), else the affinity mask."""
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


def pmc_traffic(symbol_prefix):
    """HBM-side bytes per launch of a kernel from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (tools/pmc_probe.py -> tools/pmc_traffic.py -> profiles/rNN_pmc_traffic.json; the counters cannot be collected from inside
    this process, so this is the committed measurement of the same command, newest round first)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            k = [e for e in d["kernels"] if e["kernel"].replace(" ", "").startswith(symbol_prefix.replace(" ", ""))]
            if k:
                return {"traffic": k[0]["traffic"], "traffic_unit": "bytes/launch (FETCH_SIZE x%.2f gfx950 correction + WRITE_SIZE)" %
                        d["calibration"]["fetch_factor_dword_loads"], "traffic_source": os.path.relpath(f, ROOT)}
        except Exception:
            continue
    return {"traffic": None}


def sync_max_time(dt, dist, device):
    import torch
    if dist is None:
        return dt
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


PROBE_STEPS = 3        # timed steps that carry per-kernel HIP-event spans (the rooflines' live durations); the rest run bare


def timed_loop(step, steps, warmup, dist, sync):
    """The contract's timing: W untimed steps, then exactly K steps between barrier + device synchronize on both sides.
    (gc.collect() + gc.disable() around the timed steps was tried against host hiccups: the 20-step region got 3 % SLOWER, 4.61 against
    4.47 ms per step in four A/B runs - the interpreter's allocator does not like the full collection in front of it - so the
    collector is left alone.)"""
    for _ in range(warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    return time.perf_counter() - t0


# ------------------------------------------------------------------------------------------------ dry run (CPU, gloo)
def run_dry(args, rank, world, dist):
    """Launcher + exchange rehearsal without a GPU: every rank fabricates its volumes' detections, the ONE all_gather
    of m3d.shard moves them, every rank checks the global list."""
    import torch
    from m3d import shard
    nvol = args.vols_per_rank or 4
    n_items = world * nvol
    cap = 300

    def fake(i):                                   # item i: (i % 7) + 1 detections whose columns all hold i
        return torch.full(((i % 7) + 1, 7), float(i))
    mine = shard.partition(n_items, rank, world)
    assert len(mine) == nvol
    if os.environ.get("M3D_BENCH_TEST_KILL_RANK") == str(rank):       # tests/test_host_logic.py: a rank that dies must not hang the rest
        os._exit(7)

    def step():
        got = shard.all_gather_detections([fake(i) for i in mine], cap, n_items, dist)
        assert len(got) == n_items
        for i, t in enumerate(got):
            assert t.shape == ((i % 7) + 1, 7) and float(t[0, 0]) == float(i), (i, t.shape)
    dt = timed_loop(step, args.steps, args.warmup, dist, lambda: None)
    dt = sync_max_time(dt, dist, "cpu")
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "voxels/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry": True,
                          "config": {"workload": "DRY RUN (no GPU): launcher + one all_gather of [%d,%d,7] per rank, stub detect step"
                                     % (nvol, cap + 1), "volumes_per_step": n_items, "backend": args.backend}}))


# ------------------------------------------------------------------------------------------------ PRM workloads
def cone_limited_gflop_per_peak(stride):
    """(algorithmic, issued) GFLOP of one peak's back-propagation when every layer only computes its receptive-field window (SURVEY
    8a-12): window side per layer from the top (3 -> 5 -> 7 at the RPN stride, x2 + border after each un-pool), 2*Cin*Cout*k^3 per
    voxel.  Issued: windows of 16 voxels and more run on the strip Winograd kernel (4/9), the 5^3 stem occupies 32 MFMA rows for its
    25 (dy, dx) taps, the small windows issue every product."""
    if stride == 8:
        L = [(256, 256, 3, 3), (256, 256, 3, 5), (256, 128, 3, 7), (128, 128, 3, 16), (128, 64, 3, 18), (64, 64, 3, 38), (64, 32, 3, 40),
             (32, 1, 5, 84)]
    else:
        L = [(128, 128, 3, 3), (128, 128, 3, 5), (128, 64, 3, 7), (64, 64, 3, 16), (64, 32, 3, 18), (32, 1, 5, 40)]
    alg = sum(2.0 * a * b * k ** 3 * n ** 3 for a, b, k, n in L) / 1e9
    issued = sum(2.0 * a * b * k ** 3 * n ** 3 * (32.0 / 25.0 if k == 5 else (4.0 / 9.0 if n >= 16 else 1.0)) for a, b, k, n in L) / 1e9
    return alg, issued


def bench_prm(args, rank, world, dist):
    """configs[3]: PRM_ON soma tile 1x64x160x160: forward (2 convs per layer) + batched peak back-propagation + per-detection
    Otsu binarisation down to instance labels."""
    import numpy as np
    import torch
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
    from m3d import binarize
    raw = torch.from_numpy(synth_volume(rank, (S, H, W)).astype(np.uint16)).cuda()
    mode = "nuclei" if nuclei else "soma"
    npk, nlab = [], []

    stamps = []

    def step():
        """one tile: PRM forward + box head + peak back-propagation -> uint8 quantisation (from the windows; no dense float maps) -> per-detection crop +
        normalisation -> 2D-Otsu -> largest component (+ hole fill / closing) -> instance labels (binarization_*.py loop body)"""
        stamps.append(time.perf_counter())
        out = eng.prm_tile(vol, dense=False)
        npk.append(0 if out is None else int(out["peaks"].shape[0]))
        if out is not None:
            labels, painted = binarize.segment_tile(raw, (out["windows"], out["sums"], out["origins"]), out["dets"], mode=mode)
            nlab.append(painted)
    dt = timed_loop(step, args.steps, args.warmup, dist, torch.cuda.synchronize)
    dt = sync_max_time(dt, dist, "cuda")
    ev = lambda: torch.cuda.Event(enable_timing=True)     # noqa: E731
    e = [ev() for _ in range(6)]
    e[0].record(); eng.forward(vol); e[1].record()
    out = eng.prm_tile(vol, dense=False); e[2].record()
    torch.cuda.synchronize()
    fwd_ms, prm_ms = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])
    otsu_ms, nroi, npeaks = None, 0, 0
    if out is not None:                                   # a tile without a peak above the threshold: nothing to break down
        npeaks = int(out["peaks"].shape[0])
        q = m3d.prm_quantize_windows_u8(out["windows"], out["sums"], out["origins"], (S, H, W))
        boxes = binarize.det_boxes_int(out["dets"].cpu().numpy(), (S, H, W), mode)
        torch.cuda.synchronize()
        e[4].record(); r = binarize._tile_instance_masks(raw, q, boxes, mode, 8192); e[5].record(); torch.cuda.synchronize()
        otsu_ms = e[4].elapsed_time(e[5])
        nroi = 0 if r is None else int(r[3].numel())
    # cone-limited work of the back-propagation: receptive-field windows per layer, dgrad with relu(W); algorithmic =
    # 2*Cin*Cout*k^3 per window voxel, issued = what the kernels put on the matrix cores (strip Winograd 4/9 for windows >= 16 voxels,
    # every product for the small-window GEMMs, 32/25 for the stem whose 25 (dy, dx) taps occupy 32 MFMA rows)
    cone, cone_issued = cone_limited_gflop_per_peak(cfg.stride)
    back_ms = max(prm_ms - fwd_ms, 1e-6)
    res = {"metric": "voxels/sec end-to-end infer_simple (PRM_ON tile)", "value": world * args.steps * S * H * W / dt,
           "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": ("PRM tile 1x%dx%dx%d %s net: PRM forward + box head + batched peak back-propagation + per-detection 2D-Otsu -> instance labels%s" %
                                  (S, H, W, "nuclei (stride 8, 35 anchors)" if nuclei else "soma (stride 4, 14 anchors)",
                                   "" if nuclei else " [configs[3]]")), "peaks_per_tile": npeaks,
                      "prm_forward_ms": fwd_ms, "prm_tile_ms": prm_ms, "instances_painted": int(nlab[-1].sum()) if nlab else 0,
                      "step_starts_ms_host": [round((b - a) * 1e3, 2) for a, b in zip(stamps[args.warmup:-1], stamps[args.warmup + 1:])][:args.steps]},
           "roofline": {"bound": "mfma", "kernel": "peak back-propagation of the tile's %d peaks (strip-Winograd / small-window / stem dgrad kernels)" % npeaks,
                        "achieved": npeaks * cone_issued / back_ms, "peak": 157.3, "unit": "TFLOP/s",
                        "frac": npeaks * cone_issued / back_ms / 157.3,
                        "algorithmic_tflops": npeaks * cone / back_ms,
                        "cone_limited_gflop_per_peak": cone, "issued_gflop_per_peak": cone_issued, "backward_ms": back_ms,
                        "what": "achieved / frac = fp32 MFMA FLOPs issued by the window convolutions of all peaks over backward_ms, which also holds the "
                                "tile's proposals, RoIAlign, box head and the element-wise prepare kernels; algorithmic_tflops = the cone-limited "
                                "direct count over the same time", "traffic": None},
           "otsu": {"rois": nroi, "ms": otsu_ms, "rois_per_s": nroi / otsu_ms * 1e3 if otsu_ms else None,
                    "what": "crop + normalise + 2D-Otsu + largest component%s for the tile's detections" %
                            (" + hole fill + 6-closing" if nuclei else "")}}
    if rank != 0:
        return
    if not args.no_cpu_baseline and world == 1 and npeaks > 0:
        # CPU baseline leg (the only use of oracle/ here): the oracle's restatement of PeakResponseMapping_3d.forward on the SAME tile,
        # back-propagating a capped number of peaks (a peak costs seconds of dense autograd-equivalent work on the CPU), extrapolated
        # per peak to the tile's peak count; + the per-detection Otsu loop body on the same capped sample
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        ncpu = host_cores()
        torch.set_num_threads(ncpu)
        ocfg = O.Cfg(score_thresh=0.0) if nuclei else O.Cfg.soma()
        xc = vol.cpu()
        c0 = time.perf_counter()
        O.prm_tile(P, ocfg, xc, max_peaks=0)               # forward (2 convs per layer) + proposals + box head + box results
        t_fwd = time.perf_counter() - c0
        cap_peaks = min(2, npeaks)
        c0 = time.perf_counter(); got = O.prm_tile(P, ocfg, xc, max_peaks=cap_peaks); t_peaks = max(time.perf_counter() - c0 - t_fwd, 1e-9)
        if npeaks > cap_peaks and t_fwd + min(8, npeaks) * t_peaks / cap_peaks < 25.0:     # capped at 8 peaks, ~10-30 s of CPU work
            cap_peaks = min(8, npeaks)
            c0 = time.perf_counter(); got = O.prm_tile(P, ocfg, xc, max_peaks=cap_peaks); t_peaks = max(time.perf_counter() - c0 - t_fwd, 1e-9)
        per_peak = t_peaks / cap_peaks
        # Otsu stage of the capped sample (quantised maps of the back-propagated peaks)
        prms = got[2].numpy()
        qc = np.stack([O.quantize_prm_u8(m_) for m_ in prms])
        bx = binarize.det_boxes_int(np.asarray(got[3])[:cap_peaks], (S, H, W), mode)
        c0 = time.perf_counter()
        O.segment_tile(raw.cpu().numpy(), qc, bx, mode)
        per_roi = (time.perf_counter() - c0) / cap_peaks
        t_tile = t_fwd + npeaks * (per_peak + per_roi)
        res["cpu_baseline"] = {"value": S * H * W / t_tile, "unit": "voxels/s", "cores": ncpu, "kind": "port",
                               "sample": "the same tile through the oracle's restatement (torch-CPU convs with %d threads, oracle C box ops): forward + box "
                                         "head %.2f s measured; %d of the tile's %d peaks back-propagated (%.2f s per peak) and binarised (%.3f s per "
                                         "detection), extrapolated per peak to all %d" % (ncpu, t_fwd, cap_peaks, npeaks, per_peak, per_roi, npeaks),
                               "seconds_per_tile_extrapolated": t_tile}
        res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
    print(json.dumps(res))


# ------------------------------------------------------------------------------------------------ detect / backbone
def bench_detect(args, rank, world, dist):
    import numpy as np
    import torch
    import m3d
    from m3d import shard
    from m3d.model import DetectorM3D, Probe
    from m3d.config import Cfg
    from m3d.synth import make_params, synth_volume

    backbone_only = args.workload == "backbone"
    cfg = Cfg.nuclei(in_size=(VOL, VOL, VOL))
    if args.stress_rois:
        cfg.rpn_nms_thresh = 1.0                       # nothing overlaps by more than 1: all RPN_POST_NMS_TOP_N = 1000 proposals survive
    P = make_params(stride=8, num_anchors=35, mlp_dim=cfg.mlp_dim, seed=0, head=not backbone_only)
    det = DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg)
    # 4 volumes per rank at EVERY N (configs[2] at N = 1): equal per-GPU work, so value(N) / (N * value(1)) is a weak-scaling efficiency
    nvol = args.vols_per_rank or (1 if backbone_only else 4)
    n_items = world * nvol
    im_info = np.array([VOL, VOL, VOL, 1.0], np.float64)
    cap = cfg.detections_per_im
    last = {}

    def make_batch(nv):
        """Per-rank state for nv volumes per step: (batch function raw -> packed detections, raw volumes on host / device, ndarray)."""
        items = shard.partition(world * nv, rank, world)
        rnp = np.stack([synth_volume(i, (VOL, VOL, VOL)) for i in items])               # uint16 [nv,128,128,128], what io.imread gives
        rhost = torch.from_numpy(rnp).pin_memory()
        rdev = rhost.cuda()
        xb = torch.empty((nv, 1, VOL, VOL, VOL), dtype=torch.float32, device="cuda")

        def batch(raw):
            """The rank's batch of volumes -> packed detections [nv, cap+1, 7] on the device (m3d.shard block: rows = detections,
            trailer row = count).  norm1 per volume; ONE batched pass for the convolutions, RoIAlign and the box-head GEMMs; ONE
            launch each for the proposals, the per-class NMS + cap and the cross-tile NMS + packing of all volumes; one host read
            (the proposal counts that size the GEMM)."""
            if backbone_only:                                                           # configs[1]: the 3D-conv forward alone (xb was
                return det.conv_body(xb)                                                # normalised once, outside the timed steps)
            m3d.norm1_batched(raw, f32_arith=True, out=xb)                              # blob.py:179-184, per volume statistics
            r = det.detect_batch(xb, im_info, as_dicts=False)                           # core/test.py:106-114 per volume
            last["num_rois"] = r["num_rois"]
            if "cls_boxes" not in r:
                return torch.zeros((nv, cap + 1, 7), device="cuda")
            with det.span("cross_tile_nms_pack"):                                       # core/test.py:159 (one tile per volume) + pack
                return m3d.nms3d_batched(r["cls_boxes"][:, 1], r["cls_counts"][:, 1], cfg.nms, pack_cap=cap, want_keep=False)["packed"]
        return batch, rnp, rhost, rdev, xb

    batch, raw_np, raw_host, raw_dev, xbuf = make_batch(nvol)

    if backbone_only:
        m3d.norm1_batched(raw_dev, f32_arith=True, out=xbuf)

    # `--backend gloo` (rehearsal of the N > 1 code path on a one-GPU box: ranks share the card, the exchange goes through host
    # memory) moves the packed block to the CPU for the collective; nccl (RCCL over xGMI) gathers device to device.
    via_host = dist is not None and args.backend == "gloo"

    def exchange(packed, items=None):
        return shard.all_gather_packed(packed.cpu() if via_host else packed, items or n_items, dist)   # THE exchange

    probed = {"probe": None, "left": 0}

    def step_resident():
        # HIP-event spans (Probe) on the first PROBE_STEPS steps of the timed region only: a probed step records ~40 timing events between
        # its kernels (+0.2 ms, 4 %), and past a few hundred live events the cost grows (40 probed steps: 5.9 ms per step against 4.8)
        if probed["probe"] is not None:
            det.probe = probed["probe"] if probed["left"] > 0 else None
            probed["left"] -= 1
        packed = batch(raw_dev)
        if backbone_only:
            return packed
        last["packed"] = exchange(packed)
        return last["packed"]

    # ---- (1) value: raw volumes resident in HBM
    # initialisation, before the W warm-up steps the caller asks for: every kernel's first launch (code-object load), the caching
    # allocator's and the pinned pools' growth, the HIP-event pool of the probe - none of it is the hot path
    INIT_STEPS = int(os.environ.get("M3D_BENCH_INIT_STEPS", "8"))
    det.probe = Probe() if os.environ.get("M3D_BENCH_INIT_PROBE", "1") == "1" else None
    for _ in range(INIT_STEPS):
        step_resident()
    det.probe = None
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step_resident()
    probed["probe"], probed["left"] = Probe(), PROBE_STEPS
    dt = timed_loop(step_resident, args.steps, 0, dist, torch.cuda.synchronize)
    dt = sync_max_time(dt, dist, "cpu" if via_host else "cuda")
    torch.cuda.synchronize()
    kern_ms = probed["probe"].mean_ms()
    kern_med = probed["probe"].median_ms()
    det.probe = probed["probe"] = None

    # ---- (1b) the same K steps software-pipelined over two streams: begin(k+1) = norm1 + backbone + RPN + proposals is launched
    # before finish(k) = RoIAlign + box head + box results + cross-tile NMS + exchange, so the latency-bound box kernels of one
    # batch run beside the MFMA kernels of the next (DetectorM3D.detect_batch_begin / _finish).  Reported beside `value`, which
    # stays the serial loop: there every kernel has the chip to itself and the per-kernel event timings mean what they say.
    def measure_pipelined():
        sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
        xb2 = [xbuf, torch.empty_like(xbuf)]

        def run_pipelined(n):
            prev, st = None, None
            for i in range(n + 1):
                if i < n:
                    with torch.cuda.stream(sA):
                        m3d.norm1_batched(raw_dev, f32_arith=True, out=xb2[i & 1])
                        st = det.detect_batch_begin(xb2[i & 1], im_info)
                if prev is not None:
                    with torch.cuda.stream(sB):
                        r = det.detect_batch_finish(prev, as_dicts=False)
                        packed = (m3d.nms3d_batched(r["cls_boxes"][:, 1], r["cls_counts"][:, 1], cfg.nms, pack_cap=cap, want_keep=False)["packed"]
                                  if "cls_boxes" in r else torch.zeros((nvol, cap + 1, 7), device="cuda"))
                        last["packed_pipelined"] = exchange(packed)
                prev = st if i < n else None
        torch.cuda.synchronize()
        run_pipelined(2)
        torch.cuda.synchronize()
        runs = []
        for _ in range(3):                               # three repeats: the overlap depends on how the two streams' kernels meet
            t0 = time.perf_counter()
            run_pipelined(args.steps)
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0) / args.steps * 1e3)
        ms = sorted(runs)[1]
        same = bool(torch.equal(last["packed_pipelined"].cpu(), last["packed"].cpu())) if "packed" in last else None
        return {"value": n_items * VOL ** 3 / (ms * 1e-3), "unit": "voxels/s", "ms_per_step": ms, "ms_per_step_runs": [round(r, 4) for r in runs],
                "what": "the same %d steps with begin(k+1) (norm1, backbone, RPN, proposals) launched on a second stream before "
                        "finish(k) (RoIAlign, box head, box results, cross-tile NMS, exchange); median of three repeats" % args.steps,
                "identical_to_serial": same}

    # ---- (1c) the same K steps interleaved on ONE stream: begin(k+1) is enqueued before the host waits for the proposal counts of
    # batch k, so that wait (the one host read of a step) never leaves the GPU idle; no kernel runs beside another.
    def measure_interleaved():
        xb2 = [xbuf, torch.empty_like(xbuf)]

        def run(n):
            m3d.norm1_batched(raw_dev, f32_arith=True, out=xb2[0])
            st = det.detect_batch_begin(xb2[0], im_info)
            for k in range(n):
                nxt = None
                if k + 1 < n:
                    m3d.norm1_batched(raw_dev, f32_arith=True, out=xb2[(k + 1) & 1])
                    nxt = det.detect_batch_begin(xb2[(k + 1) & 1], im_info)
                r = det.detect_batch_finish(st, as_dicts=False)
                packed = (m3d.nms3d_batched(r["cls_boxes"][:, 1], r["cls_counts"][:, 1], cfg.nms, pack_cap=cap, want_keep=False)["packed"]
                          if "cls_boxes" in r else torch.zeros((nvol, cap + 1, 7), device="cuda"))
                last["packed_interleaved"] = exchange(packed)
                st = nxt
        torch.cuda.synchronize()
        run(2)
        torch.cuda.synchronize()
        runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0) / args.steps * 1e3)
        ms = sorted(runs)[1]
        same = bool(torch.equal(last["packed_interleaved"].cpu(), last["packed"].cpu())) if "packed" in last else None
        return {"value": n_items * VOL ** 3 / (ms * 1e-3), "unit": "voxels/s", "ms_per_step": ms, "ms_per_step_runs": [round(r, 4) for r in runs],
                "what": "the same %d steps on one stream with begin(k+1) (norm1, backbone, RPN, proposals) enqueued before the host "
                        "reads the proposal counts of batch k; median of three repeats" % args.steps,
                "identical_to_serial": same}

    inter = None
    if not backbone_only and world == 1 and args.interleaved:
        try:
            inter = measure_interleaved()
        except Exception as e:
            inter = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.synchronize()

    piped = None
    if not backbone_only and world == 1 and args.pipelined:   # opt-in (N = 1 only): on some boxes the two streams overlap (0.80 x the serial
        # step), on others they do not (1.01 x) - see DESIGN.md 5; the default line carries only measurements that reproduce
        try:
            piped = measure_pipelined()
        except Exception as e:                       # ... nor lose the main line
            piped = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.synchronize()

    # ---- (2) end to end, host to host: pinned raw volumes -> H2D on a copy stream (double-buffered) -> step -> D2H
    e2e = None
    if not backbone_only:
        copy_stream = torch.cuda.Stream()
        bufs = [torch.empty_like(raw_dev) for _ in range(2)]
        ready = [torch.cuda.Event() for _ in range(2)]
        freed = [torch.cuda.Event() for _ in range(2)]
        host_out = [torch.empty((world, nvol, cap + 1, 7), dtype=torch.float32).pin_memory() for _ in range(2)]
        landed = [torch.cuda.Event() for _ in range(2)]
        state = {"i": 0}

        def upload(b):
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(freed[b])
                bufs[b].copy_(raw_host, non_blocking=True)
                ready[b].record(copy_stream)

        for b in range(2):
            freed[b].record()
        upload(0)

        def step_host():
            b = state["i"] & 1
            state["i"] += 1
            torch.cuda.current_stream().wait_event(ready[b])
            packed = batch(bufs[b])
            freed[b].record()
            # the NEXT step's volumes cross PCIe under this step's box head.  The upload is issued after this step's launches: on this
            # stack the 16.8 MB pinned hipMemcpyAsync holds the calling thread for ~0.5 ms, and issued first it kept the GPU waiting
            # for the step's kernels (host-to-host 5.0-5.2 ms against 4.7 resident; now equal)
            upload(b ^ 1)
            g = exchange(packed)
            if rank == 0:
                # outputs double-buffered like the inputs: this step's detections start their way to the host, the host waits for the
                # PREVIOUS step's (a full drain per step would leave the GPU idle while the next step is being launched: +0.25 ms);
                # the loop's closing synchronize lands the last ones inside the timed region
                host_out[b].copy_(g, non_blocking=True)
                landed[b].record()
                if state["i"] > 1:
                    landed[b ^ 1].synchronize()
            return g
        dt2 = timed_loop(step_host, args.steps, max(1, args.warmup // 2), dist, torch.cuda.synchronize)
        dt2 = sync_max_time(dt2, dist, "cpu" if via_host else "cuda")
        e2e = {"value": n_items * args.steps * VOL ** 3 / dt2, "unit": "voxels/s", "ms_per_step": dt2 / args.steps * 1e3,
               "includes": "H2D of the raw uint16 volumes (pinned, copy stream, double-buffered) + D2H of the gathered [%d,%d,7] detections "
                           "(pinned, double-buffered: the host holds step k's detections before step k+1 ends, all of them before the clock stops)"
                           % (n_items, cap + 1)}

    # ---- (3) sustained: the resident loop again for at least two seconds (the timed region above lasts ~0.1 s)
    sustained = None
    if not backbone_only:
        n_sus = max(args.steps, int(2.0 / max(dt / args.steps, 1e-6)) + 1)            # the same count on every rank (dt is the max over ranks)
        dts = timed_loop(step_resident, n_sus, 0, dist, torch.cuda.synchronize)
        dts = sync_max_time(dts, dist, "cpu" if via_host else "cuda")
        sustained = {"value": n_items * n_sus * VOL ** 3 / dts, "unit": "voxels/s", "ms_per_step": dts / n_sus * 1e3, "steps": n_sus,
                     "seconds": dts, "clock": "per-kernel clocks under this load: profiles/r03_mfma_busy.txt (GRBM_GUI_ACTIVE, separate PMC pass)"}

    # ---- (4) N > 1: what the scaling ratio is made of.  (a) the same per-rank batches WITHOUT the exchange; (b) rank 0's batch alone
    # on its GPU while the other ranks wait (a single-GPU run of the same batch inside this job); (c) the all_gather alone;
    # (d) BASELINE configs[4]'s shape (8 volumes per rank) when this run uses another batch
    no_xchg = same_batch = xchg = cfg4 = None
    if dist is not None and not backbone_only:
        dt3 = timed_loop(lambda: batch(raw_dev), args.steps, 1, dist, torch.cuda.synchronize)
        dt3 = sync_max_time(dt3, dist, "cpu" if via_host else "cuda")
        no_xchg = {"value": n_items * args.steps * VOL ** 3 / dt3, "unit": "voxels/s", "ms_per_step": dt3 / args.steps * 1e3,
                   "what": "all ranks' batches of %d volumes, no all_gather (max over ranks)" % nvol}
        dist.barrier()
        if rank == 0:
            dt4 = timed_loop(lambda: batch(raw_dev), args.steps, 1, None, torch.cuda.synchronize)
            same_batch = {"value": nvol * args.steps * VOL ** 3 / dt4, "unit": "voxels/s", "ms_per_step": dt4 / args.steps * 1e3,
                          "what": "rank 0's batch of %d volumes alone (the other ranks wait at a barrier): the single-GPU rate of the same "
                                  "per-rank work, measured inside this job" % nvol}
        dist.barrier()
        packed0 = batch(raw_dev)
        torch.cuda.synchronize()
        nx = 50
        dt5 = timed_loop(lambda: exchange(packed0), nx, 5, dist, torch.cuda.synchronize)
        dt5 = sync_max_time(dt5, dist, "cpu" if via_host else "cuda")
        xchg = {"microseconds_per_all_gather": dt5 / nx * 1e6, "ranks": dist.get_world_size(), "backend": dist.get_backend(),
                "bytes_per_rank": int(nvol * (cap + 1) * 7 * 4),
                "what": "one all_gather_into_tensor of the packed [%d,%d,7] block per rank, issued back to back" % (nvol, cap + 1)}
        if nvol != 8:
            b8, _, _, rdev8, _ = make_batch(8)
            dt6 = timed_loop(lambda: exchange(b8(rdev8), world * 8), args.steps, 2, dist, torch.cuda.synchronize)
            dt6 = sync_max_time(dt6, dist, "cpu" if via_host else "cuda")
            cfg4 = {"value": world * 8 * args.steps * VOL ** 3 / dt6, "unit": "voxels/s", "ms_per_step": dt6 / args.steps * 1e3,
                    "volumes_per_rank": 8, "volumes_per_step": world * 8,
                    "what": "BASELINE configs[4]'s partition (8 volumes per rank; 64 over 8 GPUs) with the exchange"}

    if rank != 0:
        return
    voxels = n_items * args.steps * VOL ** 3
    # ---- rooflines from the live HIP-event spans of the timed region.  `roofline` = the 3D-convolution family, the quantity the metric
    # names: MFMA FLOPs issued by all its launches over their summed duration; `rooflines` = one entry per conv layer + fc1
    wino = det.wino_mode
    kern = {k: round(v, 4) for k, v in sorted(kern_ms.items(), key=lambda kv: -kv[1])}
    roofs = {}
    work = det.conv_work(nvol, (VOL, VOL, VOL))
    fam_issued = fam_alg = fam_ms = 0.0
    for name, wk in work.items():
        if name not in kern_ms:
            continue
        ms = kern_ms[name]
        fam_issued += wk["issued_flop"]; fam_alg += wk["algorithmic_flop"]; fam_ms += ms
        roofs[name] = {"bound": "mfma", "kernel": wk["kernel"], "shape": wk["shape"], "launch": "one launch over the rank's batch of %d volumes" % nvol,
                       "achieved": wk["issued_flop"] / (ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": wk["issued_flop"] / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, "kernel_ms": ms,
                       "issued_gflop_per_launch": wk["issued_flop"] / 1e9, "algorithmic_gflop_per_launch": wk["algorithmic_flop"] / 1e9,
                       "algorithmic_tflops": wk["algorithmic_flop"] / (ms * 1e-3) / 1e12}
    if "conv2b" in roofs:
        w2 = "conv3d_wino24_kernel<4, 16, 2, 1, true" if "F(2x4" in work["conv2b"]["kernel"] else "conv3d_wino2e_kernel<4, 32, 2, 2, true>"
        roofs["conv2b"].update(pmc_traffic({2: w2, 1: "conv3d_wino_kernel<4, 32, 1, 2, 2, 4, 1, true>",
                                            0: "conv3d_mfma_kernel<3, 2, 32, 4, 2, 2, 2, true, 1>"}[wino]))
    conv_family = None
    if fam_ms > 0:
        conv_family = {"bound": "mfma",
                       "kernel": "3D-convolution family of the step: dsn_body conv1a..conv4b (+BN+ReLU+MaxPool fused) and the RPN convs, %d launches "
                                 "over the rank's batch of %d volumes" % (len([n for n in work if n in kern_ms]), nvol),
                       "achieved": fam_issued / (fam_ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": fam_issued / (fam_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                       "algorithmic_tflops": fam_alg / (fam_ms * 1e-3) / 1e12, "kernel_ms": fam_ms,
                       "issued_gflop_per_step": fam_issued / 1e9, "algorithmic_gflop_per_step": fam_alg / 1e9,
                       "what": "achieved / frac = fp32 MFMA FLOPs ISSUED (Winograd F(2x4,3x3) issues 1/3 - F(2x2,3x3) 4/9 -, the stem's F(2,5) 78/125 of the "
                               "direct convolution's multiply-adds) over the summed live duration of the launches; algorithmic_tflops = the "
                               "direct-convolution count 2*Cin*Cout*k^3 per voxel over the same time; per layer: rooflines",
                       "traffic": (roofs.get("conv2b", {}) or {}).get("traffic"),
                       "traffic_what": "HBM bytes per launch of the largest member (conv2b), PMC: " + str((roofs.get("conv2b", {}) or {}).get("traffic_source"))}
    if "fc1" in kern_ms and "num_rois" in last:
        ms = kern_ms["fc1"]
        M = int(sum(last["num_rois"]))
        Kf, Nf = 256 * 343, cfg.mlp_dim
        fl = 2.0 * M * Nf * Kf
        if "fc1" in getattr(det, "fc_split", {}):
            # bf16x3 split: six bf16 MFMAs per fp32 multiply-add (exact 3-way cut of both operands) -> priced against the bf16 peak
            r = {"bound": "mfma", "launch": "one launch over the RoIs of the rank's %d volumes (M = %d rows)" % (nvol, M),
                 "kernel": "fc_x3_gemm_kernel (Box_Head.fc1: [M,87808] x [1024,87808]^T at fp32 accuracy on v_mfma_f32_32x32x16_bf16: "
                           "exact 3-way bf16 cut of x and W, 6 products per fp32 product, split-K; + fc_reduce_kernel; <= 64 rows past a multiple of the "
                           "256-row tile go through fc_gemm_kernel's ragged-tile path in the same span)",
                 "achieved": 6.0 * fl / (ms * 1e-3) / 1e12, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": 6.0 * fl / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, "kernel_ms": ms,
                 "issued_gflop_per_launch": 6.0 * fl / 1e9, "algorithmic_gflop_per_launch": fl / 1e9,
                 "algorithmic_equivalent_tflops": fl / (ms * 1e-3) / 1e12,
                 "fp32_mfma_peak_multiple": fl / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                 "ceiling_tflops": BF16_MFMA_PEAK_TFLOPS / 6.0,
                 "frac_of_ceiling": fl / (ms * 1e-3) / 1e12 / (BF16_MFMA_PEAK_TFLOPS / 6.0),
                 "algorithmic_bytes_per_launch": M * Kf * 4.0 + Nf * Kf * 6.0 + M * Nf * 4.0,
                 "note": "achieved/frac count the bf16 MFMA FLOPs ISSUED (6 x 2MNK) against the dense bf16 peak; "
                         "algorithmic_equivalent_tflops = 2MNK / time, fp32_mfma_peak_multiple = that over the 157.3 TF fp32-input MFMA "
                         "peak the round-1 kernel was bound by.  Error vs fp64 equals the fp32 kernel's (tests/test_gpu_ops.py)"}
            r.update(pmc_traffic("fc_x3_gemm_kernel"))
        else:
            r = {"bound": "mfma", "launch": "one launch over the RoIs of the rank's %d volumes (M = %d rows)" % (nvol, M),
                 "kernel": "fc_gemm_kernel (Box_Head.fc1: [M,87808] x [1024,87808]^T, split-K fp32 MFMA GEMM; + fc_reduce_kernel)",
                 "achieved": fl / (ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": fl / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, "kernel_ms": ms, "algorithmic_gflop_per_launch": fl / 1e9,
                 "algorithmic_bytes_per_launch": (M + Nf) * Kf * 4.0 + M * Nf * 4.0,
                 "note": "every multiply-add of the GEMM is issued (no Winograd): achieved = 2*M*N*K / time of the GEMM + its split-K reduction"}
            r.update(pmc_traffic("fc_gemm_kernel"))
        roofs["fc1"] = r
    roof = conv_family
    body_ms = sum(v for k, v in kern_ms.items() if k.startswith("conv")) / nvol      # spans cover the whole batch
    res = {"metric": METRIC, "value": voxels / dt, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": ("dsn_body forward (7 conv3d + BN + ReLU + 3 maxpool; the volume normalised beforehand), 1x1x128x128x128 per rank [configs[1]]"
                                   if backbone_only else
                                   "detection-mode infer_simple: raw u16 volume -> norm1 -> dsn_body -> RPN -> proposals -> RoIAlign3D -> 2-MLP head "
                                   "-> decode -> NMS -> cross-tile NMS, batch of %d x (1x128^3) per rank, one all_gather of detections per step [%s]%s"
                                   % (nvol, "configs[2]" if (world == 1 and nvol == 4) else "%d volumes over %d GPUs%s" %
                                      (n_items, world, " = configs[4]" if n_items == 64 and world == 8 else ""),
                                      " STRESS: RPN NMS off, %d RoIs per volume (RPN_POST_NMS_TOP_N)" % cfg.post_nms_topN if args.stress_rois else "")),
                      "volumes_per_step": n_items, "volumes_per_rank": nvol, "backend": (args.backend if world > 1 else None), "net": "nuclei stride-8 dsn_body, 35 anchors, MLP 1024",
                      "inputs": "raw uint16 volumes resident in HBM at the start of the timed region",
                      "rois_per_volume": (float(np.mean(last["num_rois"])) if "num_rois" in last else None),
                      "dets_per_volume": (float(last["packed"][:, :, cap, 0].mean().item()) if "packed" in last else None),
                      "backbone_gflop_per_volume": backbone_flops(VOL) / 1e9, "backbone_ms_per_volume": body_ms,
                      "backbone_algorithmic_tflops": backbone_flops(VOL) / (body_ms * 1e-3) / 1e12 if body_ms else None,
                      "kernel_ms_per_launch": kern,
                      "init_steps_before_warmup": INIT_STEPS,
                      "kernel_ms_source": "HIP-event spans on the launch stream, first %d of the %d timed steps (a probed step carries ~40 event records, "
                                          "+0.2 ms; the other timed steps run bare)" % (min(PROBE_STEPS, args.steps), args.steps),
                      "kernel_ms_per_launch_median": {k: round(v, 4) for k, v in sorted(kern_med.items(), key=lambda kv: -kv[1])}},
           "roofline": roof, "rooflines": roofs}
    if not backbone_only and getattr(det, "fc_split", None):
        res["dtype_note"] = ("every operand, accumulator and result is fp32; fc1 / fc2 multiply on the bf16 matrix cores after an EXACT 3-way bf16 cut "
                             "of both fp32 operands (6 MFMAs per product, fp32 accumulation; error vs fp64 = the fp32-input kernel's, "
                             "tests/test_gpu_ops.py); M3D_FC_SPLIT=0 selects the fp32-input MFMA kernel")
    if piped is not None:
        res["pipelined"] = piped
    if inter is not None:
        res["interleaved"] = inter
    if e2e is not None:
        res["e2e_host_to_host"] = e2e
    if sustained is not None:
        res["sustained"] = sustained
    res["value_definition"] = ("voxels of all volumes of the step / wall time of the timed steps, raw uint16 volumes resident in HBM (bench contract); "
                               "SURVEY 8d's host-to-host definition of the same steps: e2e_host_to_host")
    for k, v in (("without_exchange", no_xchg), ("single_gpu_same_batch", same_batch), ("exchange", xchg), ("configs4_shape", cfg4)):
        if v is not None:
            res[k] = v
    if not args.no_cpu_baseline and world == 1 and not args.stress_rois:      # contract: CPU baseline on rank 0 at N = 1 only
        # CPU baseline leg: the ONLY place bench.py touches oracle/ (the checker's restatement of the same per-volume pipeline)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as O
        ncpu = host_cores()
        torch.set_num_threads(ncpu)
        ocfg = O.Cfg()

        def cpu_volume(raw):
            x = torch.from_numpy(O.norm1(raw, np.float32).astype(np.float32)).view(1, 1, VOL, VOL, VOL)
            with torch.no_grad():
                if backbone_only:
                    return O.dsn_body_forward(P, x, 8)
                r = O.detect_tile(P, ocfg, x)
            d = r["cls_boxes"][1]
            return d[O.nms_3d(np.ascontiguousarray(d, dtype=np.float32), ocfg.nms)] if len(d) else d
        cpu_volume(raw_np[0])                        # warm-up (thread pool, page-in)
        nrep, tcpu = 0, 0.0
        while tcpu < 12.0 and nrep < 8:
            c0 = time.perf_counter()
            cpu_volume(raw_np[nrep % nvol])
            tcpu += time.perf_counter() - c0
            nrep += 1
        res["cpu_baseline"] = {"value": nrep * VOL ** 3 / tcpu, "unit": "voxels/s", "cores": ncpu, "kind": "port",
                               "sample": "%d of the same 1x128^3 volumes through the oracle's restatement of the same pipeline (NumPy norm1, "
                                         "torch-CPU fp32 convs/linears with %d threads, oracle C proposals/RoIAlign/NMS on 1 thread), %.1f s"
                                         % (nrep, ncpu, tcpu)}
        res["speedup_vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 200 for --workload backbone, whose step is < 1 ms)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 5; 50 for --workload backbone)")
    ap.add_argument("--workload", default="detect", choices=["detect", "backbone", "prm", "prm-nuclei"])
    ap.add_argument("--vols-per-rank", type=int, default=0, help="volumes per rank per step (default 4 at every N; 8 = BASELINE configs[4]'s partition)")
    ap.add_argument("--interleaved", action="store_true", help="also time the one-stream begin(k+1) / finish(k) loop (N = 1)")
    ap.add_argument("--pipelined", action="store_true", help="also time the two-stream begin(k+1) / finish(k) loop (N = 1)")
    ap.add_argument("--stress-rois", action="store_true", help="RPN NMS threshold 1.0: every volume gives RPN_POST_NMS_TOP_N = 1000 RoIs to the box head")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dry", action="store_true", help="launcher + exchange rehearsal without a GPU (stub step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.workload == "backbone" else 20     # a timed region of ~0.1-0.2 s either way
    if args.warmup is None:
        args.warmup = 50 if args.workload == "backbone" else 5      # (3 left the first timed steps ~5 % slow: clocks / allocator still settling)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if not args.dry and args.backend == "nccl":
            import torch                                    # device_count() does not initialise the GPU
            have = torch.cuda.device_count()
            if have < args.gpus:
                raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s); one rank per GPU over RCCL needs %d "
                                 "(--backend gloo rehearses more ranks than GPUs)" % (args.gpus, have, args.gpus))
        sys.exit(launch_ranks(args, sys.argv[1:]))          # children are created before anything here touches a GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if args.dry:
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=rank, world_size=world)
        run_dry(args, rank, world, dist)
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback); --dry rehearses the launcher on CPU")
        torch.cuda.set_device(local_rank % torch.cuda.device_count())   # more ranks than GPUs only in the gloo rehearsal below
        if world > 1:
            import torch.distributed as dist
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
        if args.workload in ("prm", "prm-nuclei"):
            bench_prm(args, rank, world, dist)
        else:
            bench_detect(args, rank, world, dist)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

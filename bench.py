#!/usr/bin/env python3
"""bench.py — throughput of the 3D detection hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload detect|backbone|prm|prm-nuclei]

Default workload = BASELINE.json configs[2]: the full detection-mode pipeline of tools/infer_simple.py:249-265 /
lib/core/test.py:54-177 on a batch of synthetic 1x128x128x128 volumes per rank (4 at N = 1; 8 per rank at N > 1 =
configs[4]: 64 volumes over 8 GPUs).  One "step" = every volume of the batch through
    raw uint16 volume -> norm1 (blob.py:179-184, on device) -> dsn_body -> RPN -> proposals (on device) -> RoIAlign3D ->
    2-MLP head -> decode/clip -> per-class NMS + cap -> cross-tile NMS (core/test.py:159)
followed by the path's ONE exchange: a single all_gather of the padded detections [vols, 301, 7] (m3d.shard; a no-op at N = 1).
`value` (voxels/s, whole job) is measured with the raw volumes resident in HBM, as the contract asks; the same line also
carries `e2e_host_to_host`: the same steps with the raw volumes coming from pinned host memory (H2D on a copy stream,
double-buffered) and the gathered detections copied back to the host (SURVEY 8d's end-to-end definition).
`roofline` is for the dominant hand-written kernel of the step, timed live with HIP events on its launch stream;
`cpu_baseline` is the oracle's restatement of the same per-volume pipeline (torch-CPU convs + oracle C ops) on a bounded sample.

N > 1: `python bench.py --gpus N` starts N fresh child processes itself (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in
their environment, created before this process touches the GPU) unless a launcher (torch.distributed.run) already did.
`--dry --backend gloo` runs the same launcher + exchange with a stub step on CPU (tests/test_host_logic.py).
Other workloads: backbone (configs[1]: dsn_body forward only), prm (configs[3]: soma PRM tile), prm-nuclei.
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


def timed_loop(step, steps, warmup, dist, sync):
    """The contract's timing: W untimed steps, then exactly K steps between barrier + device synchronize on both sides."""
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
    nvol = args.vols_per_rank or (4 if world == 1 else 8)
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
    """Algorithmic FLOPs of one peak's back-propagation when every layer only computes its receptive-field window:
    window side per layer from the top (3 -> 5 -> 7 at the RPN stride, x2 + border after each un-pool), 2*Cin*Cout*k^3 per voxel."""
    if stride == 8:
        L = [(256, 256, 3, 3), (256, 256, 3, 5), (256, 128, 3, 7), (128, 128, 3, 16), (128, 64, 3, 18), (64, 64, 3, 38), (64, 32, 3, 40),
             (32, 1, 5, 84)]
    else:
        L = [(128, 128, 3, 3), (128, 128, 3, 5), (128, 64, 3, 7), (64, 64, 3, 16), (64, 32, 3, 18), (32, 1, 5, 40)]
    return sum(2.0 * a * b * k ** 3 * n ** 3 for a, b, k, n in L) / 1e9


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

    def step():
        """one tile: PRM forward + box head + peak back-propagation -> uint8 quantisation (from the windows; no dense float maps) -> per-detection crop +
        normalisation -> 2D-Otsu -> largest component (+ hole fill / closing) -> instance labels (binarization_*.py loop body)"""
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
    e[3].record()
    q = m3d.prm_quantize_windows_u8(out["windows"], out["sums"], out["origins"], (S, H, W))
    boxes = binarize.det_boxes_int(out["dets"].cpu().numpy(), (S, H, W), mode)
    torch.cuda.synchronize()
    e[4].record(); r = binarize._tile_instance_masks(raw, q, boxes, mode, 8192); e[5].record(); torch.cuda.synchronize()
    fwd_ms, prm_ms, otsu_ms = e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[4].elapsed_time(e[5])
    nroi = 0 if r is None else int(r[3].numel())
    # cone-limited algorithmic work of the back-propagation: receptive-field windows per layer, dgrad with relu(W)
    cone = cone_limited_gflop_per_peak(cfg.stride)
    if rank == 0:
        print(json.dumps({"metric": "voxels/sec end-to-end infer_simple (PRM_ON tile)", "value": world * args.steps * S * H * W / dt,
                          "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic",
                          "config": {"workload": ("PRM tile 1x%dx%dx%d %s net: PRM forward + box head + batched peak back-propagation + per-detection 2D-Otsu -> instance labels%s" %
                                                 (S, H, W, "nuclei (stride 8, 35 anchors)" if nuclei else "soma (stride 4, 14 anchors)",
                                                  "" if nuclei else " [configs[3]]")), "peaks_per_tile": npk[-1],
                                     "prm_forward_ms": fwd_ms, "prm_tile_ms": prm_ms, "instances_painted": int(nlab[-1].sum()) if nlab else 0},
                          "roofline": {"bound": "mfma", "kernel": "peak back-propagation (all window convs + stem), cone-limited algorithmic FLOPs",
                                       "achieved": npk[-1] * cone / max(prm_ms - fwd_ms, 1e-6), "peak": 157.3, "unit": "TFLOP/s",
                                       "frac": npk[-1] * cone / max(prm_ms - fwd_ms, 1e-6) / 157.3,
                                       "cone_limited_gflop_per_peak": cone, "backward_ms": prm_ms - fwd_ms,
                                       "note": "backward_ms also holds proposals, RoIAlign and the box head of the tile; Winograd issues 4/9 "
                                               "of the 3^3 window convs' multiplies, so this is an algorithmic-equivalent figure"},
                          "otsu": {"rois": nroi, "ms": otsu_ms, "rois_per_s": nroi / otsu_ms * 1e3 if otsu_ms > 0 else None,
                                   "what": "crop + normalise + 2D-Otsu + largest component%s for the tile's detections" %
                                           (" + hole fill + 6-closing" if nuclei else "")}}))


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
    P = make_params(stride=8, num_anchors=35, mlp_dim=cfg.mlp_dim, seed=0, head=not backbone_only)
    det = DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg)
    nvol = args.vols_per_rank or (1 if backbone_only else (4 if world == 1 else 8))
    n_items = world * nvol
    mine = shard.partition(n_items, rank, world)
    raw_np = np.stack([synth_volume(i, (VOL, VOL, VOL)) for i in mine])              # uint16 [nvol,128,128,128], what io.imread gives
    raw_host = torch.from_numpy(raw_np).pin_memory()
    raw_dev = raw_host.cuda()
    im_info = np.array([VOL, VOL, VOL, 1.0], np.float64)
    cap = cfg.detections_per_im
    last = {}

    xbuf = torch.empty((nvol, 1, VOL, VOL, VOL), dtype=torch.float32, device="cuda")

    def batch(raw):
        """The rank's batch of volumes -> packed detections [nvol, cap+1, 7] on the device (m3d.shard block: rows = detections,
        trailer row = count).  norm1 per volume; ONE batched pass for the convolutions, RoIAlign and the box-head GEMMs; ONE
        launch each for the proposals, the per-class NMS + cap and the cross-tile NMS + packing of all volumes; one host read
        (the proposal counts that size the GEMM)."""
        if backbone_only:                                                               # configs[1]: the 3D-conv forward alone (xbuf was
            return det.conv_body(xbuf)                                                  # normalised once, outside the timed steps)
        m3d.norm1_batched(raw, f32_arith=True, out=xbuf)                                # blob.py:179-184, per volume statistics
        r = det.detect_batch(xbuf, im_info, as_dicts=False)                             # core/test.py:106-114 per volume
        last["num_rois"] = r["num_rois"]
        if "cls_boxes" not in r:
            return torch.zeros((nvol, cap + 1, 7), device="cuda")
        with det.span("cross_tile_nms_pack"):                                           # core/test.py:159 (one tile per volume) + pack
            return m3d.nms3d_batched(r["cls_boxes"][:, 1], r["cls_counts"][:, 1], cfg.nms, pack_cap=cap, want_keep=False)["packed"]

    if backbone_only:
        m3d.norm1_batched(raw_dev, f32_arith=True, out=xbuf)

    # `--backend gloo` (rehearsal of the N > 1 code path on a one-GPU box: ranks share the card, the exchange goes through host
    # memory) moves the packed block to the CPU for the collective; nccl (RCCL over xGMI) gathers device to device.
    via_host = dist is not None and args.backend == "gloo"

    def exchange(packed):
        return shard.all_gather_packed(packed.cpu() if via_host else packed, n_items, dist)            # THE exchange

    def step_resident():
        packed = batch(raw_dev)
        if backbone_only:
            return packed
        last["packed"] = exchange(packed)
        return last["packed"]

    # ---- (1) value: raw volumes resident in HBM
    for _ in range(args.warmup):
        step_resident()
    det.probe = Probe()
    dt = timed_loop(step_resident, args.steps, 0, dist, torch.cuda.synchronize)
    dt = sync_max_time(dt, dist, "cpu" if via_host else "cuda")
    torch.cuda.synchronize()
    kern_ms = det.probe.mean_ms()
    kern_med = det.probe.median_ms()
    det.probe = None

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
        t0 = time.perf_counter()
        run_pipelined(args.steps)
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t0
        same = bool(torch.equal(last["packed_pipelined"].cpu(), last["packed"].cpu())) if "packed" in last else None
        return {"value": n_items * args.steps * VOL ** 3 / dtp, "unit": "voxels/s", "ms_per_step": dtp / args.steps * 1e3,
                "what": "the same %d steps with begin(k+1) (norm1, backbone, RPN, proposals) launched on a second stream before "
                        "finish(k) (RoIAlign, box head, box results, cross-tile NMS, exchange)" % args.steps,
                "identical_to_serial": same}

    piped = None
    if not backbone_only and world == 1:             # N = 1 only: an extra measurement must not be able to stall a multi-rank run
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
        host_out = torch.empty((world, nvol, cap + 1, 7), dtype=torch.float32).pin_memory()
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
            upload(b ^ 1)                                        # next step's volumes cross PCIe while this step computes
            torch.cuda.current_stream().wait_event(ready[b])
            packed = batch(bufs[b])
            freed[b].record()
            g = exchange(packed)
            if rank == 0:
                host_out.copy_(g, non_blocking=True)
                torch.cuda.current_stream().synchronize()        # detections are on the host when the step ends
            return g
        dt2 = timed_loop(step_host, args.steps, max(1, args.warmup // 2), dist, torch.cuda.synchronize)
        dt2 = sync_max_time(dt2, dist, "cpu" if via_host else "cuda")
        e2e = {"value": n_items * args.steps * VOL ** 3 / dt2, "unit": "voxels/s", "ms_per_step": dt2 / args.steps * 1e3,
               "includes": "H2D of the raw uint16 volumes (pinned, copy stream, double-buffered) + D2H of the gathered [%d,%d,7] detections"
                           % (n_items, cap + 1)}

    # ---- (3) N > 1: the same per-rank batches WITHOUT the exchange, so that the collective's cost (and the effect of the larger
    # per-rank batch of configs[4] on the single-GPU rate) can be read off the line
    no_xchg = None
    if dist is not None and not backbone_only:
        dt3 = timed_loop(lambda: batch(raw_dev), args.steps, 1, dist, torch.cuda.synchronize)
        dt3 = sync_max_time(dt3, dist, "cpu" if via_host else "cuda")
        no_xchg = {"value": n_items * args.steps * VOL ** 3 / dt3, "unit": "voxels/s", "ms_per_step": dt3 / args.steps * 1e3,
                   "what": "all ranks' batches of %d volumes, no all_gather (max over ranks)" % nvol}

    if rank != 0:
        return
    voxels = n_items * args.steps * VOL ** 3
    # ---- rooflines of the two largest hand-written kernels, from the live HIP-event spans of the timed region; `roofline` is
    # the one with the longer launch (the dominant kernel of the step)
    wino = det.wino_mode
    conv2b_alg = nvol * conv_flops(64, 64, 3, (VOL // 2) ** 3)              # 57.98 GFLOP algorithmic per volume, batch in one launch
    kern = {k: round(v, 4) for k, v in sorted(kern_ms.items(), key=lambda kv: -kv[1])}
    roofs = {}
    if "conv2b" in kern_ms:
        ms = kern_ms["conv2b"]
        issued = conv2b_alg * WINO_WORK[wino]
        r = {"bound": "mfma",
             "launch": "one launch over the rank's batch of %d volumes" % nvol,
             "kernel": {2: "conv3d_wino2e_kernel<4,32,2,2,true> (conv2b 64->64 3^3 @64^3, Winograd F(2x2,3x3) on (y,x), eta-split 8 waves + fused BN/ReLU/MaxPool)",
                        1: "conv3d_wino_kernel<4,32,1,2,2,4,1,true> (conv2b, Winograd F(2,3) along x + fused BN/ReLU/MaxPool)",
                        0: "conv3d_mfma_kernel<3,2,32,4,2,2,2,true,1> (conv2b direct + fused BN/ReLU/MaxPool)"}[wino],
             "achieved": issued / (ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": issued / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, "kernel_ms": ms,
             "issued_gflop_per_launch": issued / 1e9, "algorithmic_gflop_per_launch": conv2b_alg / 1e9,
             "algorithmic_equivalent_tflops": conv2b_alg / (ms * 1e-3) / 1e12,
             "note": "achieved/frac count the MFMA FLOPs the kernel ISSUES (Winograd: %s of the algorithmic 2*Cin*Cout*27 per voxel); "
                     "algorithmic_equivalent_tflops is the direct-convolution FLOP count over the same time" %
                     {2: "4/9", 1: "2/3", 0: "1"}[wino]}
        r.update(pmc_traffic({2: "conv3d_wino2e_kernel<4, 32, 2, 2, true>", 1: "conv3d_wino_kernel<4, 32, 1, 2, 2, 4, 1, true>",
                              0: "conv3d_mfma_kernel<3, 2, 32, 4, 2, 2, 2, true, 1>"}[wino]))
        roofs["conv2b"] = r
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
    roof = None
    if roofs:
        dom = max(roofs, key=lambda k: roofs[k]["kernel_ms"])
        roof = roofs[dom]
    body_ms = sum(v for k, v in kern_ms.items() if k.startswith("conv")) / nvol      # spans cover the whole batch
    res = {"metric": METRIC, "value": voxels / dt, "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": ("dsn_body forward (7 conv3d + BN + ReLU + 3 maxpool; the volume normalised beforehand), 1x1x128x128x128 per rank [configs[1]]"
                                   if backbone_only else
                                   "detection-mode infer_simple: raw u16 volume -> norm1 -> dsn_body -> RPN -> proposals -> RoIAlign3D -> 2-MLP head "
                                   "-> decode -> NMS -> cross-tile NMS, batch of %d x (1x128^3) per rank, one all_gather of detections per step [%s]"
                                   % (nvol, "configs[2]" if world == 1 else "configs[4] shape: %d volumes over %d GPUs" % (n_items, world))),
                      "volumes_per_step": n_items, "volumes_per_rank": nvol, "backend": (args.backend if world > 1 else None), "net": "nuclei stride-8 dsn_body, 35 anchors, MLP 1024",
                      "inputs": "raw uint16 volumes resident in HBM at the start of the timed region",
                      "rois_per_volume": (float(np.mean(last["num_rois"])) if "num_rois" in last else None),
                      "dets_per_volume": (float(last["packed"][:, :, cap, 0].mean().item()) if "packed" in last else None),
                      "backbone_gflop_per_volume": backbone_flops(VOL) / 1e9, "backbone_ms_per_volume": body_ms,
                      "backbone_algorithmic_tflops": backbone_flops(VOL) / (body_ms * 1e-3) / 1e12 if body_ms else None,
                      "kernel_ms_per_launch": kern,
                      "kernel_ms_per_launch_median": {k: round(v, 4) for k, v in sorted(kern_med.items(), key=lambda kv: -kv[1])}},
           "roofline": roof, "rooflines": roofs}
    if not backbone_only and getattr(det, "fc_split", None):
        res["dtype_note"] = ("every operand, accumulator and result is fp32; fc1 / fc2 multiply on the bf16 matrix cores after an EXACT 3-way bf16 cut "
                             "of both fp32 operands (6 MFMAs per product, fp32 accumulation; error vs fp64 = the fp32-input kernel's, "
                             "tests/test_gpu_ops.py); M3D_FC_SPLIT=0 selects the fp32-input MFMA kernel")
    if piped is not None:
        res["pipelined"] = piped
    if e2e is not None:
        res["e2e_host_to_host"] = e2e
    if no_xchg is not None:
        res["without_exchange"] = no_xchg
    if not args.no_cpu_baseline and world == 1:      # contract: CPU baseline on rank 0 at N = 1 only
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
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 10; 200 for --workload backbone, whose step is < 1 ms)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 3; 50 for --workload backbone)")
    ap.add_argument("--workload", default="detect", choices=["detect", "backbone", "prm", "prm-nuclei"])
    ap.add_argument("--vols-per-rank", type=int, default=0, help="volumes per rank per step (default: 4 at N=1, 8 at N>1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dry", action="store_true", help="launcher + exchange rehearsal without a GPU (stub step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.workload == "backbone" else 10     # a timed region of ~0.2 s either way: the clocks have settled
    if args.warmup is None:
        args.warmup = 50 if args.workload == "backbone" else 3

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

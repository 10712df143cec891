"""Soak test of the multi-stream PRM tile (norm convs on a second stream, binarisation on a third, copy stream of the volume driver): the same
volumes through m3d.infer.infer_prm again and again, other volumes in between (so that the caching allocator hands the streams different
blocks every time), every repeat compared with the first one bit for bit - a cross-stream hazard would show as a difference sooner or later.
usage: python tools/soak_streams.py [repeats]"""
import sys; sys.path.insert(0, "/root/repo"); import __graft_entry__  # noqa
import numpy as np, torch
import bench
from m3d.model import DetectorM3D
from m3d.prm import PRMEngine
from m3d.config import Cfg
from m3d.synth import synth_volume
from m3d import infer as minfer, binarize

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8


class A:            # bench.prm_params reads the logit scale from its args
    prm_rpn_logit_scale = 0.25


bad = 0
for ds in ("nuclei", "soma"):
    cfg = Cfg.nuclei(score_thresh=0.0) if ds == "nuclei" else Cfg.soma()
    P = bench.prm_params(cfg, A)
    eng = PRMEngine(DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg))
    shape = (59, 230, 260) if ds == "nuclei" else (70, 200, 180)
    vols = [synth_volume(300 + i, shape) for i in range(3)]
    first = {}
    side = torch.cuda.Stream()
    for r in range(reps):
        for vi, im in enumerate(vols):
            res = minfer.infer_prm(eng, im, dataset=ds, out_dir=None, keep_maps=True, tile_pipeline=bool(r & 1))
            key = [(t["num"], t["dets"].tobytes(), t["peaks"].tobytes(), b"".join(m.tobytes() for m in t["prm_u8"])) for t in res]
            if vi not in first:
                first[vi] = key
                print(ds, "volume", vi, "tiles", len(res), "peaks", sum(len(t["dets"]) for t in res), flush=True)
            elif key != first[vi]:
                bad += 1
                print("MISMATCH", ds, "repeat", r, "volume", vi, flush=True)
        # one tile through the binarisation on its own stream, against the in-stream stage
        S, H, W = cfg.in_size
        raw = torch.from_numpy(synth_volume(400 + r, (S, H, W)).astype(np.uint16)).cuda()
        from m3d import tiling
        vol = torch.from_numpy(tiling.norm1(raw.cpu().numpy(), np.float32).astype(np.float32)).reshape(1, 1, S, H, W).cuda()
        out = eng.prm_tile(vol, dense=False)
        if out is not None:
            mode = "nuclei" if ds == "nuclei" else "soma"
            a = binarize.segment_tile(raw, (out["windows"], out["sums"], out["origins"]), out["dets"], mode=mode)
            l, p, done = binarize.segment_tile_on(side, raw, (out["windows"], out["sums"], out["origins"]), out["dets"], mode=mode)
            del out
            junk = torch.full((64, 84, 84, 84), float("nan"), device="cuda")
            del junk
            done.synchronize()
            if not (torch.equal(a[0], l) and torch.equal(a[1], p)):
                bad += 1
                print("MISMATCH binarise", ds, r, flush=True)
    print(ds, "repeats", reps, "done", flush=True)
print("soak:", "OK" if bad == 0 else "%d MISMATCHES" % bad)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""One case of tools/fuzz_tile.py in detail: python tools/fuzz_tile_case.py SEED CASE - where the engine's kept detections differ from the
oracle's, prints both lists' sizes, the unmatched rows and how close their scores / overlaps are to the thresholds that decided them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instanceseg-without-voxelwise-labeling_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, m3d
import oracle as O
from m3d.model import DetectorM3D
from m3d.prm import PRMEngine

seed0, ci = int(sys.argv[1]), int(sys.argv[2])
torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
rs = np.random.RandomState(seed0 * 1000 + ci)
stride = int(rs.choice([4, 8]))
A = 35 if stride == 8 else 14
shape = tuple(int(stride * rs.randint(2, 7) + rs.choice([0, 0, 1, 3])) for _ in range(3))
shape = (min(shape[0], 40), min(shape[1] + 16, 80), min(shape[2] + 16, 80))
P = O.make_params(stride=stride, num_anchors=A, mlp_dim=32, seed=int(rs.randint(1000)))
kw = dict(mlp_dim=32, score_thresh=float(rs.choice([0.0, 0.05])), pre_nms_topN=int(rs.choice([50, 300, 1000])), post_nms_topN=int(rs.choice([30, 300, 1000])))
cfg = O.Cfg(**kw) if stride == 8 else O.Cfg.soma(**kw)
if stride == 8:
    P = dict(P)
    for k in ("RPN.RPN_cls_score.weight", "RPN.RPN_cls_score.bias"):
        P[k] = P[k] * 0.25
mode = rs.randint(3)
vol = rs.rand(1, 1, *shape).astype(np.float32)
if mode == 1:
    vol *= (rs.rand(1, 1, *shape) > 0.5)
elif mode == 2:
    vol = np.round(vol * 4) / 4
vol = torch.from_numpy(vol)
thr = float(rs.choice([0.0, 0.1, 0.3]))
print("stride", stride, "tile", shape, "score_thresh", cfg.score_thresh, "nms", cfg.nms, "rpn_nms", cfg.rpn_nms_thresh, "peak thr", thr)
det = DetectorM3D({k: v.cuda() for k, v in P.items()}, cfg)
eng = PRMEngine(det)
out = eng.prm_tile(vol.cuda(), peak_threshold=thr, dense=False)
with torch.no_grad():
    ref = O.prm_tile(P, cfg, vol, peak_threshold=thr, max_peaks=0)
gd, rd = out["dets"].numpy(), np.asarray(ref[3])
print("engine dets", gd.shape, "oracle dets", rd.shape)
# the stages in between: proposals, box-head scores
r_g = det.detect_tile(vol.cuda())
r_o = O.detect_tile(P, cfg, vol)
print("rois engine / oracle:", r_g["rois"].shape[0], r_o["rois"].shape[0])
n = min(r_g["rois"].shape[0], r_o["rois"].shape[0])
print("rois equal (first %d rows): %s; max |cls diff| %.3e; max |pred box diff| %.3e" % (
    n, np.array_equal(r_g["rois"].cpu().numpy()[:n], r_o["rois"][:n]),
    np.abs(r_g["cls"].cpu().numpy()[:n] - r_o["cls"][:n]).max(), np.abs(r_g["pred_boxes"].cpu().numpy()[:n] - r_o["pred_boxes"][:n]).max()))
def key(d):
    return {tuple(np.round(x[:6], 2)) for x in d}
kg, ko = key(gd), key(rd)
for x in rd:
    if tuple(np.round(x[:6], 2)) not in kg:
        print("only in oracle:", x)
for x in gd:
    if tuple(np.round(x[:6], 2)) not in ko:
        print("only in engine:", x)
# how close is the unmatched detection's overlap with a kept, higher-scoring one to the NMS threshold?
allb = np.vstack([gd, rd])
for x in [x for x in rd if tuple(np.round(x[:6], 2)) not in kg] + [x for x in gd if tuple(np.round(x[:6], 2)) not in ko]:
    ov = O.bbox_overlaps_3d(x[None, :6].astype(np.float32), allb[:, :6].astype(np.float32))[0]
    near = np.argsort(-ov)[:4]
    print("   overlaps of", np.round(x, 4), "with the nearest kept rows:", [(float(ov[j]), float(allb[j, 6])) for j in near])
print("detect_tile keys", sorted(r_g.keys()), sorted(r_o.keys()))
for k in ("det_scores", "det_boxes", "keep_idx", "cls_keep_idx"):
    if k in r_g and k in r_o:
        a = r_g[k].cpu().numpy() if hasattr(r_g[k], "cpu") else np.asarray(r_g[k]); b = np.asarray(r_o[k])
        print(k, a.shape, b.shape)
sg = r_g["cls"].cpu().numpy()[:, 1]; so = r_o["cls"][:, 1]
bg = r_g["pred_boxes"].cpu().numpy()[:, 6:12]; bo = r_o["pred_boxes"][:, 6:12]
dg = np.hstack((bg, sg[:, None])).astype(np.float32); do = np.hstack((bo, so[:, None])).astype(np.float32)
kg_, ko_ = O.nms_3d(dg, cfg.nms), O.nms_3d(do, cfg.nms)
print("oracle NMS on the ENGINE's head outputs keeps", len(kg_), "on the oracle's", len(ko_), "; engine NMS on engine outputs", len(m3d.nms3d(torch.from_numpy(dg).cuda(), cfg.nms)))
miss = sorted(set(ko_.tolist()) - set(kg_.tolist())); extra = sorted(set(kg_.tolist()) - set(ko_.tolist()))
print("rows kept only with oracle outputs", miss, "only with engine outputs", extra)
for i in miss + extra:
    ov_g = O.bbox_overlaps_3d(dg[i:i + 1, :6], dg[:, :6])[0]; ov_o = O.bbox_overlaps_3d(do[i:i + 1, :6], do[:, :6])[0]
    j = [int(t) for t in np.argsort(-ov_o)[1:4]]
    print("  row", i, "score eng/orc %.7f %.7f" % (dg[i, 6], do[i, 6]), "neighbours", [(t, float(ov_g[t]), float(ov_o[t]), float(dg[t, 6]), float(do[t, 6])) for t in j])
print("PRM-mode forward vs detection-mode forward (engine): max |prob diff| %.3e" % float((out["crm"][0] - r_g["rpn_prob"][0]).abs().max()) if out["crm"].shape[1:] == r_g["rpn_prob"].shape[1:] or True else "")
feat, prob, deltas, saved, top = eng.forward(vol.cuda())
im_info = np.array(list(shape) + [1.0], np.float64)
rois2, probs2, kidx2 = det.proposals(prob, deltas, im_info)
print("PRM-forward proposals:", rois2.shape[0], " detection-forward proposals:", r_g["rois"].shape[0], " oracle:", r_o["rois"].shape[0])
a, b = rois2.cpu().numpy(), r_g["rois"].cpu().numpy()
if a.shape == b.shape:
    print("  max |roi diff| PRM-fwd vs det-fwd %.3e" % np.abs(a - b).max())
else:
    sa, sb = {tuple(np.round(x, 3)) for x in a}, {tuple(np.round(x, 3)) for x in b}
    print("  only in det-fwd:", [x for x in sb - sa][:3], " only in PRM-fwd:", [x for x in sa - sb][:3])
po = torch.from_numpy(r_o["rpn_prob"]) if not torch.is_tensor(r_o["rpn_prob"]) else r_o["rpn_prob"]
print("  max |prob - oracle prob|: PRM-fwd %.3e  det-fwd %.3e" % (float((prob.cpu() - po).abs().max()), float((r_g["rpn_prob"].cpu() - po).abs().max())))

#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (he13689/yolo) on CPU.

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--ref /root/reference] [--out tests/golden]

The reference is never copied: it is imported from --ref, fed seeded synthetic weights/inputs, and only
inputs + outputs (data) are written.  Shims, all recorded in the fixtures' `notes`:
  * empty stub modules for cv2 / torchvision / pycocotools (absent here; imported but unused on this path);
  * np.float = float, np.int = int (removed from numpy >= 1.24; used at utils/utils.py:117,168,186);
  * torch.Tensor.cuda = identity (utils/utils.py:170-172 and models/yolo.py:58,70 call .cuda() unconditionally);
  * torchvision.ops.nms (not in the reference tree, not installed) = oracle.postproc.torchvision_nms behind a
    recorder, so everything the reference does UP TO that call is captured exactly; the greedy core itself is
    "parity unpinned";
  * models/v4.py:252 uses an undefined self.softmax: the v4 network is run up to its three head maps and the
    (textually identical) models/v3spp.py:172-188 tail is applied to them.
"""
import argparse
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def import_reference(ref):
    import pandas  # noqa: F401  (must be imported before the numpy aliases below)
    np.float = float
    np.int = int
    for name in ["cv2", "torchvision", "pycocotools", "pycocotools.coco", "pycocotools.cocoeval"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["pycocotools.coco"].COCO = object
    sys.modules["pycocotools.cocoeval"].COCOeval = object
    sys.modules["torchvision"].ops = types.SimpleNamespace(nms=None)
    sys.path.insert(0, ref)
    import torch
    torch.Tensor.cuda = lambda self, *a, **k: self


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default="", help="write only the fixtures whose name starts with one of these comma-separated prefixes")
    args = ap.parse_args()
    args.only = tuple(args.only.split(","))
    os.makedirs(args.out, exist_ok=True)
    import_reference(args.ref)

    import torch
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from oracle import postproc as O
    from yolo_amd.synthetic import _gen, clustered_predictions, images, init_deterministic, scale_fa_heads

    def save(name, **arrs):
        if not name.startswith(args.only):
            return
        path = os.path.join(args.out, name + ".npz")
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
        print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")

    rnd = lambda key, *shape: torch.randn(*shape, generator=_gen(0, key))

    # ---- reference modules -------------------------------------------------------------------------
    import yolov5_official.model as rmodel
    import yolov5_official.modules as rmods
    import yolov5_official.utils as rutils
    import models.yolo as ryolo
    import models.common as rcommon
    import utils.utils as fa_utils
    import utils.moduleUtils as fa_mods
    import models.v3tiny, models.v2d19, models.v3spp, models.v3, models.v4
    from models import model_config as rcfg
    import data.vocapi_evaluator as rvoc

    # ================= G1: Detect decode (yolov5_official/model.py:56-81, models/yolo.py:42-71) ========
    anchors = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
    for tag, cls in (("official", rmodel.Detect), ("compat", ryolo.Detect)):
        for nc in (3, 80):
            det = cls(nc=nc, anchors=anchors, ch=(8, 8, 8)).eval()
            det.stride = torch.tensor([8., 16., 32.])
            no = nc + 5
            shapes = [(2, 3 * no, 8, 6), (2, 3 * no, 4, 3), (2, 3 * no, 2, 2)]   # non-square on purpose
            raw = [rnd(f"g1{tag}{nc}{i}", *s) * 2 for i, s in enumerate(shapes)]
            for m in det.m:  # identity head convs: feed raw maps straight into the decode
                m.weight.data.zero_(); m.bias.data.zero_()
            det.m = torch.nn.ModuleList(torch.nn.Identity() for _ in range(3))
            z, xs = det([t.clone() for t in raw])
            save(f"g1_decode_{tag}_nc{nc}", z=z.numpy(), raw0=raw[0].numpy(), raw1=raw[1].numpy(), raw2=raw[2].numpy(),
                 x0=xs[0].numpy(), stride=np.array([8., 16., 32.], np.float32),
                 anchor_grid=det.anchor_grid.view(3, 3, 2).numpy())

    # ================= family A detectors: G2 (decode), G6 (whole model), with recorded stages ==========
    def run_family_a(name, net, x, head_names, heads_override=None):
        net.eval()
        rec = {}
        mod = sys.modules[type(net).__module__]
        orig_pp = mod.postprocess

        def spy(bb, scores, conf_thresh, num_classes, nms_threshold):
            rec["pre_bboxes"], rec["pre_scores"] = bb.copy(), scores.copy()
            return orig_pp(bb, scores, conf_thresh, num_classes, nms_threshold)

        hooks = [getattr(net, h).register_forward_hook(lambda m, i, o, h=h: rec.__setitem__("head_" + h, o.detach().numpy().copy()))
                 for h in head_names]
        mod.postprocess = spy
        try:
            with torch.no_grad():
                out = net(x)
        finally:
            mod.postprocess = orig_pp
            for h in hooks:
                h.remove()
        rec["out_bboxes"], rec["out_scores"], rec["out_cls"] = out
        return rec

    def fa_model(modname, cls_name, size, cfg):
        cls = getattr(sys.modules[modname], cls_name)
        net = cls(size, 20, False, anchor_size=cfg["anchor_size_voc"])
        init_deterministic(net, seed=0)
        # residual trunks let activations grow with random BatchNorm gains: unscaled, most scores of these detectors
        # saturate to exactly 1.0 (ties whose visiting order numpy leaves open)
        scale_fa_heads(net, modname.split(".")[-1])
        return net

    def sample(a, step=53):
        return a.reshape(-1)[::step].copy()

    notes = {}
    # v3tiny: tiny input (full tensors) + the C1 configuration 416x416 (full tensors: 2 535 boxes)
    for size in (64, 416):
        net = fa_model("models.v3tiny", "YOLOv3", size, rcfg.yolov3_tiny_cfg)
        x = images(1, size, seed=size, kind="randn")
        r = run_family_a("v3tiny", net, x, ["pred_1", "pred_2"])
        save(f"g6_v3tiny_{size}", x_seed=size, **r)
    # v2d19: 416 (845 boxes) and 128
    for size in (128, 416):
        net = fa_model("models.v2d19", "YOLOv2", size, rcfg.yolov2_d19_cfg)
        x = images(1, size, seed=size + 1, kind="randn")
        r = run_family_a("v2d19", net, x, ["pred"])
        save(f"g6_v2d19_{size}", x_seed=size + 1, **r)
    # v2r50 (ResNet([2,2,2,2]) trunk, head at stride 16) and v2slim (DarkNet_Tiny trunk, stride 32)
    import models.v2r50, models.v2slim
    for tag, cfg, sizes in (("v2r50", rcfg.yolov2_r50_cfg, (128, 416)), ("v2slim", rcfg.yolov2_slim_cfg, (128, 416))):
        for size in sizes:
            net = fa_model(f"models.{tag}", "YOLOv2", size, cfg)
            x = images(1, size, seed=size + 2, kind="randn")
            r = run_family_a(tag, net, x, ["pred"])
            save(f"g6_{tag}_{size}", x_seed=size + 2, **r)
    # v3spp / v3: small full tensors at 128 (v3spp also at 608: sampled, final outputs full)
    for modname, tag in (("models.v3spp", "v3spp"), ("models.v3", "v3")):
        net = fa_model(modname, "YOLOv3", 128, rcfg.yolov3_d53_cfg)
        x = images(1, 128, seed=128, kind="randn")
        r = run_family_a(tag, net, x, ["pred_1", "pred_2", "pred_3"])
        save(f"g6_{tag}_128", x_seed=128, **r)
    net = fa_model("models.v3spp", "YOLOv3", 608, rcfg.yolov3_d53_cfg)
    x = images(1, 608, seed=608, kind="randn")
    r = run_family_a("v3spp", net, x, ["pred_1", "pred_2", "pred_3"])
    save("g6_v3spp_608", x_seed=608, step=53, **{k: (sample(v) if k.startswith(("head_", "pre_")) else v) for k, v in r.items()},
         head_sums=np.array([r[f"head_pred_{i}"].astype(np.float64).sum() for i in (1, 2, 3)]),
         head_abs_sums=np.array([np.abs(r[f"head_pred_{i}"].astype(np.float64)).sum() for i in (1, 2, 3)]))
    # v4: network up to the head maps + the v3spp tail (see module docstring)
    for size in (128,):
        net = fa_model("models.v4", "YOLOv4", size, rcfg.yolov4_cfg).eval()
        x = images(1, size, seed=size + 4, kind="randn")
        heads = {}
        hooks = [getattr(net, h).register_forward_hook(lambda m, i, o, h=h: heads.__setitem__(h, o.detach()))
                 for h in ("head_det_1", "head_det_2", "head_det_3")]
        try:
            with torch.no_grad():
                net(x)
        except AttributeError as e:
            notes["v4"] = f"reference eval branch raised: {e}"
        for h in hooks:
            h.remove()
        hs = [heads[f"head_det_{i}"].numpy() for i in (1, 2, 3)]
        # v3spp tail applied to the v4 head maps through the reference's own functions
        tot_conf, tot_cls, tot_box, hw = [], [], [], 0
        A, ncl = net.num_anchors, net.num_classes
        for p in [torch.from_numpy(h) for h in hs]:
            bs, anc, hh, ww = p.shape
            p = p.permute(0, 2, 3, 1).contiguous().view(bs, hh * ww, anc)
            tot_conf.append(p[:, :, :A].contiguous().view(bs, hh * ww * A, 1))
            tot_cls.append(p[:, :, A:(1 + ncl) * A].contiguous().view(bs, hh * ww * A, ncl))
            tot_box.append(p[:, :, (1 + ncl) * A:].contiguous().view(bs, hh * ww * A, -1))
            hw += hh * ww
        conf_pred = torch.sigmoid(torch.cat(tot_conf, 1))[0]
        xyxy = torch.cat(tot_box, 1).view(1, hw, A, 4)
        bboxes = torch.clamp((fa_utils.decode_boxes(xyxy, net.all_anchors_wh, net.grid_cell) / net.input_size)[0], 0., 1.)
        scores = torch.softmax(torch.cat(tot_cls, 1)[0], 1) * conf_pred
        ob, osc, oc = fa_utils.postprocess(bboxes.numpy(), scores.numpy(), net.conf_thresh, ncl, net.nms_thresh)
        save(f"g6_v4_{size}", x_seed=size + 4, head_head_det_1=hs[0], head_head_det_2=hs[1], head_head_det_3=hs[2],
             pre_bboxes=bboxes.numpy(), pre_scores=scores.numpy(), out_bboxes=ob, out_scores=osc, out_cls=oc)

    # ================= G3: postprocess / nms on synthetic boxes (utils/utils.py:15-42,101-133) ==========
    g = _gen(0, "g3")
    for case, (n, ncl, conf, thr) in enumerate([(600, 20, 0.001, 0.5), (2500, 20, 0.01, 0.6), (300, 3, 0.3, 0.45)]):
        centers = torch.rand(25, 2, generator=g)
        c = centers[torch.randint(0, 25, (n,), generator=g)] + torch.randn(n, 2, generator=g) * 0.02
        wh = torch.rand(n, 2, generator=g) * 0.2 + 0.02
        bb = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1)
        if case == 0:   # degenerate boxes: zero area, and pairs that only touch (1e-28 floor)
            bb[:10, 2:] = bb[:10, :2]
            bb[10:20] = bb[:10]
            bb[20:25] = torch.tensor([0.1, 0.1, 0.2, 0.2]); bb[25:30] = torch.tensor([0.2, 0.2, 0.3, 0.3])
        sc = torch.softmax(torch.randn(n, ncl, generator=g) * 2, 1) * torch.rand(n, 1, generator=g)
        ob, osc, oc = fa_utils.postprocess(bb.numpy().copy(), sc.numpy().copy(), conf, ncl, thr)
        k1 = fa_utils.nms(bb.numpy(), sc[:, 0].numpy().copy(), thr)
        save(f"g3_postprocess_{case}", bb=bb.numpy(), scores=sc.numpy(), conf=conf, thr=thr, nc=ncl,
             out_bb=ob, out_scores=osc, out_cls=oc, nms_keep_class0_scores=np.array(k1, dtype=np.int64))

    # ================= G4: non_max_suppression (yolov5_official/utils.py:113-195) ======================
    calls = []

    def recording_nms(boxes, scores, thr):
        calls.append((boxes.numpy().copy(), scores.numpy().copy(), float(thr)))
        return torch.from_numpy(O.torchvision_nms(boxes.numpy(), scores.numpy(), thr))

    rutils.torchvision.ops.nms = recording_nms
    cases = [
        ("small_ml", clustered_predictions(4, 512, 20, seed=1, img_size=320.0), dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)),
        ("small_best", clustered_predictions(4, 512, 20, seed=2, img_size=320.0), dict(conf_thres=0.25, iou_thres=0.45)),
        ("small_agnostic", clustered_predictions(3, 700, 5, seed=3, img_size=320.0), dict(conf_thres=0.1, iou_thres=0.5, agnostic=True, multi_label=True)),
        ("few", clustered_predictions(2, 300, 4, seed=4, img_size=320.0, obj_pass=0.05), dict(conf_thres=0.25, iou_thres=0.45, max_det=1000)),
        ("big_ml", clustered_predictions(2, 25200, 80, seed=5), dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)),
        ("big_best", clustered_predictions(2, 25200, 80, seed=6, obj_pass=0.02), dict(conf_thres=0.25, iou_thres=0.45)),
    ]
    for tag, pred, kw in cases:
        calls.clear()
        before = pred.clone()
        out = rutils.non_max_suppression(pred, **kw)
        assert torch.equal(before, pred), "reference mutated its input"
        arrs = {}
        if pred.numel() < 200000:
            arrs["pred"] = pred.numpy()
        else:
            arrs["pred_seed"] = np.array({"big_ml": 5, "big_best": 6}[tag])
            arrs["pred_args"] = np.array(list(pred.shape) + [int({"big_ml": 30, "big_best": 2}[tag])])
        for i, o in enumerate(out):
            arrs[f"out{i}"] = o.numpy()
        for i, (b, s, t) in enumerate(calls):
            arrs[f"nms_boxes{i}"], arrs[f"nms_scores{i}"] = b, s
        arrs["kw"] = np.array(json.dumps(kw))
        arrs["n_img"] = pred.shape[0]
        arrs["n_calls"] = len(calls)
        save(f"g4_nms_{tag}", **arrs)

    # ================= G12: the alternative NMS flavours (yolov5_official/utils.py:318-548) ============
    # torchvision.ops.boxes.{nms, batched_nms} are absent like torchvision.ops.nms above: the published greedy rule
    # (oracle.postproc.torchvision_nms; batched = the same on boxes offset by class * (max coordinate + 1)) stands in,
    # so the 'hard_nms' / 'hard_nms_batch' methods are pinned only up to that call.
    def _tv_nms(boxes, scores, thr):
        return torch.from_numpy(O.torchvision_nms(boxes.numpy(), scores.numpy(), float(thr)))

    def _tv_batched(boxes, scores, idxs, thr):
        off = idxs.to(boxes) * (boxes.max() + 1)
        return _tv_nms(boxes + off[:, None], scores, thr)

    rutils.torchvision.ops.boxes = types.SimpleNamespace(nms=_tv_nms, batched_nms=_tv_batched)
    g12 = [("a", clustered_predictions(3, 600, 6, seed=11, img_size=320.0), dict(conf_thres=0.1, nms_thres=0.5, multi_cls=True)),
           ("b", clustered_predictions(2, 900, 3, seed=12, img_size=320.0, obj_pass=0.6), dict(conf_thres=0.3, nms_thres=0.6, multi_cls=False)),
           ("c", clustered_predictions(2, 4000, 2, seed=13, clusters=6, obj_pass=0.9), dict(conf_thres=0.05, nms_thres=0.45, multi_cls=True))]
    for tag, pred, kw in g12:
        arrs = {"pred": pred.numpy(), "kw": np.array(json.dumps(kw))}
        for method in ["hard_nms_batch", "hard_nms", "hard_nms_myself", "and", "merge", "soft_nms", "diou_nms"]:
            before = pred.clone()
            out = rutils.non_max_suppression2(pred, method=method, **kw)
            assert torch.equal(before, pred), "reference mutated its input"
            for i, o in enumerate(out):
                arrs[f"{method}_{i}"] = np.zeros((0, 6), np.float32) if o is None else o.numpy()
        save(f"g12_nms2_{tag}", **arrs)
    for tag, pred, kw in [("a", clustered_predictions(3, 500, 4, seed=21, img_size=320.0), dict(conf_thres=0.1, iou_thres=0.6, max_box=256)),
                          ("b", clustered_predictions(2, 2500, 1, seed=22, clusters=12, obj_pass=0.8), dict(conf_thres=0.2, iou_thres=0.5, max_box=1500)),
                          ("c", clustered_predictions(2, 400, 3, seed=23, img_size=320.0), dict(conf_thres=0.1, iou_thres=0.45, max_box=200, agnostic=True, classes=[0, 2]))]:
        out = rutils.diou_non_max_suppression(pred.clone(), **kw)
        arrs = {"pred": pred.numpy(), "kw": np.array(json.dumps(kw))}
        for i, o in enumerate(out):
            arrs[f"out{i}"] = o.numpy()
        save(f"g12_diou_{tag}", **arrs)

    # ================= G5: blocks on tiny tensors (fp32) ==============================================
    def block_case(tag, block, x, eps=None):
        init_deterministic(block, seed=5)
        if eps is not None:
            for m in block.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.eps = eps
        block.eval()
        with torch.no_grad():
            y = block(x)
        save(tag if tag.startswith("g9_") else f"g5_{tag}", x=x.numpy(), y=y.numpy())

    x16 = rnd("g5x16", 2, 16, 12, 10)
    x32 = rnd("g5x32", 2, 32, 12, 10)
    block_case("b_conv3s2", rmods.Conv(16, 32, 3, 2), x16, eps=1e-3)
    block_case("b_bottleneck", rmods.Bottleneck(32, 32), x32, eps=1e-3)
    block_case("b_c3", rmods.C3(32, 32, n=2), x32, eps=1e-3)
    block_case("b_c3_noshortcut", rmods.C3(32, 16, n=1, shortcut=False), x32, eps=1e-3)
    block_case("b_bottleneckcsp", rmods.BottleneckCSP(32, 32, n=2), x32, eps=1e-3)
    block_case("b_spp", rmods.SPP(32, 32), rnd("g5spp", 2, 32, 20, 20), eps=1e-3)
    block_case("b_focus", rmods.Focus(3, 16, 3), rnd("g5focus", 2, 3, 16, 12), eps=1e-3)
    block_case("b_common_conv", rcommon.Conv(16, 32, 3, 1), x16)
    block_case("a_conv", fa_mods.Conv(16, 32, k=3, p=1, s=2), x16)
    block_case("a_resblock", fa_mods.resblock(32, nblocks=2), x32)
    block_case("a_cspstage", fa_mods.CSPStage(32, n=2), x32)
    block_case("a_cspbottleneck", fa_mods.CSPBottleneck(32, 32, n=2, shortcut=False), x32)
    block_case("a_bottleneck", fa_mods.Bottleneck(32, 32), x32)
    block_case("a_spp", fa_mods.SPP(), rnd("g5aspp", 2, 8, 19, 19))
    block_case("a_reorg", fa_mods.reorg_layer(2), rnd("g5reorg", 2, 8, 6, 4))
    up = torch.nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
    xu = rnd("g5up", 2, 8, 13, 13)
    save("g5_a_upsample_bilinear", x=xu.numpy(), y=up(xu).numpy())
    xp = rnd("g5pool", 2, 8, 13, 13)
    pool = torch.nn.Sequential(torch.nn.ZeroPad2d((0, 1, 0, 1)), torch.nn.MaxPool2d((2, 2), 1))
    save("g5_a_zeropad_pool", x=xp.numpy(), y=pool(xp).numpy())

    # ================= G6: YOLOv5s whole model ========================================================
    cfg_path = os.path.join(args.ref, "yolov5_official", "yamls", "yolov5s.yaml")
    for nc in (80, 20):
        m = rmodel.Model(cfg_path, nc=nc).eval()
        init_deterministic(m, seed=0, head_convs=list(m.model[-1].m))
        x = images(2, 64, seed=64)
        with torch.no_grad():
            z, xs = m(x)
        save(f"g6_v5s_nc{nc}_64", x_seed=64, z=z.numpy(), x0=xs[0].numpy(), x1=xs[1].numpy(), x2=xs[2].numpy(),
             stride=m.stride.numpy(), anchors=m.model[-1].anchors.numpy(), anchor_grid=m.model[-1].anchor_grid.numpy())
    m = rmodel.Model(cfg_path).eval()
    init_deterministic(m, seed=0, head_convs=list(m.model[-1].m))
    x = images(1, 640, seed=640)
    with torch.no_grad():
        z, xs = m(x)
    save("g6_v5s_nc80_640", x_seed=640, step=97, z_sample=sample(z.numpy(), 97), z_sum=z.double().sum().item(),
         z_abs_sum=z.double().abs().sum().item(), head_sample0=sample(xs[0].numpy(), 97), head_sample1=sample(xs[1].numpy(), 97),
         head_sample2=sample(xs[2].numpy(), 97), obj_pass_025=(z[..., 4] > 0.25).float().mean().item(),
         obj_pass_0001=(z[..., 4] > 0.001).float().mean().item())
    # compat model (models/yolo.py Detect: transposed grid) on a small non-trivial input
    mc = ryolo.Model(cfg_path, nc=20).eval()
    init_deterministic(mc, seed=0, head_convs=list(mc.model[-1].m))
    with torch.no_grad():
        zc, _ = mc(images(1, 64, seed=65))
    save("g6_v5s_compat_nc20_64", x_seed=65, z=zc.numpy())

    # ================= G11: Model.fuse() and forward(augment=True) (yolov5_official/model.py:201-265) ===
    # fuse() ends in an undefined self.info() (AttributeError) AFTER every Conv has been fused: caught here.
    m = rmodel.Model(cfg_path, nc=20).eval()
    init_deterministic(m, seed=0, head_convs=list(m.model[-1].m))
    xa = torch.rand(2, 3, 96, 128, generator=_gen(66, "g11aug"))
    with torch.no_grad():
        z_plain, _ = m(xa)
        z_aug, none = m(xa.clone(), augment=True)
        assert none is None
        try:
            m.fuse()
        except AttributeError as e:
            notes["fuse"] = f"reference fuse() raised after fusing: {e}"
        assert not any(hasattr(mod, "bn") for mod in m.modules() if type(mod) is rmods.Conv)
        z_fused, _ = m(xa)
    save("g11_v5s_fuse_augment", z_plain=z_plain.numpy(), z_aug=z_aug.numpy(), z_fused=z_fused.numpy(), x=xa.numpy())

    # ================= G7: metric helpers ==============================================================
    g = _gen(0, "g7")
    b1 = torch.rand(30, 2, generator=g) * 300
    box1 = torch.cat([b1, b1 + torch.rand(30, 2, generator=g) * 100 + 1], 1)
    b2 = torch.rand(12, 2, generator=g) * 300
    box2 = torch.cat([b2, b2 + torch.rand(12, 2, generator=g) * 100 + 1], 1)
    xywh = torch.rand(20, 4, generator=g) * 100
    coords = torch.rand(15, 4, generator=g) * 640
    sc1 = rutils.scale_coords((640, 640), coords.clone(), (480, 360))
    sc2 = rutils.scale_coords((384, 640), coords.clone(), (720, 1280), ((0.5, 0.5), (0.0, 12.0)))
    # process_batch: predictions built around labels
    labels = torch.cat([torch.randint(0, 4, (12, 1), generator=g).float(), box2], 1)
    preds = torch.cat([box2.repeat(3, 1) + torch.randn(36, 4, generator=g) * 8, torch.rand(36, 1, generator=g),
                       torch.randint(0, 4, (36, 1), generator=g).float()], 1)
    iouv = torch.linspace(0.5, 0.95, 10)
    correct = rutils.process_batch(preds, labels, iouv)
    # ap_per_class on synthetic stats
    ntp = 400
    tp = (torch.rand(ntp, 10, generator=g) < torch.linspace(0.7, 0.2, 10)).numpy()
    conf = torch.rand(ntp, generator=g).numpy()
    pcls = torch.randint(0, 6, (ntp,), generator=g).float().numpy()
    tcls = torch.randint(0, 5, (150,), generator=g).float().numpy()   # class 5 has predictions but no labels
    p, r, apm, f1, apc = rutils.ap_per_class(tp, conf, pcls, tcls)
    rec = np.sort(torch.rand(50, generator=g).numpy())
    prec = np.sort(torch.rand(50, generator=g).numpy())[::-1].copy()
    ap_one = rutils.compute_ap(rec, prec)
    save("g7_v5_metrics", box1=box1.numpy(), box2=box2.numpy(), iou=rutils.box_iou(box1, box2).numpy(),
         xywh=xywh.numpy(), xyxy=rutils.xywh2xyxy(xywh).numpy(), xywh_back=rutils.xyxy2xywh(rutils.xywh2xyxy(xywh)).numpy(),
         coords=coords.numpy(), scaled1=sc1.numpy(), scaled2=sc2.numpy(),
         preds=preds.numpy(), labels=labels.numpy(), iouv=iouv.numpy(), correct=correct.numpy(),
         tp=tp, conf=conf, pcls=pcls, tcls=tcls, p=p, r=r, ap=apm, f1=f1, ap_class=apc,
         rec=rec, prec=prec, ap_one=ap_one[0], mpre=ap_one[1], mrec=ap_one[2])

    # VOC: voc_ap + voc_eval through the reference class with a stub dataset (its evaluate() needs a
    # dataset.pull_item that does not exist, data/vocapi_evaluator.py:61)
    import tempfile
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "VOC2007", "Annotations"))
    os.makedirs(os.path.join(tmp, "VOC2007", "ImageSets", "Main"))
    np.bool = bool  # data/vocapi_evaluator.py:260 (alias removed from numpy); set only now, after all imports
    labelmap = ("cat", "dog", "bird", "fish")   # 'fish' never detected -> ap -1
    n_img = 12
    names = [f"{i:06d}" for i in range(n_img)]
    gts = {}
    with open(os.path.join(tmp, "VOC2007", "ImageSets", "Main", "test.txt"), "w") as fh:
        fh.write("\n".join(names) + "\n")
    for n in names:
        objs = []
        for _ in range(int(torch.randint(1, 5, (1,), generator=g))):
            x1, y1 = [int(v) for v in torch.randint(0, 300, (2,), generator=g)]
            w, h = [int(v) for v in torch.randint(30, 150, (2,), generator=g)]
            objs.append(dict(name=labelmap[int(torch.randint(0, 4, (1,), generator=g))], pose="Unspecified", truncated=0,
                             difficult=int(torch.rand(1, generator=g) < 0.15), bbox=[x1, y1, x1 + w, y1 + h]))
        gts[n] = objs
        xml = "<annotation>" + "".join(
            f"<object><name>{o['name']}</name><pose>{o['pose']}</pose><truncated>0</truncated><difficult>{o['difficult']}</difficult>"
            f"<bndbox><xmin>{o['bbox'][0]}</xmin><ymin>{o['bbox'][1]}</ymin><xmax>{o['bbox'][2]}</xmax><ymax>{o['bbox'][3]}</ymax></bndbox></object>"
            for o in objs) + "</annotation>"
        with open(os.path.join(tmp, "VOC2007", "Annotations", n + ".xml"), "w") as fh:
            fh.write(xml)
    ev = rvoc.VOCAPIEvaluator.__new__(rvoc.VOCAPIEvaluator)
    ev.data_root, ev.labelmap, ev.set_type, ev.year, ev.display = tmp + "/", labelmap, "test", "2007", False
    ev.devkit_path = tmp + "/VOC2007"
    ev.annopath = os.path.join(tmp, "VOC2007", "Annotations", "%s.xml")
    ev.imgsetpath = os.path.join(tmp, "VOC2007", "ImageSets", "Main", "test.txt")
    ev.output_dir = os.path.join(tmp, "out")
    ev.dataset = types.SimpleNamespace(ids=[(tmp, n) for n in names])
    all_boxes = [[np.empty((0, 5), np.float32) for _ in names] for _ in labelmap]
    for ci, cname in enumerate(labelmap[:3]):
        for ii, n in enumerate(names):
            dets = []
            for o in gts[n]:
                if o["name"] == cname and float(torch.rand(1, generator=g)) < 0.8:
                    jit = torch.randn(4, generator=g).numpy() * 12
                    dets.append(list(np.array(o["bbox"], np.float32) + jit) + [float(torch.rand(1, generator=g))])
            for _ in range(int(torch.randint(0, 3, (1,), generator=g))):
                x1, y1 = [float(v) for v in torch.rand(2, generator=g) * 300]
                dets.append([x1, y1, x1 + 60, y1 + 80, float(torch.rand(1, generator=g)) * 0.5])
            if dets:
                all_boxes[ci][ii] = np.array(dets, np.float32)
    class _Dets(np.ndarray):
        """`dets == []` (data/vocapi_evaluator.py:145) raises on numpy >= 2 for non-empty arrays; with the numpy
        of the reference's time it was simply falsy.  Restore that one comparison."""
        def __eq__(self, other):
            return False if isinstance(other, list) else np.ndarray.__eq__(self, other)
    notes["voc"] = "all_boxes entries wrapped so that `dets == []` is False (numpy>=2 raises there)"
    ev.evaluate_detections([[b.view(_Dets) for b in row] for row in all_boxes])
    aps = []
    for cname in labelmap:
        aps.append(ev.voc_eval(ev.get_voc_results_file_template(cname), cname, os.path.join(ev.devkit_path, "annotations_cache"),
                               0.5, True)[2])
    aps_area = [ev.voc_eval(ev.get_voc_results_file_template(c), c, os.path.join(ev.devkit_path, "annotations_cache"), 0.5, False)[2]
                for c in labelmap]
    flat = {f"boxes_{ci}_{ii}": all_boxes[ci][ii] for ci in range(4) for ii in range(n_img)}
    save("g7_voc", aps=np.array(aps), aps_area=np.array(aps_area), map=ev.map, labelmap=np.array(labelmap), names=np.array(names),
         gts=np.array(json.dumps(gts)), voc_ap07=ev.voc_ap(rec, prec, True), voc_ap_area=ev.voc_ap(rec, prec, False), **flat)

    # ================= G10: pre-processing (yolov5_official/utils.py:40-75, data/transformers.py:456-471) ==========
    # cv2 is not installed: the reference's letterbox / base_transform run with cv2.resize = the oracle's restatement
    # of OpenCV's 8-bit bilinear path and cv2.copyMakeBorder = a constant pad, both behind recorders.  What is pinned is
    # everything the reference itself computes: resize target, border widths, ratio / pad, colour, normalisation.
    from oracle import preproc as OP
    sys.modules["torchvision"].transforms = types.ModuleType("torchvision.transforms")   # imported, never used on this path
    import data.transformers as rtf
    cv2 = sys.modules["cv2"]
    calls = []
    def _resize(img, dsize, interpolation=None):
        calls.append(("resize", tuple(int(v) for v in dsize), interpolation))
        return OP.resize_linear_u8(img, dsize)
    def _border(img, top, bottom, left, right, kind, value=None):
        calls.append(("border", (top, bottom, left, right), tuple(value)))
        out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, 3), np.uint8)
        out[:] = np.asarray(value, np.uint8)
        out[top:top + img.shape[0], left:left + img.shape[1]] = img
        return out
    cv2.resize, cv2.copyMakeBorder, cv2.INTER_LINEAR, cv2.BORDER_CONSTANT = _resize, _border, 1, 0
    rutils.cv2 = cv2
    rtf.cv2 = cv2
    gp = np.random.default_rng(10)
    pre = {}
    lb_cases = [((120, 160), (160, 160), dict(auto=False)), ((75, 100), (160, 160), dict(auto=True)),
                ((90, 160), (96, 96), dict(auto=False)), ((192, 192), (96, 96), dict(auto=False)),
                ((100, 37), (96, 64), dict(auto=False, scaleup=False)), ((67, 100), 96, dict(auto=False, scaleFill=True)),
                ((64, 64), (64, 64), dict(auto=False)), ((37, 53), (128, 160), dict(auto=False)),
                ((125, 88), (160, 160), dict(auto=True, stride=64))]
    for i, (shape, new_shape, kw) in enumerate(lb_cases):
        img = gp.integers(0, 256, size=shape + (3,), dtype=np.uint8)
        calls.clear()
        out, ratio, pad = rutils.letterbox(img.copy(), new_shape=new_shape, **kw)
        rs = [c for c in calls if c[0] == "resize"]
        bd = [c for c in calls if c[0] == "border"][0]
        pre[f"lb{i}_img"], pre[f"lb{i}_out"] = img, out
        pre[f"lb{i}_new_shape"] = np.array(new_shape if isinstance(new_shape, tuple) else (new_shape, new_shape))
        pre[f"lb{i}_kw"] = np.array(json.dumps(kw))
        pre[f"lb{i}_resize"] = np.array(rs[0][1] if rs else (-1, -1))
        pre[f"lb{i}_border"] = np.array(bd[1])
        pre[f"lb{i}_ratio"], pre[f"lb{i}_pad"] = np.array(ratio, np.float64), np.array(pad, np.float64)
    pre["n_lb"] = len(lb_cases)
    for i, (shape, size) in enumerate([((75, 100), 96), ((120, 160), 64), ((128, 128), 64), ((20, 30), 64)]):
        img = gp.integers(0, 256, size=shape + (3,), dtype=np.uint8)
        bt = rtf.BaseTransform(size)
        x, _, _ = bt(img.copy())
        pre[f"bt{i}_img"], pre[f"bt{i}_out"], pre[f"bt{i}_size"] = img, x, size
        pre[f"bt{i}_mean"], pre[f"bt{i}_std"] = bt.mean, bt.std
    pre["n_bt"] = 4
    save("g10_preproc", **pre)

    # ================= G8: structure (state_dict keys + shapes) =======================================
    struct = {}
    def keys_of(model):
        return [[k, list(v.shape)] for k, v in model.state_dict().items()]
    struct["yolov5s_nc80"] = keys_of(rmodel.Model(cfg_path))
    struct["yolov5s_nc80_extra"] = {"save": list(rmodel.Model(cfg_path).save), "stride": [8.0, 16.0, 32.0]}
    struct["v3tiny"] = keys_of(fa_model("models.v3tiny", "YOLOv3", 416, rcfg.yolov3_tiny_cfg))
    struct["v2d19"] = keys_of(fa_model("models.v2d19", "YOLOv2", 416, rcfg.yolov2_d19_cfg))
    struct["v2r50"] = keys_of(fa_model("models.v2r50", "YOLOv2", 416, rcfg.yolov2_r50_cfg))
    struct["v2slim"] = keys_of(fa_model("models.v2slim", "YOLOv2", 416, rcfg.yolov2_slim_cfg))
    struct["v3spp"] = keys_of(fa_model("models.v3spp", "YOLOv3", 608, rcfg.yolov3_d53_cfg))
    struct["v3"] = keys_of(fa_model("models.v3", "YOLOv3", 416, rcfg.yolov3_d53_cfg))
    struct["v4"] = keys_of(fa_model("models.v4", "YOLOv4", 608, rcfg.yolov4_cfg))
    struct["notes"] = notes
    if "g8_structure".startswith(args.only):
        with open(os.path.join(args.out, "g8_structure.json"), "w") as fh:
            json.dump(struct, fh)

    # ================= G9: the author's own YOLOv5 re-creations (models/v5.py, models/v5v2.py) =========
    # blocks with a post-concat BatchNorm + LeakyReLU and with weight-shared repeated sub-blocks
    import models.v5 as rv5
    import models.v5v2 as rv5v2
    block_case("g9_a_cspresunit", fa_mods.CSPResUnit(32, 32, 2), x32)
    block_case("g9_a_cspmodule", fa_mods.CSPModule(64, 64, n=3, e=[0.33, 0.5]), x32)
    block_case("g9_a_focus", fa_mods.Focus(3, 32, e=0.5), rnd("g9focus", 2, 3, 16, 12))
    block_case("g9_b_convblock", rv5v2.ConvBlock(16, 32, 3, 2), x16)
    block_case("g9_b_cspresunit", rv5v2.CSPResUnit(32, 32, 3), x32)
    block_case("g9_b_cspmodule1", rv5v2.CSPModule1(64, 64, n=3, e=[0.33, 0.5]), x32)
    block_case("g9_b_cspmodule2", rv5v2.CSPModule2(64, 64, e=0.5, n=2), x32)
    block_case("g9_b_sppmodule", rv5v2.SPPModule(64, 64, e=0.5), rnd("g9spp", 2, 32, 20, 20))
    block_case("g9_b_focus", rv5v2.Focus(3, 32, e=0.5), rnd("g9focus2", 2, 3, 16, 12))
    anchors9 = [[10, 13], [16, 30], [33, 23], [30, 61], [62, 45], [59, 119], [116, 90], [156, 198], [373, 326]]
    own = {"v5": rv5.YOLOv5(nc=20), "v5v2": rv5v2.YOLOv5(nc=20, anchors=anchors9, input_size=128)}
    struct9 = {}
    for tag, net in own.items():
        init_deterministic(net, seed=0)
        net.eval()
        struct9[tag] = keys_of(net)
        x = images(2, 128, seed=129, kind="randn")
        with torch.no_grad():
            ys, ym, yl = net(x)
        save(f"g9_{tag}_128", x_seed=129, ys=ys.numpy(), ym=ym.numpy(), yl=yl.numpy())
    if "g9_structure".startswith(args.only):
        with open(os.path.join(args.out, "g9_structure.json"), "w") as fh:
            json.dump(struct9, fh)
    print("notes:", notes)


if __name__ == "__main__":
    main()

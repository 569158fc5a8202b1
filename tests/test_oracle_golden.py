"""The oracle (oracle/) against the golden vectors captured from the imported reference
(tools/gen_golden.py).  CPU only.  Tolerances: integer / index results exact; float32 results to a
few ulp (numpy vs torch transcendental implementations differ in the last bits)."""
import json

import numpy as np
import pytest
import torch

from helpers import FA_SPECS, build_fa, build_v5s, fa_thresholds, oracle_fa_heads, rel_err, v5s_cfg
from oracle import metrics as OM
from oracle import nets as ON
from oracle import postproc as O
from yolo_amd.synthetic import clustered_predictions, images


# ---- G1: Detect decode ---------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,transposed", [("official", False), ("compat", True)])
@pytest.mark.parametrize("nc", [3, 80])
def test_detect_decode(golden, tag, transposed, nc):
    g = golden(f"g1_decode_{tag}_nc{nc}")
    z, xs = O.detect_decode([g["raw0"], g["raw1"], g["raw2"]], g["stride"], g["anchor_grid"], 3, nc + 5, transposed)
    assert z.shape == g["z"].shape
    np.testing.assert_allclose(z, g["z"], rtol=2e-6, atol=1e-6)
    np.testing.assert_array_equal(xs[0], g["x0"])


def test_transposed_grid_differs(golden):
    a, b = golden("g1_decode_official_nc3"), golden("g1_decode_compat_nc3")
    np.testing.assert_array_equal(a["raw0"], b["raw0"] * 0 + a["raw0"])
    assert np.abs(a["z"][..., :2] - b["z"][..., :2]).max() > 1.0  # the two Detect variants really disagree


# ---- G2 / G6: family-A decode from recorded head maps, and oracle networks ---------------------------
FA_CASES = [("v3tiny", 64), ("v3tiny", 416), ("v2d19", 128), ("v2d19", 416), ("v3spp", 128), ("v3", 128), ("v4", 128),
            ("v2r50", 128), ("v2r50", 416), ("v2slim", 128), ("v2slim", 416)]


@pytest.mark.parametrize("tag,size", FA_CASES)
def test_family_a_decode(golden, tag, size):
    g = golden(f"g6_{tag}_{size}")
    net = build_fa(tag, size)
    heads = [g["head_" + h] for h in FA_SPECS[tag][3]]
    anchors = [np.array(a, np.float32).reshape(-1, 2) for a in net._level_anchors()]
    bb, sc = O.fa_decode(heads, anchors, net.num_anchors, 20, FA_SPECS[tag][4], size)
    np.testing.assert_allclose(bb[0], g["pre_bboxes"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(sc[0], g["pre_scores"], rtol=1e-5, atol=1e-7)


def assert_same_detections(bb, sc, cl, g, scores_in):
    """Exact equality; when the best-class scores contain exact ties the visiting order of the reference's
    numpy argsort is implementation-defined (utils/utils.py:22), so the rows are compared as a multiset."""
    best = np.asarray(scores_in).max(1)
    if len(np.unique(best)) == len(best):
        np.testing.assert_array_equal(bb, g["out_bboxes"])
        np.testing.assert_array_equal(sc, g["out_scores"])
        np.testing.assert_array_equal(cl, g["out_cls"])
        return
    def rows(b, s, c):
        r = np.concatenate([np.asarray(c, np.float64)[:, None], np.asarray(s, np.float64)[:, None], np.asarray(b, np.float64)], 1)
        return r[np.lexsort(r.T[::-1])]
    np.testing.assert_array_equal(rows(bb, sc, cl), rows(g["out_bboxes"], g["out_scores"], g["out_cls"]))


@pytest.mark.parametrize("tag,size", FA_CASES)
def test_family_a_postprocess_on_recorded_boxes(golden, tag, size):
    g = golden(f"g6_{tag}_{size}")
    conf, thr = fa_thresholds(tag)
    bb, sc, cl = O.postprocess(g["pre_bboxes"], g["pre_scores"], conf, 20, thr)
    assert_same_detections(bb, sc, cl, g, g["pre_scores"])


@pytest.mark.parametrize("tag,size", [("v3tiny", 64), ("v2d19", 128), ("v3spp", 128), ("v3", 128), ("v4", 128),
                                      ("v2r50", 128), ("v2slim", 128)])
def test_oracle_networks(golden, tag, size):
    g = golden(f"g6_{tag}_{size}")
    net = build_fa(tag, size)
    x = images(1, size, seed=int(g["x_seed"]), kind="randn")
    with torch.no_grad():
        heads = oracle_fa_heads(tag, net.state_dict(), x)
    for h, name in zip(heads, FA_SPECS[tag][3]):
        assert rel_err(h.numpy(), g["head_" + name]) < 2e-5, name


def test_oracle_v3spp_608_samples(golden):
    g = golden("g6_v3spp_608")
    net = build_fa("v3spp", 608)
    x = images(1, 608, seed=608, kind="randn")
    with torch.no_grad():
        heads = oracle_fa_heads("v3spp", net.state_dict(), x)
    step = int(g["step"])
    for h, name in zip(heads, ["pred_1", "pred_2", "pred_3"]):
        assert rel_err(h.numpy().reshape(-1)[::step], g["head_" + name]) < 5e-5


@pytest.mark.parametrize("nc", [80, 20])
def test_oracle_yolov5s(golden, nc):
    g = golden(f"g6_v5s_nc{nc}_64")
    m = build_v5s(nc)
    x = images(2, 64, seed=64)
    with torch.no_grad():
        heads = ON.yolov5_heads(m.state_dict(), x, v5s_cfg())
    z, xs = O.detect_decode([h.numpy() for h in heads], g["stride"], g["anchor_grid"].reshape(3, 3, 2), 3, nc + 5)
    for i in range(3):
        assert rel_err(xs[i], g[f"x{i}"]) < 2e-5
    np.testing.assert_allclose(z, g["z"], rtol=2e-4, atol=2e-4)


# ---- G3: postprocess / nms ---------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [0, 1, 2])
def test_postprocess(golden, case):
    g = golden(f"g3_postprocess_{case}")
    bb, sc, cl = O.postprocess(g["bb"], g["scores"], float(g["conf"]), int(g["nc"]), float(g["thr"]))
    np.testing.assert_array_equal(bb, g["out_bb"])
    np.testing.assert_array_equal(sc, g["out_scores"])
    np.testing.assert_array_equal(cl, g["out_cls"])
    keep = O.nms_numpy(g["bb"], g["scores"][:, 0], float(g["thr"]))
    np.testing.assert_array_equal(np.array(keep), g["nms_keep_class0_scores"])


# ---- G4: non_max_suppression -----------------------------------------------------------------------------
def _g4_pred(g, tag):
    if "pred" in g:
        return g["pred"]
    n, nb, no, pct = [int(v) for v in g["pred_args"]]
    return clustered_predictions(n, nb, no - 5, seed=int(g["pred_seed"]), obj_pass=pct / 100.0).numpy()


@pytest.mark.parametrize("tag", ["small_ml", "small_best", "small_agnostic", "few", "big_ml", "big_best"])
def test_non_max_suppression(golden, tag):
    g = golden(f"g4_nms_{tag}")
    kw = json.loads(str(g["kw"]))
    pred = _g4_pred(g, tag)
    # (1) the candidate stage: exactly what reaches torchvision.ops.nms
    call = 0
    for i in range(int(g["n_img"])):
        x6, boxes, scores = O.nms_candidates(pred[i], kw["conf_thres"], kw.get("multi_label", False), None,
                                             kw.get("max_det", 300), kw.get("agnostic", False))
        if len(x6):
            np.testing.assert_array_equal(boxes, g[f"nms_boxes{call}"])
            np.testing.assert_array_equal(scores, g[f"nms_scores{call}"])
            call += 1
    assert call == int(g["n_calls"])
    # (2) end to end (the greedy core is the oracle's own on both sides: "parity unpinned" for it)
    out = O.non_max_suppression(pred, **kw)
    for i, o in enumerate(out):
        np.testing.assert_array_equal(o, g[f"out{i}"])


def test_torchvision_nms_matches_family_nms_away_from_boundaries(golden):
    """Cross-check of the unpinned restatement against the reference's in-tree numpy nms
    (yolov5_official/utils.py:551-578 == utils/utils.py:15-42): on boxes with positive overlap the two differ
    only in `>` vs `<=`, so with no IoU exactly at the threshold the kept sets agree."""
    g = golden("g3_postprocess_1")
    bb, sc = g["bb"][:400], g["scores"][:400, 0]
    a = sorted(int(i) for i in O.torchvision_nms(bb, sc, 0.5))
    # the eps floor only matters for disjoint boxes (ovr ~ 0 either way)
    b = sorted(O.nms_numpy(bb, sc, 0.5))
    assert a == b


# ---- G7: metric helpers ----------------------------------------------------------------------------------
def test_v5_metric_helpers(golden):
    g = golden("g7_v5_metrics")
    np.testing.assert_allclose(O.box_iou(g["box1"], g["box2"]), g["iou"], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(O.xywh2xyxy(g["xywh"]), g["xyxy"])
    np.testing.assert_array_equal(OM.process_batch(g["preds"], g["labels"], g["iouv"]), g["correct"])
    p, r, ap, f1, cls = OM.ap_per_class(g["tp"], g["conf"], g["pcls"], g["tcls"])
    np.testing.assert_allclose(ap, g["ap"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(p, g["p"], rtol=1e-12)
    np.testing.assert_allclose(r, g["r"], rtol=1e-12)
    np.testing.assert_array_equal(cls, g["ap_class"])
    one = OM.compute_ap(g["rec"], g["prec"])
    np.testing.assert_allclose(one[0], g["ap_one"], rtol=1e-12)


def test_voc_metric(golden):
    g = golden("g7_voc")
    pr = golden("g7_v5_metrics")   # the recall / precision arrays the reference's voc_ap was evaluated on
    assert abs(OM.voc_ap(pr["rec"], pr["prec"], True) - float(g["voc_ap07"])) < 1e-12
    assert abs(OM.voc_ap(pr["rec"], pr["prec"], False) - float(g["voc_ap_area"])) < 1e-12
    gts = json.loads(str(g["gts"]))
    names = [str(n) for n in g["names"]]
    labelmap = [str(c) for c in g["labelmap"]]
    for ci, cname in enumerate(labelmap):
        lines = []
        for ii, n in enumerate(names):
            d = g[f"boxes_{ci}_{ii}"]
            for k in range(d.shape[0]):
                lines.append("{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}\n".format(n, d[k, -1], d[k, 0] + 1, d[k, 1] + 1,
                                                                           d[k, 2] + 1, d[k, 3] + 1))
        recs, npos = {}, 0
        for n in names:
            R = [o for o in gts[n] if o["name"] == cname]
            diff = np.array([o["difficult"] for o in R]).astype(bool)
            npos += int(sum(~diff))
            recs[n] = {"bbox": np.array([o["bbox"] for o in R]), "difficult": diff, "det": [False] * len(R)}
        ap = OM.voc_eval_lines(lines, recs, npos, 0.5, True)[2]
        assert abs(ap - float(g["aps"][ci])) < 1e-12, cname
    assert float(g["aps"][3]) == -1.0  # the undetected class contributes -1 to the mean


# ---- G10: pre-processing (letterbox / base_transform) ---------------------------------------------------
def test_resize_known_answers():
    """Hand-computed values of the 11-bit fixed-point bilinear arithmetic, plus its invariants."""
    from oracle import preproc as OP
    row = np.array([[[0, 0, 0], [255, 255, 255]]], np.uint8)
    # scale 0.5: samples at -0.25 (edge: 0), 0.25 (1536/512), 0.75 (512/1536), 1.25 (edge: 255)
    np.testing.assert_array_equal(OP.resize_linear_u8(row, (4, 1))[0, :, 0], [0, 64, 191, 255])
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(12, 10, 3), dtype=np.uint8)
    np.testing.assert_array_equal(OP.resize_linear_u8(img, (10, 12)), img)                      # same size: copy
    box = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    np.testing.assert_array_equal(OP.resize_linear_u8(img, (5, 6)), box)                        # exact half: box mean
    flat = np.full((7, 9, 3), 201, np.uint8)
    for d in ((20, 31), (4, 3), (9, 30)):
        assert (OP.resize_linear_u8(flat, d) == 201).all()                                      # constants survive
    up = OP.resize_linear_u8(img, (23, 29))
    assert up.shape == (29, 23, 3) and up.min() >= img.min() and up.max() <= img.max()


def test_preproc_geometry_and_pixels_golden(golden):
    from oracle import preproc as OP
    from yolo_amd import preprocess as P
    g = golden("g10_preproc")
    for i in range(int(g["n_lb"])):
        img, kw = g[f"lb{i}_img"], json.loads(str(g[f"lb{i}_kw"]))
        new_shape = tuple(int(v) for v in g[f"lb{i}_new_shape"])
        for geo in (OP.letterbox_geometry, P.letterbox_geometry):      # the checker and the product's host logic
            new_unpad, ratio, pad, top, bottom, left, right = geo(img.shape[:2], new_shape, **kw)
            if tuple(g[f"lb{i}_resize"]) != (-1, -1):
                assert tuple(new_unpad) == tuple(g[f"lb{i}_resize"]), i
            else:
                assert tuple(new_unpad) == img.shape[:2][::-1], i
            assert (top, bottom, left, right) == tuple(g[f"lb{i}_border"]), i
            np.testing.assert_array_equal(np.array(ratio, np.float64), g[f"lb{i}_ratio"])
            np.testing.assert_array_equal(np.array(pad, np.float64), g[f"lb{i}_pad"])
        out, ratio, pad = OP.letterbox(img, new_shape, **kw)
        np.testing.assert_array_equal(out, g[f"lb{i}_out"])
    for i in range(int(g["n_bt"])):
        x = OP.base_transform(g[f"bt{i}_img"], int(g[f"bt{i}_size"]), g[f"bt{i}_mean"], g[f"bt{i}_std"])
        np.testing.assert_array_equal(x, g[f"bt{i}_out"])


# ---- the alternative NMS flavours (yolov5_official/utils.py:318-548) ---------------------------------------------------
NMS2_EXACT = ("hard_nms_batch", "hard_nms", "hard_nms_myself", "and", "diou_nms")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_non_max_suppression2_golden(golden, tag):
    """Every method of non_max_suppression2 against the reference's outputs: selections, order, scores and classes
    exact; 'merge' coordinates (a masked sum whose order torch leaves open) and 'soft_nms' scores (exp) to 1e-6."""
    g = golden(f"g12_nms2_{tag}")
    kw = json.loads(str(g["kw"]))
    for method in O.NMS2_METHODS:
        out = O.non_max_suppression2(g["pred"], method=method, **kw)
        for i, o in enumerate(out):
            ref = g[f"{method}_{i}"]
            o = np.zeros((0, 6), np.float32) if o is None else o
            assert o.shape == ref.shape, (method, i)
            if method in NMS2_EXACT:
                np.testing.assert_array_equal(o, ref, err_msg=f"{method} {i}")
            else:
                np.testing.assert_array_equal(o[:, 5], ref[:, 5])
                np.testing.assert_allclose(o, ref, rtol=2e-6, atol=1e-6, err_msg=f"{method} {i}")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_diou_non_max_suppression_golden(golden, tag):
    """Cluster-Weighted DIoU-NMS: kept slots, scores and classes exact (padding rows included: NaN box, zero score);
    the weighted coordinates to 1e-6 relative (a matmul whose summation order is the library's)."""
    g = golden(f"g12_diou_{tag}")
    kw = json.loads(str(g["kw"]))
    out = O.diou_non_max_suppression(g["pred"], **kw)
    for i, o in enumerate(out):
        ref = g[f"out{i}"]
        assert o.shape == ref.shape
        np.testing.assert_array_equal(o[:, 4:], ref[:, 4:])
        np.testing.assert_array_equal(np.isnan(o), np.isnan(ref))
        np.testing.assert_allclose(np.nan_to_num(o[:, :4]), np.nan_to_num(ref[:, :4]), rtol=2e-6, atol=1e-4)


def test_bf16_storage_floor_is_off_by_default_and_small():
    """oracle.nets.bf16_storage: outside the context the networks are the plain fp32 ones (bit for bit before and after a
    use of it); inside, weights and activations are rounded to bf16 and the head maps move by a fraction of a percent on
    a small input -- the floor the bf16 GPU path is measured against."""
    import torch
    from helpers import build_fa, build_v5s, oracle_fa_heads, rel_l2, v5s_cfg
    from oracle import nets as ON
    from yolo_amd.synthetic import images
    m = build_v5s(20)
    sd = {k: v.detach().float() for k, v in m.state_dict().items()}
    x = images(1, 64, seed=64)
    with torch.no_grad():
        a = ON.yolov5_heads(sd, x, v5s_cfg())
        with ON.bf16_storage():
            f = ON.yolov5_heads(sd, x, v5s_cfg())
        b = ON.yolov5_heads(sd, x, v5s_cfg())
    for u, v, w in zip(a, b, f):
        assert torch.equal(u, v)
        assert 1e-4 < rel_l2(w.numpy(), u.numpy()) < 2e-2
    for tag in ("v4", "v2r50", "v3spp"):
        net = build_fa(tag, 64)
        sd = {k: v.detach().float() for k, v in net.state_dict().items()}
        x = images(1, 64, seed=3, kind="randn")
        with torch.no_grad():
            a = oracle_fa_heads(tag, sd, x)
            with ON.bf16_storage():
                f = oracle_fa_heads(tag, sd, x)
        for u, w in zip(a, f):
            assert 1e-4 < rel_l2(w.numpy(), u.numpy()) < 3e-2, tag

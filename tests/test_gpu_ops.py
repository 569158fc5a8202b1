"""GPU parity tests of the individual libydet.so operators, called through the C ABI (ctypes).

Bars: index / integer results bit-exact against the oracle and the golden vectors; float32 decode within a
few ulp-level tolerances (device vs host exp); convolution within 1e-3 (fp32 path) / 2e-2 (bf16 path) of a
plain PyTorch fp32 reference, relative to the largest reference magnitude.
"""
import json

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import rel_err
from oracle import postproc as O
from yolo_amd import _ffi, engine, ops
from yolo_amd.synthetic import _gen, clustered_predictions

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(t):  # NCHW numpy/torch -> NHWC CUDA fp32 tensor
    return torch.as_tensor(t).permute(0, 2, 3, 1).contiguous().to(DEV)


# ---- decode ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,transposed", [("official", False), ("compat", True)])
@pytest.mark.parametrize("nc", [3, 80])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_decode_v5_golden(golden, tag, transposed, nc, dtype):
    g = golden(f"g1_decode_{tag}_nc{nc}")
    raws = [g["raw0"], g["raw1"], g["raw2"]]
    heads = [nhwc(r).to(dtype) for r in raws]
    z = ops.decode_v5(heads, g["stride"].tolist(), g["anchor_grid"].reshape(3, -1).tolist(), 3, nc + 5, transposed)
    if dtype == torch.float32:
        np.testing.assert_allclose(z.cpu().numpy(), g["z"], rtol=3e-6, atol=2e-6)
    else:  # bf16 heads: compare with the oracle fed the same rounded inputs
        ref, _ = O.detect_decode([h.float().permute(0, 3, 1, 2).cpu().numpy() for h in heads], g["stride"],
                                 g["anchor_grid"], 3, nc + 5, transposed)
        np.testing.assert_allclose(z.cpu().numpy(), ref, rtol=3e-6, atol=2e-6)


def test_decode_v5_padded_ld():
    """Head buffers with a padded channel stride (ld 256 for 255 channels) decode like dense ones."""
    g = _gen(3, "ld")
    raw = [torch.randn(2, 255, s, s, generator=g) for s in (8, 4, 2)]
    dense = [nhwc(r) for r in raw]
    padded = []
    for d in dense:
        buf = torch.zeros(*d.shape[:3], 256, device=DEV)
        buf[..., :255] = d
        padded.append(buf[..., :255])
    strides, anchors = [8., 16., 32.], [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
    a = ops.decode_v5(dense, strides, anchors, 3, 85)
    b = ops.decode_v5(padded, strides, anchors, 3, 85)
    assert torch.equal(a, b)


@pytest.mark.parametrize("nc,dtype,na", [(80, torch.float32, 3), (3, torch.float32, 3), (200, torch.float32, 3),
                                         (80, torch.bfloat16, 3), (20, torch.float32, 2), (7, torch.float32, 5)])
def test_decode_v5_row_max_and_nms_shortcut(nc, dtype, na):
    """decode_v5 also emits per box max_c(cls_c * obj): exact against torch on the decoded tensor, and NMS fed
    with it returns bit-identical results to NMS that scans the tensor itself.  nc=200 exercises the
    wide-row (no > 128) kernel variant, na != 3 the generic anchor loop; the maps are non-square."""
    g = _gen(5, f"rowmax{nc}")
    no = nc + 5
    shapes = [(12, 10), (6, 5), (3, 3)]
    raw = [torch.randn(3, na * no, h, w, generator=g) * 2 - 1 for h, w in shapes]
    heads = [nhwc(r).to(dtype) for r in raw]
    strides = [8., 16., 32.]
    anchors = (torch.rand(3, 2 * na, generator=g) * 100 + 5).tolist()
    z = ops.decode_v5(heads, strides, anchors, na, no)
    ref, _ = O.detect_decode([h.float().permute(0, 3, 1, 2).cpu().numpy() for h in heads], strides,
                             np.array(anchors, np.float32).reshape(3, na, 2), na, no)
    np.testing.assert_allclose(z.cpu().numpy(), ref, rtol=3e-6, atol=2e-6)
    rm = ops.row_max_of(z)
    assert rm is not None and rm.shape == z.shape[:2]
    expect = (z[..., 5:] * z[..., 4:5]).max(-1).values
    assert torch.equal(rm, expect)
    for kw in (dict(conf_thres=0.001, iou_thres=0.6, multi_label=True), dict(conf_thres=0.25, iou_thres=0.45),
               dict(conf_thres=0.05, iou_thres=0.5, multi_label=True, max_det=50)):
        d1, c1 = ops.nms_v5(z, **kw)                    # shortcut: row maxima from the decode kernel
        d2, c2 = ops.nms_v5(z, row_max=None, **kw)      # full scan
        assert torch.equal(c1, c2) and torch.equal(d1, d2)
        want = O.non_max_suppression(z.cpu().numpy(), **kw)
        for i, w in enumerate(want):
            np.testing.assert_array_equal(d1[i, : int(c1[i])].cpu().numpy(), w)
    # the shortcut is dropped as soon as the tensor is modified or replaced
    z2 = z.clone()
    assert ops.row_max_of(z2) is None
    z[0, 0, 4] = 0.5
    assert ops.row_max_of(z) is None


@pytest.mark.parametrize("tag,size,v2", [("v3tiny", 64, False), ("v3tiny", 416, False), ("v2d19", 128, True),
                                         ("v2d19", 416, True), ("v3spp", 128, True), ("v3", 128, False),
                                         ("v4", 128, True)])
def test_decode_v2v3_golden(golden, tag, size, v2):
    from helpers import FA_SPECS, build_fa
    g = golden(f"g6_{tag}_{size}")
    net = build_fa(tag, size)
    heads = [nhwc(g["head_" + h]) for h in FA_SPECS[tag][3]]
    boxes, scores = ops.decode_v2v3(heads, net._level_anchors(), net.num_anchors, 20, 0 if v2 else 1, float(size))
    np.testing.assert_allclose(boxes[0].cpu().numpy(), g["pre_bboxes"], rtol=2e-5, atol=3e-6)
    np.testing.assert_allclose(scores[0].cpu().numpy(), g["pre_scores"], rtol=2e-5, atol=1e-7)


# ---- YOLOv5 NMS -------------------------------------------------------------------------------------------------
def _g4_pred(g):
    if "pred" in g:
        return torch.from_numpy(g["pred"])
    n, nb, no, pct = [int(v) for v in g["pred_args"]]
    return clustered_predictions(n, nb, no - 5, seed=int(g["pred_seed"]), obj_pass=pct / 100.0)


def _check_nms(pred, out_list, **kw):
    det, cnt = ops.nms_v5(pred.to(DEV), kw.get("conf_thres", 0.25), kw.get("iou_thres", 0.45),
                          multi_label=kw.get("multi_label", False), agnostic=kw.get("agnostic", False),
                          max_det=kw.get("max_det", 300), classes=kw.get("classes"))
    det, cnt = det.cpu().numpy(), cnt.cpu().numpy()
    for i, ref in enumerate(out_list):
        assert cnt[i] == len(ref), f"image {i}: {cnt[i]} detections, expected {len(ref)}"
        np.testing.assert_array_equal(det[i, : cnt[i]], ref)   # bit-exact rows, same order
        assert not det[i, cnt[i]:].any()


@pytest.mark.parametrize("tag", ["small_ml", "small_best", "small_agnostic", "few", "big_ml", "big_best"])
def test_nms_v5_golden(golden, tag):
    g = golden(f"g4_nms_{tag}")
    kw = json.loads(str(g["kw"]))
    _check_nms(_g4_pred(g), [g[f"out{i}"] for i in range(int(g["n_img"]))], **kw)


@pytest.mark.parametrize("case", ["ties_all_equal", "ties_threshold_bin", "flat_classes", "empty", "one", "classes",
                                  "classes_best_many", "classes_ml_many", "nan_scores"])
def test_nms_v5_edge_cases(case):
    g = _gen(11, case)
    kw = dict(conf_thres=0.001, iou_thres=0.6, multi_label=True)
    if case == "ties_all_equal":   # every candidate has the same score: exact top-k path, index order decides
        pred = torch.zeros(2, 3000, 9)
        pred[..., :2] = torch.rand(2, 3000, 2, generator=g) * 600
        pred[..., 2:4] = 30.0
        pred[..., 4] = 0.5
        pred[..., 5:] = 0.5
    elif case == "ties_threshold_bin":   # > 4096 candidates inside one histogram bin around the cut
        pred = clustered_predictions(2, 6000, 4, seed=9, obj_pass=1.0)
        pred[..., 4] = 0.75
        pred[..., 5:] = 0.5 + torch.rand(2, 6000, 4, generator=g) * 1e-4
    elif case == "flat_classes":   # every good row carries many candidates of almost the same score
        pred = clustered_predictions(2, 3000, 20, seed=10, obj_pass=1.0)
        pred[..., 5:] = 0.5 + torch.rand(2, 3000, 20, generator=g) * 1e-3
    elif case == "empty":
        pred = clustered_predictions(3, 500, 4, seed=3)
        pred[..., 4] = 0.0005
    elif case == "one":
        pred = clustered_predictions(2, 64, 1, seed=4, obj_pass=0.5)   # nc == 1 disables multi_label
    elif case == "classes":
        pred = clustered_predictions(2, 800, 6, seed=5)
        kw = dict(conf_thres=0.2, iou_thres=0.45, classes=[1, 4])
    elif case in ("classes_best_many", "classes_ml_many"):
        # a class filter with far more than max_det rows above conf: in best-class mode a row whose best class is
        # filtered out has no candidate even if one of the kept classes scores above conf (yolov5_official/utils.py:
        # 160-166), and must not be counted when the pre-selection sizes its score cut
        pred = clustered_predictions(2, 6000, 6, seed=12, obj_pass=0.9)
        pred[..., 5:] = torch.rand(2, 6000, 6, generator=g)
        kw = dict(conf_thres=0.05, iou_thres=0.45, classes=[1, 4], multi_label=case == "classes_ml_many")
        ref = O.non_max_suppression(pred.numpy(), **kw)
        assert all(len(r) > 20 for r in ref)
    else:   # NaN / inf inputs must not hang or crash; compared to the oracle on the finite image only
        pred = clustered_predictions(2, 400, 3, seed=6)
        pred[1, ::7, 4] = float("nan")
        ref = O.non_max_suppression(pred[:1].numpy(), **kw)
        det, cnt = ops.nms_v5(pred.to(DEV), 0.001, 0.6, multi_label=True)
        np.testing.assert_array_equal(det[0, : int(cnt[0])].cpu().numpy(), ref[0])
        return
    ref = O.non_max_suppression(pred.numpy(), **kw)
    _check_nms(pred, ref, **kw)


def test_nms_v5_full_size_properties():
    """BASELINE size (64 x 25200 x 85): properties that do not need the oracle at full size --
    idempotence on its own output, descending scores, count bound -- plus oracle equality on two images."""
    pred = clustered_predictions(64, 25200, 80, seed=21, obj_pass=0.3)
    det, cnt = ops.nms_v5(pred.to(DEV), 0.001, 0.6, multi_label=True)
    det_h, cnt_h = det.cpu().numpy(), cnt.cpu().numpy()
    assert (cnt_h <= 300).all() and (cnt_h > 0).all()
    for i in range(64):
        s = det_h[i, : cnt_h[i], 4]
        assert (np.diff(s) <= 0).all()
    ref = O.non_max_suppression(pred[:2].numpy(), 0.001, 0.6, multi_label=True)
    for i in range(2):
        np.testing.assert_array_equal(det_h[i, : cnt_h[i]], ref[i])
    # idempotence: feed the survivors back as single-class-score predictions; nothing more is removed
    again = torch.zeros(64, 300, 85)
    xyxy = torch.from_numpy(det_h[..., :4])
    again[..., 0] = (xyxy[..., 0] + xyxy[..., 2]) / 2
    again[..., 1] = (xyxy[..., 1] + xyxy[..., 3]) / 2
    again[..., 2] = xyxy[..., 2] - xyxy[..., 0]
    again[..., 3] = xyxy[..., 3] - xyxy[..., 1]
    again[..., 4] = 1.0
    cls = torch.from_numpy(det_h[..., 5]).long()
    again.scatter_(2, (cls + 5).unsqueeze(-1), torch.from_numpy(det_h[..., 4:5]))
    det2, cnt2 = ops.nms_v5(again.to(DEV), 0.0009, 0.6, multi_label=True)
    # xywh round trip moves coordinates by an ulp, so allow the rare boundary flip
    assert (cnt2.cpu().numpy() >= cnt_h - 1).all() and (cnt2.cpu().numpy() <= cnt_h).all()


# ---- family-A NMS -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [0, 1, 2])
def test_nms_perclass_golden(golden, case):
    g = golden(f"g3_postprocess_{case}")
    bb, sc = torch.from_numpy(g["bb"]).to(DEV)[None], torch.from_numpy(g["scores"]).to(DEV)[None]
    keep, cls, best = ops.nms_perclass(bb, sc, float(g["conf"]), float(g["thr"]))
    k = keep[0].bool().cpu().numpy()
    np.testing.assert_array_equal(g["bb"][k], g["out_bb"])
    np.testing.assert_array_equal(best[0].cpu().numpy()[k], g["out_scores"])
    np.testing.assert_array_equal(cls[0].cpu().numpy()[k].astype(np.int64), g["out_cls"])


def test_nms_perclass_batch_and_wrappers(golden):
    from yolo_amd.utils import utils as fa
    gs = [golden(f"g3_postprocess_{c}") for c in (0, 1)]
    n = 600
    bb = torch.stack([torch.from_numpy(g["bb"][:n]) for g in gs]).to(DEV)
    sc = torch.stack([torch.from_numpy(g["scores"][:n]) for g in gs]).to(DEV)
    keep, cls, best = ops.nms_perclass(bb, sc, 0.001, 0.5)
    for i, g in enumerate(gs):
        rb, rs, rc = O.postprocess(g["bb"][:n], g["scores"][:n], 0.001, 20, 0.5)
        k = keep[i].bool().cpu().numpy()
        np.testing.assert_array_equal(g["bb"][:n][k], rb)
        np.testing.assert_array_equal(cls[i].cpu().numpy()[k], rc)
    # numpy-in / numpy-out API of the reference
    g = gs[0]
    ob, osc, oc = fa.postprocess(g["bb"], g["scores"], float(g["conf"]), int(g["nc"]), float(g["thr"]))
    np.testing.assert_array_equal(ob, g["out_bb"])
    np.testing.assert_array_equal(oc, g["out_cls"])
    assert oc.dtype == np.int64 and ob.dtype == np.float32
    assert fa.nms(g["bb"], g["scores"][:, 0], float(g["thr"])) == [int(i) for i in g["nms_keep_class0_scores"]]


def test_nms_perclass_single_class_large():
    """All boxes in one class (longest serial chain) against the oracle."""
    g = _gen(5, "one-class")
    n = 3000
    c = torch.rand(n, 2, generator=g)
    wh = torch.rand(n, 2, generator=g) * 0.1 + 0.01
    bb = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, 1)
    sc = torch.rand(n, 1, generator=g)
    keep, _, _ = ops.nms_perclass(bb.to(DEV)[None], sc.to(DEV)[None], 0.0, 0.5)
    ref = sorted(O.nms_numpy(bb.numpy(), sc[:, 0].numpy(), 0.5))
    assert np.nonzero(keep[0].cpu().numpy())[0].tolist() == ref


def _skewed_classes(seed, N, n, nc, heavy):
    """Boxes of which most fall in the first len(heavy) classes (fractions `heavy`), with repeated boxes and tied scores."""
    g = _gen(seed, "skewed-classes")
    c = torch.rand(N, n, 2, generator=g)
    wh = torch.rand(N, n, 2, generator=g) * 0.12 + 0.01
    bb = torch.cat([c - wh / 2, c + wh / 2], 2).clamp(0, 1)
    bb[:, 100:200] = bb[:, 300:400]                       # identical boxes
    sc = torch.rand(N, n, nc, generator=g) * 0.3
    u = torch.rand(N, n, generator=g)
    top = torch.rand(N, n, generator=g) * 0.5 + 0.4
    top = (top * 64).round() / 64                         # many equal scores
    lo = 0.0
    for k, f in enumerate(heavy):
        m = (u >= lo) & (u < lo + f)
        sc[..., k][m] = top[m]
        lo += f
    return bb, sc


@pytest.mark.parametrize("workspace", [None, "min", "one"])
def test_nms_perclass_long_segments(workspace):
    """Class segments of thousands of boxes: the bit-matrix path (full workspace), the round-by-round walk (minimal
    workspace) and a workspace with room for the first long segment of an image only -- all equal to the oracle."""
    N, n, nc = 3, 6000, 12
    bb, sc = _skewed_classes(11, N, n, nc, (0.5, 0.3))
    lib = _ffi.lib()
    ws = workspace
    best_h, cls_h = sc.max(2)
    counts_h = np.stack([np.bincount(cls_h[i][best_h[i] >= 0.05].numpy(), minlength=nc) for i in range(N)])
    words = lambda cnt: 64 * (((cnt + 63) // 64) * ((cnt + 63) // 64 + 1) // 2)      # upper triangle, 64-row strips
    assert counts_h[:, 0].min() > 2500 and counts_h[:, 1].min() > 1300        # both sides of the 256-box rule are hit
    assert 0 < counts_h[:, 2:].max() < 256
    if workspace == "one":      # room for the matrix of the first class of every image and not for the second one's
        room = max(words(int(c)) for c in counts_h[:, 0])
        assert all(room - words(int(c0)) < words(int(c1)) for c0, c1 in counts_h[:, :2])
        ws = lib.ydet_nms_perclass_workspace_bytes_min(N, n, nc) + N * 8 * room
    keep, cls, best = ops.nms_perclass(bb.to(DEV), sc.to(DEV), 0.05, 0.45, workspace_bytes=ws)
    for i in range(N):
        rb, rs, rc = O.postprocess(bb[i].numpy(), sc[i].numpy(), 0.05, nc, 0.45)
        k = keep[i].bool().cpu().numpy()
        assert k.sum() == len(rb)
        np.testing.assert_array_equal(bb[i].numpy()[k], rb)
        np.testing.assert_array_equal(best[i].cpu().numpy()[k], rs)
        np.testing.assert_array_equal(cls[i].cpu().numpy()[k], rc)


def test_nms_perclass_segment_not_multiple_of_64_all_kept():
    """Disjoint boxes (nothing suppressed: the longest walk) in one class, lengths around the block size, the matrix
    threshold (256) and both ends of the LDS sort's range (512 .. 8192)."""
    for n in (255, 256, 257, 319, 320, 511, 512, 513, 1025, 2049, 8192, 8193):
        ix = torch.arange(n, dtype=torch.float32)
        bb = torch.stack([ix * 2, torch.zeros(n), ix * 2 + 1, torch.ones(n)], 1)
        sc = torch.rand(n, 1, generator=_gen(n, "all-kept")) + 0.1
        keep, _, _ = ops.nms_perclass(bb.to(DEV)[None], sc.to(DEV)[None], 0.0, 0.5)
        assert int(keep.sum()) == n
        # and every box identical: only the best one survives
        keep, _, _ = ops.nms_perclass(bb[:1].repeat(n, 1).to(DEV)[None], sc.to(DEV)[None], 0.0, 0.5)
        k = np.nonzero(keep[0].cpu().numpy())[0].tolist()
        assert k == sorted(O.nms_numpy(bb[:1].repeat(n, 1).numpy(), sc[:, 0].numpy(), 0.5))


def test_box_iou(golden):
    g = golden("g7_v5_metrics")
    iou = ops.box_iou(torch.from_numpy(g["box1"]).to(DEV), torch.from_numpy(g["box2"]).to(DEV))
    np.testing.assert_allclose(iou.cpu().numpy(), g["iou"], rtol=1e-6, atol=1e-7)


# ---- the alternative NMS flavours (yolov5_official/utils.py:318-548) ---------------------------------------------------
NMS2_EXACT = ("hard_nms_batch", "hard_nms", "hard_nms_myself", "and", "diou_nms")


def _check_nms2(pred, kw, method, expect):
    from yolo_amd.yolov5_official.utils import non_max_suppression2
    out = non_max_suppression2(torch.as_tensor(pred).to(DEV), method=method, **kw)
    assert len(out) == len(expect)
    for i, (o, ref) in enumerate(zip(out, expect)):
        ref = np.zeros((0, 6), np.float32) if ref is None else ref
        if o is None:
            assert len(ref) == 0, (method, i)
            continue
        o = o.cpu().numpy()
        assert o.shape == ref.shape, (method, i, o.shape, ref.shape)
        assert (np.diff(o[:, 4]) <= 0).all(), "rows must come in descending score order"
        if method != "merge":
            # the order among (nearly) equal scores is the sort's own in the reference (torch.argsort, unstable): rows
            # are compared as a set, keyed by class and box
            o = o[np.lexsort((o[:, 3], o[:, 2], o[:, 1], o[:, 0], o[:, 5]))]
            ref = ref[np.lexsort((ref[:, 3], ref[:, 2], ref[:, 1], ref[:, 0], ref[:, 5]))]
        if method in NMS2_EXACT:
            np.testing.assert_array_equal(o, ref, err_msg=f"{method} image {i}")
        else:   # 'merge': summation order of the weighted mean; 'soft_nms': exp on the device
            np.testing.assert_array_equal(o[:, 5], ref[:, 5])
            np.testing.assert_allclose(o, ref, rtol=3e-6, atol=1e-4 if method == "merge" else 1e-6, err_msg=f"{method} image {i}")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
@pytest.mark.parametrize("method", list(O.NMS2_METHODS))
def test_nms2_golden(golden, tag, method):
    """ydet_nms2 against the outputs of the reference's non_max_suppression2 (fixtures G12): rows, order, scores and
    classes bit-exact; coordinates of 'merge' and scores of 'soft_nms' to 3e-6."""
    g = golden(f"g12_nms2_{tag}")
    kw = json.loads(str(g["kw"]))
    n = g["pred"].shape[0]
    _check_nms2(g["pred"], kw, method, [g[f"{method}_{i}"] for i in range(n)])


@pytest.mark.parametrize("method", ["hard_nms_myself", "diou_nms", "soft_nms", "and", "merge", "hard_nms_batch"])
def test_nms2_full_size_against_oracle(method):
    """25200 boxes x 80 classes: classes with more than 500 candidates (the 500-row limit), empty images, and the
    multi-class expansion at conf 0.001; the oracle is fed the same tensor."""
    pred = clustered_predictions(2, 25200, 80, seed=31, obj_pass=0.04)
    pred[1, :, 4] = 0.0                               # an image without candidates
    conf = 0.2 if method == "hard_nms_batch" else 0.02
    kw = dict(conf_thres=conf, nms_thres=0.5, multi_cls=True)
    _check_nms2(pred, kw, method, O.non_max_suppression2(pred.numpy(), method=method, **kw))


def test_nms2_edge_cases():
    from yolo_amd.yolov5_official.utils import non_max_suppression2
    pred = clustered_predictions(1, 64, 3, seed=5, img_size=64.0)
    one = pred.clone()
    one[0, 1:, 4] = 0.0                               # a single box: every method returns it untouched (:387-389)
    for method in O.NMS2_METHODS:
        _check_nms2(one, dict(conf_thres=0.01, nms_thres=0.5, multi_cls=False), method,
                    O.non_max_suppression2(one.numpy(), 0.01, 0.5, False, method))
    tiny = pred.clone()
    tiny[0, :, 2:4] = 1.5                             # every box below the 2-pixel limit (:339)
    assert non_max_suppression2(tiny.to(DEV), 0.01, 0.5)[0] is None
    with pytest.raises(ValueError):
        non_max_suppression2(pred.to(DEV), method="nope")
    with pytest.raises(RuntimeError):
        non_max_suppression2(pred, 0.01, 0.5)         # CPU tensor: there is no CPU path


def _check_cluster(pred, kw, expect):
    from yolo_amd.yolov5_official.utils import diou_non_max_suppression
    out = diou_non_max_suppression(torch.as_tensor(pred).to(DEV), **kw)
    for i, (o, ref) in enumerate(zip(out, expect)):
        o = o.cpu().numpy()
        assert o.shape == ref.shape, (i, o.shape, ref.shape)
        np.testing.assert_array_equal(o[:, 4:], ref[:, 4:])                       # kept slots: scores and classes exact
        np.testing.assert_array_equal(np.isnan(o), np.isnan(ref))                 # padding rows carry NaN boxes
        np.testing.assert_allclose(np.nan_to_num(o[:, :4]), np.nan_to_num(ref[:, :4]), rtol=3e-6, atol=2e-4)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_cluster_diou_nms_golden(golden, tag):
    """ydet_nms_cluster_diou against the reference's diou_non_max_suppression (fixtures G12)."""
    g = golden(f"g12_diou_{tag}")
    kw = json.loads(str(g["kw"]))
    _check_cluster(g["pred"], kw, [g[f"out{i}"] for i in range(g["pred"].shape[0])])


def test_cluster_diou_nms_full_size_against_oracle():
    pred = clustered_predictions(2, 25200, 80, seed=33, obj_pass=0.05)
    kw = dict(conf_thres=0.1, iou_thres=0.6, max_box=1500)
    _check_cluster(pred, kw, O.diou_non_max_suppression(pred.numpy(), **kw))


# ---- pooling / resampling / layout against torch fp32 ------------------------------------------------------------
def _run_graph(build, *xs, compute="fp32"):
    g = engine.Graph(DEV, compute)
    build(g, [g.input_nchw(*x.shape) for x in xs])
    return g.finalize().run(*[x.to(DEV) for x in xs])


@pytest.mark.parametrize("compute,tol", [("fp32", 0.0), ("bf16", 8e-3)])
@pytest.mark.parametrize("C", [8, 24, 64])
def test_spatial_ops(compute, tol, C):
    g = _gen(2, f"sp{C}")
    x = torch.randn(2, C, 12, 10, generator=g)
    xs = torch.randn(2, C, 19, 19, generator=g)
    if compute == "bf16":
        x, xs = x.bfloat16().float(), xs.bfloat16().float()

    def build(gr, ins):
        t, ts = ins
        gr.to_nchw(gr.maxpool2(t), "pool")
        gr.to_nchw(gr.maxpool2s1(t), "pools1")
        gr.to_nchw(gr.upsample2x(t, 0), "near")
        gr.to_nchw(gr.upsample2x(t, 1), "bil")
        gr.to_nchw(gr.space_to_depth(t, 0), "reorg")
        gr.to_nchw(gr.space_to_depth(t, 1), "focus")
        gr.to_nchw(gr.spp(ts), "spp")
        gr.to_nchw(gr.cat([gr.maxpool2(t), gr.maxpool2(t)]), "cat")

    out = _run_graph(build, x, xs, compute=compute)
    ref = {
        "pool": F.max_pool2d(x, 2, 2),
        "pools1": F.max_pool2d(F.pad(x, (0, 1, 0, 1)), 2, 1),
        "near": F.interpolate(x, scale_factor=2, mode="nearest"),
        "bil": F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True),
        "focus": torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1),
        "spp": torch.cat([xs] + [F.max_pool2d(xs, k, 1, k // 2) for k in (5, 9, 13)], 1),
    }
    b, c, h, w = x.shape
    ref["reorg"] = x.view(b, c, h // 2, 2, w // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(b, 4 * c, h // 2, w // 2)
    ref["cat"] = torch.cat([ref["pool"], ref["pool"]], 1)
    for k, r in ref.items():
        got = out[k].float().cpu()
        assert got.shape == r.shape, k
        if k == "bil":
            assert rel_err(got, r) <= max(tol, 2e-6), k
        else:
            assert rel_err(got, r) <= tol, k


def test_golden_pool_and_upsample(golden):
    for name, fn in (("g5_a_upsample_bilinear", lambda gr, t: gr.upsample2x(t, 1)),
                     ("g5_a_zeropad_pool", lambda gr, t: gr.maxpool2s1(t)),
                     ("g5_a_reorg", lambda gr, t: gr.space_to_depth(t, 0)),
                     ("g5_a_spp", lambda gr, t: gr.spp(t))):
        g = golden(name)
        x = torch.from_numpy(g["x"])
        out = _run_graph(lambda gr, ins: gr.to_nchw(fn(gr, ins[0]), "y"), x)
        np.testing.assert_allclose(out["y"].cpu().numpy(), g["y"], rtol=2e-6, atol=2e-6, err_msg=name)


# ---- convolution against torch fp32 -------------------------------------------------------------------------------
CONV_CASES = [
    # N, H, W, Cin, Cout, k, s, p, act, residual
    (2, 16, 16, 16, 32, 3, 1, 1, "silu", None),
    (2, 16, 16, 32, 32, 3, 1, 1, "silu", "after"),
    (1, 20, 20, 64, 64, 1, 1, 0, "leaky", None),
    (2, 17, 13, 32, 64, 3, 2, 1, "leaky", None),        # odd sizes, stride 2
    (2, 16, 16, 8, 16, 3, 1, 1, "leaky", None),          # narrow
    (1, 13, 13, 128, 255, 1, 1, 0, "none", None),        # ragged head conv
    (1, 13, 13, 64, 75, 1, 1, 0, "none", None),
    (3, 9, 9, 256, 128, 3, 1, 1, "leaky", "before"),
    (1, 26, 26, 512, 256, 1, 1, 0, "leaky", None),       # deep K
    (1, 8, 8, 24, 160, 3, 1, 1, "silu", None),           # Cin not a multiple of the K-step
    (2, 40, 40, 64, 128, 3, 2, 1, "silu", None),
    (1, 6, 6, 1024, 512, 1, 1, 0, "silu", None),
    (2, 11, 11, 40, 40, 5, 1, 2, "none", None),          # 5x5 kernel
    # grids where the dispatcher picks the 128 x 256 eight-wave tile (800 -> 400 workgroups), ragged last tile
    (16, 80, 79, 64, 128, 3, 1, 1, "silu", "after"),
    (16, 80, 80, 128, 250, 1, 1, 0, "none", None),
    # 1x1, 128 / 256 input channels, at least 65536 pixels: the streaming kernel (bf16); ragged pixel and channel counts
    (12, 80, 80, 128, 128, 1, 1, 0, "silu", None),
    (11, 79, 77, 256, 200, 1, 1, 0, "leaky", None),
]


@pytest.mark.parametrize("compute,tol", [("fp32", 1e-3), ("bf16", 2e-2)])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_against_torch(compute, tol, case):
    N, H, W, Cin, Cout, k, s, p, act, resk = case
    g = _gen(7, f"conv{case}")
    conv = nn.Conv2d(Cin, Cout, k, s, p, bias=True)
    bn = nn.BatchNorm2d(Cout) if act != "none" else None
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (Cin * k * k)) ** 0.5)
        conv.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
        if bn is not None:
            bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
            bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
            bn.eval()
    x = torch.randn(N, Cin, H, W, generator=g)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    r = torch.randn(N, Cout, Ho, Wo, generator=g) if resk else None
    actf = {"silu": F.silu, "leaky": lambda t: F.leaky_relu(t, 0.1), "none": lambda t: t}[act]
    with torch.no_grad():
        y = conv(x)
        if bn is not None:
            y = bn(y)
        ref = actf(y + r) if resk == "before" else (actf(y) + (r if resk else 0))
    code = {"silu": _ffi.ACT_SILU, "leaky": _ffi.ACT_LEAKY, "none": _ffi.ACT_NONE}[act]

    def build(gr, ins):
        res = ins[1] if resk else None
        mode = {None: _ffi.RES_NONE, "after": _ffi.RES_AFTER_ACT, "before": _ffi.RES_BEFORE_ACT}[resk]
        out = gr.conv(ins[0], conv, bn, act=code, slope=0.1, res=res, res_mode=mode,
                      out_dtype=_ffi.F32 if act == "none" else None)
        gr.to_nchw(out, "y")

    out = _run_graph(build, *([x, r] if resk else [x]), compute=compute)
    assert out["y"].shape == ref.shape
    assert rel_err(out["y"].float().cpu(), ref) < tol
    if N == 16 and k == 3:
        assert engine.conv_kernel_label(Cin, Cout, k, k, p, _ffi.BF16, _ffi.BF16, M=N * Ho * Wo, stride=s,
                                        res=resk is not None).startswith("conv_dma<128x256")
    if N in (11, 12):
        assert engine.conv_kernel_label(Cin, Cout, k, k, p, _ffi.BF16, _ffi.BF16, M=N * Ho * Wo, stride=s).startswith("conv_pw<")


@pytest.mark.parametrize("geom", [(30, 48, 48, 128, 128, "silu"), (29, 49, 47, 256, 192, "leaky"), (31, 47, 49, 128, 256, "silu")])
def test_conv_pw_streaming_equals_tiled(geom):
    """A 1x1 convolution over at least 65536 pixels goes to the streaming kernel (weights resident in LDS, pixel
    fragments straight from the map into registers), the same layer over one image to the tiled kernel: same MFMA, same
    K order, same epilogue -- every image of the batch equals its own single-image run bit for bit."""
    N, H, W, cin, cout, act = geom
    gen = _gen(41, f"pw{geom}")
    x = torch.randn(N, cin, H, W, generator=gen).bfloat16().float()
    conv = nn.Conv2d(cin, cout, 1).requires_grad_(False)
    conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / cin) ** 0.5)
    conv.bias.copy_(torch.randn(cout, generator=gen) * 0.1)
    code = _ffi.ACT_SILU if act == "silu" else _ffi.ACT_LEAKY
    assert engine.conv_kernel_label(cin, cout, 1, 1, 0, _ffi.BF16, _ffi.BF16, M=N * H * W).startswith("conv_pw<")
    assert engine.conv_kernel_label(cin, cout, 1, 1, 0, _ffi.BF16, _ffi.BF16, M=H * W).startswith("conv_dma<")

    def build(gr, ins):
        gr.to_nchw(gr.conv(ins[0], conv, act=code, slope=0.1), "y")

    full = _run_graph(build, x, compute="bf16")["y"].float().cpu()
    ref = F.silu(conv(x)) if act == "silu" else F.leaky_relu(conv(x), 0.1)
    assert rel_err(full.numpy(), ref.numpy()) < 2e-2
    for n in (0, N // 2, N - 1):
        one = _run_graph(build, x[n:n + 1], compute="bf16")["y"].float().cpu()
        assert torch.equal(one[0], full[n]), f"image {n}"


@pytest.mark.parametrize("geom", [(32, 40, 40, 128, 512, 3, 1, True), (31, 41, 39, 64, 400, 3, 1, False), (30, 44, 44, 512, 256, 1, 1, False),
                                  (32, 40, 40, 96, 256, 3, 1, False), (32, 80, 80, 128, 256, 3, 2, False), (25, 38, 38, 256, 512, 3, 1, True),
                                  (64, 38, 38, 512, 256, 1, 1, True), (61, 19, 19, 192, 700, 3, 1, False), (64, 19, 19, 256, 512, 3, 1, True)])
def test_conv_256_tile_equals_small_launch(geom):
    """Wide layers with a long K and about a tile per CU take the 256 x 256 tile -- eight waves with the software-pipelined
    K loop and buffer-addressed operands (padding taps and rows beyond the last pixel are out-of-range offsets that read
    zeros) where Cin is a multiple of 64, sixteen waves otherwise (Cin = 96 here); one image of the same layer takes a
    128-wide tile.  Same MFMA, same K order, same epilogue: every image equals its own single-image run bit for bit
    (ragged channel tile: 400 = 256 + 144; ragged pixel tiles; residual; stride 2; K = 2304).  The last two shapes take the
    256 x 192 form of the pipelined tile (its last round of workgroups fills better: 482 tiles instead of 361; 345 instead
    of 261 with a ragged channel tile 700 = 512 + 188; 242 instead of 182, which alone would be too few for a wide tile)."""
    N, H, W, cin, cout, k, s, res = geom
    gen = _gen(43, f"t256{geom}")
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(N, cin, H, W, generator=gen).bfloat16().float()
    r = torch.randn(N, cout, Ho, Wo, generator=gen).bfloat16().float() if res else None
    conv = nn.Conv2d(cin, cout, k, s, p).requires_grad_(False)
    conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (cin * k * k)) ** 0.5)
    conv.bias.copy_(torch.randn(cout, generator=gen) * 0.1)
    lab = engine.conv_kernel_label(cin, cout, k, k, p, _ffi.BF16, _ffi.BF16, M=N * Ho * Wo, stride=s, res=res)
    assert lab.startswith("conv_dma<256x192" if geom[:3] in ((64, 38, 38), (61, 19, 19), (64, 19, 19)) else "conv_dma<256x")
    assert not engine.conv_kernel_label(cin, cout, k, k, p, _ffi.BF16, _ffi.BF16, M=Ho * Wo, stride=s,
                                        res=res).startswith("conv_dma<256x256")

    def build(gr, ins):
        kw = dict(res=ins[1], res_mode=_ffi.RES_AFTER_ACT) if res else {}
        gr.to_nchw(gr.conv(ins[0], conv, act=_ffi.ACT_SILU, **kw), "y")

    full = _run_graph(build, *([x, r] if res else [x]), compute="bf16")["y"].float().cpu()
    ref = F.silu(conv(x)) + (r if res else 0)
    assert rel_err(full.numpy(), ref.numpy()) < 2e-2
    for n in (0, N // 2, N - 1):
        one = _run_graph(build, *([x[n:n + 1], r[n:n + 1]] if res else [x[n:n + 1]]), compute="bf16")["y"].float().cpu()
        assert torch.equal(one[0], full[n]), f"image {n}"


@pytest.mark.parametrize("geom", [(64, 40, 40, 128, 128, "silu", "after"), (40, 37, 53, 128, 200, "silu", None), (33, 76, 76, 64, 96, "leaky", "after"),
                                  (3, 152, 152, 64, 128, "silu", None), (2, 130, 224, 64, 128, "leaky", None), (36, 23, 40, 128, 256, "silu", "before")])
def test_conv_patch_equals_tiled_launch(geom):
    """3x3 / stride 1 / pad 1 layers over 64 or 128 channels with a tile of whole rows per CU take the kernel that holds the
    rows' input patch in LDS (persistent workgroups: the next tile's patch and the residual pieces arrive under the K loop);
    one image of the same layer has too few tiles and takes the tiled kernel.  Same MFMA, same K order, same epilogue:
    every image equals its own single-image run bit for bit (ragged last row tile: 37 = 9 x 4 + 1 and 23 = 4 x 5 + 3; ragged
    channel tile: 200 = 128 + 72, 96; the widest map the kernel takes; residual after and before the activation)."""
    N, H, W, cin, cout, act, res = geom
    gen = _gen(47, f"patch{geom}")
    x = torch.randn(N, cin, H, W, generator=gen).bfloat16().float()
    r = torch.randn(N, cout, H, W, generator=gen).bfloat16().float() if res else None
    conv = nn.Conv2d(cin, cout, 3, 1, 1).requires_grad_(False)
    conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (cin * 9)) ** 0.5)
    conv.bias.copy_(torch.randn(cout, generator=gen) * 0.1)
    assert engine.conv_kernel_label(cin, cout, 3, 3, 1, _ffi.BF16, _ffi.BF16, M=N * H * W, res=res is not None,
                                    geom=(N, H, W)) == f"conv_patch<c{cin}>"
    assert engine.conv_kernel_label(cin, cout, 3, 3, 1, _ffi.BF16, _ffi.BF16, M=H * W, res=res is not None,
                                    geom=(1, H, W)).startswith("conv_dma<")

    def build(gr, ins):
        kw = dict(res=ins[1], res_mode=_ffi.RES_AFTER_ACT if res == "after" else _ffi.RES_BEFORE_ACT) if res else {}
        gr.to_nchw(gr.conv(ins[0], conv, act=_ffi.ACT_SILU if act == "silu" else _ffi.ACT_LEAKY, slope=0.1, **kw), "y")

    full = _run_graph(build, *([x, r] if res else [x]), compute="bf16")["y"].float().cpu()
    f = F.silu if act == "silu" else (lambda t: F.leaky_relu(t, 0.1))
    ref = f(conv(x) + r) if res == "before" else f(conv(x)) + (r if res else 0)
    assert rel_err(full.numpy(), ref.numpy()) < 2e-2
    for n in (0, N // 2, N - 1):
        one = _run_graph(build, *([x[n:n + 1], r[n:n + 1]] if res else [x[n:n + 1]]), compute="bf16")["y"].float().cpu()
        assert torch.equal(one[0], full[n]), f"image {n}"


def test_conv_patch_on_channel_slices():
    """The patch kernel on operands that are channel slices of wider buffers -- input = channels 64 .. 191 of a 256-channel
    map, residual = that same slice, output = the first half of a concat buffer whose other half another layer writes:
    every image equals its single-image run (tiled kernel), and torch fp32 within the bf16 bar."""
    N, H, W = 56, 23, 40
    assert engine.conv_kernel_label(128, 128, 3, 3, 1, _ffi.BF16, _ffi.BF16, M=N * H * W, res=True, geom=(N, H, W)) == "conv_patch<c128>"
    gen = _gen(48, "patchslices")
    x = torch.randn(N, 256, H, W, generator=gen).bfloat16().float()
    conv = nn.Conv2d(128, 128, 3, 1, 1).requires_grad_(False)
    conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (128 * 9)) ** 0.5)
    conv.bias.copy_(torch.randn(128, generator=gen) * 0.1)
    other = nn.Conv2d(256, 64, 1).requires_grad_(False)
    other.weight.copy_(torch.randn(other.weight.shape, generator=gen) * (2.0 / 256) ** 0.5)
    other.bias.zero_()

    def build(gr, ins):
        xin = gr.slice(ins[0], 64, 128)
        y = gr.conv(xin, conv, act=_ffi.ACT_SILU, res=xin, res_mode=_ffi.RES_AFTER_ACT)
        z = gr.conv(ins[0], other, act=_ffi.ACT_SILU)
        gr.to_nchw(gr.cat([y, z]), "y")

    full = _run_graph(build, x, compute="bf16")["y"].float().cpu()
    ref = torch.cat([F.silu(conv(x[:, 64:192])) + x[:, 64:192], F.silu(other(x))], 1)
    assert rel_err(full.numpy(), ref.numpy()) < 2e-2
    for n in (0, N - 1):
        one = _run_graph(build, x[n:n + 1], compute="bf16")["y"].float().cpu()
        assert torch.equal(one[0], full[n]), f"image {n}"


GROUPED_CASES = [
    # N, H, W, Cin, Cout, k, s, p, groups, dilation, act, residual
    (2, 20, 20, 64, 64, 3, 1, 1, 64, 1, "silu", None),       # depthwise (DWConv, yolov5_official/modules.py:19-21)
    (2, 21, 19, 32, 64, 3, 2, 1, 32, 1, "silu", None),       # depthwise with a channel multiplier, stride 2, odd sizes
    (1, 16, 16, 32, 64, 3, 1, 1, 2, 1, "silu", "after"),     # Bottleneck(g=2).cv2 with its shortcut (:208-218)
    (2, 12, 12, 48, 24, 1, 1, 0, 3, 1, "leaky", None),       # grouped 1x1
    (1, 18, 18, 16, 32, 3, 1, 2, 1, 2, "leaky", None),       # dilated (utils/moduleUtils.py:9-13 `d`)
    (1, 17, 15, 32, 32, 3, 2, 3, 4, 3, "none", "before"),    # grouped and dilated, stride 2
]


@pytest.mark.parametrize("compute,tol", [("fp32", 1e-3), ("bf16", 2e-2)])
@pytest.mark.parametrize("case", GROUPED_CASES)
def test_conv_grouped_dilated_against_torch(compute, tol, case):
    """nn.Conv2d(groups=, dilation=) through the direct kernel, against torch fp32."""
    N, H, W, Cin, Cout, k, s, p, groups, dil, act, resk = case
    g = _gen(9, f"gconv{case}")
    conv = nn.Conv2d(Cin, Cout, k, s, p, dilation=dil, groups=groups, bias=True)
    bn = nn.BatchNorm2d(Cout) if act != "none" else None
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (Cin // groups * k * k)) ** 0.5)
        conv.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
        if bn is not None:
            bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
            bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
            bn.eval()
    x = torch.randn(N, Cin, H, W, generator=g)
    with torch.no_grad():
        y = conv(x)
        if bn is not None:
            y = bn(y)
    r = torch.randn(y.shape, generator=g) if resk else None
    actf = {"silu": F.silu, "leaky": lambda t: F.leaky_relu(t, 0.1), "none": lambda t: t}[act]
    ref = actf(y + r) if resk == "before" else (actf(y) + (r if resk else 0))
    code = {"silu": _ffi.ACT_SILU, "leaky": _ffi.ACT_LEAKY, "none": _ffi.ACT_NONE}[act]

    def build(gr, ins):
        mode = {None: _ffi.RES_NONE, "after": _ffi.RES_AFTER_ACT, "before": _ffi.RES_BEFORE_ACT}[resk]
        out = gr.conv(ins[0], conv, bn, act=code, slope=0.1, res=ins[1] if resk else None, res_mode=mode,
                      out_dtype=_ffi.F32 if act == "none" else None)
        gr.to_nchw(out, "y")

    out = _run_graph(build, *([x, r] if resk else [x]), compute=compute)
    assert out["y"].shape == ref.shape
    assert rel_err(out["y"].float().cpu(), ref) < tol


def test_grouped_modules_lower():
    """DWConv and Bottleneck(g=2) of the YOLOv5 mirrors run on the GPU path and match their own torch forward."""
    from yolo_amd.yolov5_official.modules import Bottleneck, DWConv
    from yolo_amd.synthetic import init_deterministic
    for name, mod, cin in (("dw", DWConv(32, 64, 3, 2), 32), ("bott", Bottleneck(64, 64, True, g=2), 64)):
        init_deterministic(mod, seed=3)
        mod.eval()
        x = torch.randn(2, cin, 24, 20, generator=_gen(4, name))
        with torch.no_grad():
            ref = mod(x)
        out = _run_graph(lambda gr, ins: gr.to_nchw(mod.lower(gr, ins[0]), "y"), x, compute="fp32")
        assert rel_err(out["y"].float().cpu(), ref) < 1e-3, name


def test_c3spp_lowers_and_numpy_nms_is_exported(golden):
    """C3SPP (yolov5_official/modules.py:262-268: a C3 whose middle is one SPP block) runs on the GPU path and matches its
    own torch forward; models.common offers Contract / Expand / Concat; yolov5_official.utils.nms (:551-578, the numpy
    greedy NMS) gives the reference's recorded kept list."""
    from yolo_amd.models import common as C
    from yolo_amd.synthetic import init_deterministic
    from yolo_amd.yolov5_official import utils as U
    from yolo_amd.yolov5_official.modules import C3SPP, Concat, Contract, Expand
    for compute, tol in (("fp32", 1e-3), ("bf16", 2e-2)):
        mod = C3SPP(64, 64)
        init_deterministic(mod, seed=3)
        mod.eval()
        x = torch.randn(2, 64, 20, 24, generator=_gen(4, "c3spp")).bfloat16().float()
        with torch.no_grad():
            ref = mod(x)
        out = _run_graph(lambda gr, ins: gr.to_nchw(mod.lower(gr, ins[0]), "y"), x, compute=compute)
        assert rel_err(out["y"].float().cpu(), ref) < tol, compute
    assert (C.Contract, C.Expand, C.Concat) == (Contract, Expand, Concat) and issubclass(C.C3SPP, C.C3)
    t = torch.arange(2 * 4 * 6 * 6, dtype=torch.float32).view(2, 4, 6, 6)
    assert torch.equal(Expand(2)(Contract(2)(t)), t) and Contract(2)(t).shape == (2, 16, 3, 3)
    g = golden("g3_postprocess_0")
    assert U.nms(g["bb"], g["scores"][:, 0], float(g["thr"])) == [int(i) for i in g["nms_keep_class0_scores"]]


# ---- K split over workgroups (opt-in) ---------------------------------------------------------------------------------
@pytest.mark.parametrize("geom", [(1, 256, 512, 3, 1, 13, 13, False), (1, 1024, 256, 1, 1, 13, 13, False), (2, 512, 256, 3, 2, 26, 22, False),
                                  (1, 256, 256, 3, 1, 20, 20, True), (3, 1024, 512, 1, 1, 7, 9, True), (1, 128, 192, 5, 1, 11, 11, False)])
def test_conv_split_k(geom):
    """Few tiles and a long K: with engine.set_split_k the taps (k x k) or channel ranges (1x1) go to different workgroups
    and a second kernel adds the partial sums before bias / activation / residual.  Against the unsplit launch (same
    values up to the order of the fp32 additions: bf16 outputs differ in few places, by one ulp) and against torch."""
    N, cin, cout, k, s, H, W, res = geom
    gen = _gen(31, f"splitk{geom}")
    x = torch.randn(N, cin, H, W, generator=gen).bfloat16().float()
    conv = nn.Conv2d(cin, cout, k, s, k // 2).requires_grad_(False)
    conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * (2.0 / (cin * k * k)) ** 0.5)
    Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
    r = torch.randn(N, cout, Ho, Wo, generator=gen).bfloat16().float() if res else None

    def build(gr, ins):
        kw = dict(res=ins[1], res_mode=_ffi.RES_AFTER_ACT) if res else {}
        gr.to_nchw(gr.conv(ins[0], conv, act=_ffi.ACT_LEAKY, slope=0.1, **kw), "y")

    outs = {}
    for split in (True, False):
        engine.set_split_k(split)
        try:
            g = engine.Graph(DEV, "bf16")
            build(g, [g.input_nchw(*t.shape) for t in ([x, r] if res else [x])])
            plan = g.finalize()
            conv_op = [op for op in plan.ops if op.kind == _ffi.OP_CONV][0]
            assert bool(conv_op.conv.splitk_ws) == split, "the launch was expected to qualify for a K split"
            outs[split] = plan.run(*[t.to(DEV) for t in ([x, r] if res else [x])])["y"].float().cpu()
        finally:
            engine.set_split_k(False)
    ref = F.leaky_relu(conv(x), 0.1) + (r if res else 0)
    assert rel_err(outs[True].numpy(), ref.numpy()) < 2e-2
    assert rel_err(outs[True].numpy(), outs[False].numpy()) < 1e-2
    assert (outs[True] != outs[False]).float().mean() < 0.05


def test_conv_split_k_two_destinations():
    """A split launch whose output channels go to two buffers (the sibling pair of a C3 written separately): the
    reduction kernel sends every 4-channel group to the right one."""
    gen = _gen(32, "splitk_y2")
    x = torch.randn(1, 1024, 13, 13, generator=gen).bfloat16().float()
    c1, c2 = nn.Conv2d(1024, 64, 1).requires_grad_(False), nn.Conv2d(1024, 96, 1).requires_grad_(False)
    for c in (c1, c2):
        c.weight.copy_(torch.randn(c.weight.shape, generator=gen) * (2.0 / 1024) ** 0.5)

    def build(gr, ins):
        _, (y, y2) = gr.conv_siblings(ins[0], [(c1, None), (c2, None)], act=_ffi.ACT_SILU, separate=True)
        gr.to_nchw(y, "y")
        gr.to_nchw(y2, "y2")

    outs = {}
    for split in (True, False):
        engine.set_split_k(split)
        try:
            g = engine.Graph(DEV, "bf16")
            build(g, [g.input_nchw(*x.shape)])
            plan = g.finalize()
            assert bool([op for op in plan.ops if op.kind == _ffi.OP_CONV][0].conv.splitk_ws) == split
            outs[split] = {k: v.float().cpu() for k, v in plan.run(x.to(DEV)).items()}
        finally:
            engine.set_split_k(False)
    for k, c in (("y", c1), ("y2", c2)):
        assert rel_err(outs[True][k].numpy(), F.silu(c(x)).numpy()) < 2e-2, k
        assert rel_err(outs[True][k].numpy(), outs[False][k].numpy()) < 1e-2, k


def test_conv_argument_validation():
    lib = _ffi.lib()
    d = _ffi.ConvDesc()
    rc = lib.ydet_conv2d_nhwc(d, None)
    assert rc == -1 and b"null" in lib.ydet_last_error()
    with pytest.raises(RuntimeError):
        _ffi.check(rc, "conv")


# ---- Focus + first convolution fused with the NCHW edge -----------------------------------------------------------
@pytest.mark.parametrize("shape,cout", [((2, 3, 64, 96), 32), ((1, 3, 50, 70), 32), ((2, 3, 32, 64), 48), ((1, 3, 16, 16), 64)])
@pytest.mark.parametrize("src", [torch.float32, torch.bfloat16])
def test_focus_stem_fused(shape, cout, src):
    """Image sizes that are not multiples of the 8 x 32 patch, 32 / 48 / 64 output channels, fp32 and bf16 images."""
    from yolo_amd.yolov5_official.modules import Focus
    from yolo_amd.synthetic import init_deterministic
    m = Focus(3, cout, 3)
    init_deterministic(m, seed=3)
    m.eval()
    x = torch.randn(*shape, generator=_gen(8, f"stem{shape}")).to(src)
    with torch.no_grad():
        ref = m(x.float())
    m = m.to(DEV)
    outs = {}
    for compute in ("bf16", "fp32"):
        g = engine.Graph(DEV, compute)
        y = g.focus_stem(*shape, m.conv.conv, m.conv.bn, act=_ffi.ACT_SILU)
        g.to_nchw(y, "y")
        assert [k for k, _ in g.ops][0] == ("focus_conv" if compute == "bf16" else "nchw2nhwc")
        outs[compute] = g.finalize().run(x.to(DEV))["y"].float().cpu()
    assert outs["bf16"].shape == ref.shape
    assert rel_err(outs["fp32"], ref) < (1e-3 if src == torch.float32 else 1e-2)
    assert rel_err(outs["bf16"], ref) < 2e-2


# ---- Focus + the stride-2 convolution behind it (+ the C3's cv1 | cv2) as one kernel -------------------------------
@pytest.mark.parametrize("shape", [(2, 3, 64, 96), (1, 3, 50, 70), (1, 3, 36, 132), (3, 3, 16, 16)])
@pytest.mark.parametrize("tail", ["conv", "c3", "conv_leaky"])
@pytest.mark.parametrize("src", [torch.float32, torch.bfloat16, torch.uint8])
def test_focus_pair_fused(shape, tail, src):
    """Focus(3, 32, 3) -> Conv(32, 64, 3, 2) [-> C3(64, 64)]: image sizes that are not multiples of the patch of 8 x 16
    output pixels (odd half-size maps included), with and without the 1x1 follower / second destination, fp32 / bf16 /
    raw uint8 images.  The fused launch against the layers run one by one (same bf16 intermediate values and accumulation
    order: equal up to single bf16 ulps where an epilogue contracts differently) and against torch fp32."""
    from yolo_amd.yolov5_official.modules import C3, Conv, Focus
    from yolo_amd.synthetic import init_deterministic
    mods = [Focus(3, 32, 3), Conv(32, 64, 3, 2)] + ([C3(64, 64, 1)] if tail == "c3" else [])
    net = nn.Sequential(*mods)
    if tail == "conv_leaky":
        mods[0].conv.act = mods[1].act = nn.LeakyReLU(0.1)
    init_deterministic(net, seed=11)
    net.eval()
    gen = _gen(21, f"pair{shape}")
    if src == torch.uint8:
        x = torch.randint(0, 256, shape, generator=gen, dtype=torch.uint8)
        xf = x.float() / 255.0
    else:
        x = torch.randn(*shape, generator=gen).to(src)
        xf = x.float()
    with torch.no_grad():
        ref = net(xf)
    net = net.to(DEV)
    outs = {}
    for fused in (True, False):
        engine.set_focus_pair_fusion(fused)
        try:
            g = engine.Graph(DEV, "bf16")
            act, slope = (_ffi.ACT_LEAKY, 0.1) if tail == "conv_leaky" else (_ffi.ACT_SILU, 0.0)
            t = g.focus_stem(*shape, mods[0].conv.conv, mods[0].conv.bn, act=act, slope=slope)
            for m in mods[1:]:
                t = m.lower(g, t)
            g.to_nchw(t, "y")
            plan = g.finalize()
            kinds = [k for k, _ in g.ops]
            assert kinds[0] == ("focus_pair" if fused else "focus_conv"), kinds
            outs[fused] = plan.run(x.to(DEV))["y"].float().cpu()
        finally:
            engine.set_focus_pair_fusion(True)
    assert outs[True].shape == ref.shape
    assert rel_err(outs[True].numpy(), outs[False].numpy()) < 1e-2
    assert (outs[True] != outs[False]).float().mean() < 0.02
    assert rel_err(outs[True].numpy(), ref.numpy()) < 2e-2


@pytest.mark.parametrize("shape", [(2, 3, 64, 96), (1, 3, 51, 70), (1, 3, 37, 131), (3, 3, 16, 16), (1, 3, 19, 35)])
@pytest.mark.parametrize("tail", ["conv", "resblock", "silu", "follower"])
@pytest.mark.parametrize("src", [torch.float32, torch.bfloat16, torch.uint8])
def test_rgb_pair_fused(shape, tail, src):
    """Conv(3, 32, 3, p=1) -> Conv(32, 64, 3, p=1, s=2) [-> resblock(64) | -> Conv(64, 48, 1) riding as a follower], the opening of the Darknet-53 / CSPDarknet-53
    trunks (models/basemodel/d19.py:83-123, :215-265): odd and even image sizes, sizes that are not multiples of the
    8 x 16 output patch, fp32 / bf16 / uint8 (converted on the host) images.  The fused launch against the two layers
    run one by one (same bf16 intermediate values and accumulation order) and against torch fp32."""
    from yolo_amd.utils.moduleUtils import Conv, resblock
    from yolo_amd.synthetic import init_deterministic
    mods = [Conv(3, 32, 3, p=1), Conv(32, 64, 3, p=1, s=2)] + ([resblock(64, nblocks=1)] if tail == "resblock" else [])
    if tail == "silu":
        mods[0].conv[2] = nn.SiLU()
        mods[1].conv[2] = nn.SiLU()
    if tail == "follower":
        mods.append(Conv(64, 48, 1))
    net = nn.Sequential(*mods)
    init_deterministic(net, seed=13)
    net.eval()
    gen = _gen(23, f"rgbpair{shape}")
    if src == torch.uint8:
        x = torch.randint(0, 256, shape, generator=gen, dtype=torch.uint8)
        xf = x.float() / 255.0
    else:
        x = torch.randn(*shape, generator=gen).to(src)
        xf = x.float()
    with torch.no_grad():
        ref = net(xf)
    net = net.to(DEV)
    outs = {}
    for fused in (True, False):
        engine.set_focus_pair_fusion(fused)
        try:
            g = engine.Graph(DEV, "bf16")
            t = g.input_nchw(*shape)
            for m in mods:
                t = m.lower(g, t)
            g.to_nchw(t, "y")
            plan = g.finalize()
            kinds = [k for k, _ in g.ops]
            assert kinds[0] == ("focus_pair" if fused else "rgb_conv"), kinds
            assert (plan.op_lists[0][0].kind == _ffi.OP_RGB_PAIR) == fused
            outs[fused] = plan.run(x.to(DEV))["y"].float().cpu()
        finally:
            engine.set_focus_pair_fusion(True)
    assert outs[True].shape == ref.shape
    assert rel_err(outs[True].numpy(), outs[False].numpy()) < 1e-2
    assert (outs[True] != outs[False]).float().mean() < 0.02
    assert rel_err(outs[True].numpy(), ref.numpy()) < 2e-2


@pytest.mark.parametrize("shape", [(2, 3, 64, 96), (1, 3, 50, 70), (1, 3, 36, 130), (3, 3, 16, 16), (1, 3, 18, 34)])
@pytest.mark.parametrize("cout", [16, 32, 64])
@pytest.mark.parametrize("src", [torch.float32, torch.bfloat16])
def test_rgb_stem_pool_fused(shape, cout, src):
    """Conv(3, C, 3, p=1) -> MaxPool2d(2, 2) -> Conv(C, 2C, 3, p=1): the opening of the Darknet-19 / tiny trunks
    (models/basemodel/d19.py:29, :175-182).  The pool taken in the first layer's epilogue equals the two launches bit for
    bit (max commutes with the rounding to bf16); both against torch fp32."""
    from yolo_amd.utils.moduleUtils import Conv
    from yolo_amd.synthetic import init_deterministic
    net = nn.Sequential(nn.Sequential(Conv(3, cout, 3, p=1), nn.MaxPool2d(2, 2)), Conv(cout, 2 * cout, 3, p=1))
    init_deterministic(net, seed=17)
    net.eval()
    x = torch.randn(*shape, generator=_gen(29, f"rgbpool{shape}")).to(src)
    with torch.no_grad():
        ref = net(x.float())
    net = net.to(DEV)
    outs = {}
    for fused in (True, False):
        engine.set_rgb_pool_fusion(fused)
        try:
            g = engine.Graph(DEV, "bf16")
            t = engine.lower_any(g, net, g.input_nchw(*shape))
            g.to_nchw(t, "y")
            plan = g.finalize()
            kinds = [k for k, _ in g.ops]
            assert kinds[0] == "rgb_conv" and ("maxpool2" in kinds) == (not fused), kinds
            assert (plan.op_lists[0][0].conv.Ho == shape[2] // 2) == fused
            outs[fused] = plan.run(x.to(DEV))["y"].float().cpu()
        finally:
            engine.set_rgb_pool_fusion(True)
    assert outs[True].shape == ref.shape
    assert torch.equal(outs[True], outs[False])
    assert rel_err(outs[True].numpy(), ref.numpy()) < 2e-2


# ---- 32-channel bottleneck as one kernel ---------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [(2, 32, 16, 64), (1, 32, 13, 37), (3, 32, 40, 40), (1, 32, 5, 3),
                                   (2, 64, 16, 32), (1, 64, 13, 37), (3, 64, 40, 40), (1, 64, 5, 3)])
@pytest.mark.parametrize("add", [True, False])
@pytest.mark.parametrize("act", ["silu", "leaky"])
def test_bottleneck_c32_fused(shape, add, act):
    """Tile-aligned and ragged maps (8 x 32 output tiles), with and without the shortcut: the fused kernel against
    torch fp32 (bf16 bar) and against the same two stages run as separate convolutions (same bf16 intermediate and
    accumulation order; the epilogues may contract multiply-adds differently, so the bar is one bf16 ulp)."""
    from yolo_amd.yolov5_official.modules import Bottleneck
    from yolo_amd.synthetic import init_deterministic
    m = Bottleneck(shape[1], shape[1], shortcut=add, e=1.0)
    if act == "leaky":
        m.cv1.act = m.cv2.act = nn.LeakyReLU(0.1)
    init_deterministic(m, seed=5)
    m.eval()
    x = torch.randn(*shape, generator=_gen(9, f"bneck{shape}")).bfloat16().float()
    with torch.no_grad():
        ref = m(x)
    m = m.to(DEV)
    code, slope = (_ffi.ACT_SILU, 0.0) if act == "silu" else (_ffi.ACT_LEAKY, 0.1)

    def fused(gr, ins):
        y = m.lower(gr, ins[0])
        assert [k for k, _ in gr.ops][-1] == "bottleneck"
        gr.to_nchw(y, "y")

    def split(gr, ins):
        t = gr.conv(ins[0], m.cv1.conv, m.cv1.bn, act=code, slope=slope)
        y = gr.conv(t, m.cv2.conv, m.cv2.bn, act=code, slope=slope, res=ins[0] if add else None, res_mode=_ffi.RES_AFTER_ACT)
        gr.to_nchw(y, "y")

    a = _run_graph(fused, x, compute="bf16")["y"].float().cpu()
    b = _run_graph(split, x, compute="bf16")["y"].float().cpu()
    assert a.shape == ref.shape
    assert rel_err(a, ref) < 2e-2
    assert rel_err(a, b) < 2.0 ** -8
    assert (a != b).float().mean() < 0.02


@pytest.mark.parametrize("shape", [(2, 32, 16, 64), (1, 32, 13, 37), (2, 64, 24, 40), (1, 64, 5, 3)])
def test_csp_resblock_fused(shape):
    """act(x + conv3x3_noact(conv1x1(x))) -- the residual block of the CSP stages (utils/moduleUtils.py:79-91), shortcut
    before the activation -- as the fused kernel, against torch fp32 and against the two stages run as separate convolutions
    (same bf16 intermediate and accumulation order: one bf16 ulp, a few percent of the values)."""
    from yolo_amd.utils.moduleUtils import CSPResBlock
    from yolo_amd.synthetic import init_deterministic
    m = CSPResBlock(shape[1])
    init_deterministic(m, seed=6)
    m.eval()
    x = torch.randn(*shape, generator=_gen(10, f"cspres{shape}")).bfloat16().float()
    with torch.no_grad():
        ref = m(x.clone())
    m = m.to(DEV)

    def fused(gr, ins):
        y = m.lower(gr, ins[0])
        assert [k for k, _ in gr.ops][-1] == "bottleneck"
        gr.to_nchw(y, "y")

    def split(gr, ins):
        c1, c2 = m.conv1.conv, m.conv2.conv
        t = gr.conv(ins[0], c1[0], c1[1], act=_ffi.ACT_LEAKY, slope=0.1)
        y = gr.conv(t, c2[0], c2[1], act=_ffi.ACT_LEAKY, slope=0.1, res=ins[0], res_mode=_ffi.RES_BEFORE_ACT)
        gr.to_nchw(y, "y")

    a = _run_graph(fused, x, compute="bf16")["y"].float().cpu()
    b = _run_graph(split, x, compute="bf16")["y"].float().cpu()
    assert a.shape == ref.shape
    assert rel_err(a, ref) < 2e-2
    assert rel_err(a, b) < 2.0 ** -8
    assert (a != b).float().mean() < 0.02


@pytest.mark.parametrize("shape", [(2, 64, 16, 32), (1, 64, 13, 37), (3, 64, 40, 24), (1, 64, 5, 3)])
@pytest.mark.parametrize("act", ["leaky", "silu"])
def test_resblock_c64_fused(shape, act):
    """x + Conv3x3(Conv1x1(x)) with a 1x1 that halves the 64 channels (utils/moduleUtils.py:45-59, the first residual stage
    of Darknet-53) as one kernel, tile-aligned and ragged maps: against torch fp32 and against the two stages run as
    separate convolutions (same bf16 intermediate and accumulation order: one bf16 ulp, a few percent of the values)."""
    from yolo_amd.utils.moduleUtils import resblock
    from yolo_amd.synthetic import init_deterministic
    m = resblock(shape[1], nblocks=2)
    if act == "silu":
        for pair in m.module_list:
            pair[0].conv[2] = nn.SiLU()
            pair[1].conv[2] = nn.SiLU()
    init_deterministic(m, seed=8)
    m.eval()
    x = torch.randn(*shape, generator=_gen(13, f"resblock{shape}")).bfloat16().float()
    with torch.no_grad():
        ref = m(x.clone())
    m = m.to(DEV)
    code, slope = (_ffi.ACT_SILU, 0.0) if act == "silu" else (_ffi.ACT_LEAKY, 0.1)

    def fused(gr, ins):
        y = m.lower(gr, ins[0])
        assert [k for k, _ in gr.ops][-2:] == ["bottleneck", "bottleneck"]
        gr.to_nchw(y, "y")

    def split(gr, ins):
        y = ins[0]
        for pair in m.module_list:
            c1, c2 = pair[0].conv, pair[1].conv
            t = gr.conv(y, c1[0], c1[1], act=code, slope=slope)
            y = gr.conv(t, c2[0], c2[1], act=code, slope=slope, res=y, res_mode=_ffi.RES_AFTER_ACT)
        gr.to_nchw(y, "y")

    a = _run_graph(fused, x, compute="bf16")["y"].float().cpu()
    b = _run_graph(split, x, compute="bf16")["y"].float().cpu()
    assert a.shape == ref.shape
    assert rel_err(a, ref) < 2e-2
    assert rel_err(a, b) < 2.0 ** -7
    assert (a != b).float().mean() < 0.04


@pytest.mark.parametrize("c,n", [(64, 1), (128, 2), (256, 2)])
def test_csp_stage_lowering(c, n):
    """CSPStage (utils/moduleUtils.py:95-107): the 32- and 64-channel stages lower to a sibling convolution with two
    destinations, one fused kernel per residual block and conv3 over the joined halves; wider stages update the concat
    buffer in place.  Against torch fp32."""
    from yolo_amd.utils.moduleUtils import CSPStage
    from yolo_amd.synthetic import init_deterministic
    m = CSPStage(c, n=n)
    init_deterministic(m, seed=7)
    m.eval()
    x = torch.randn(2, c, 21, 27, generator=_gen(12, f"cspstage{c}")).bfloat16().float()
    with torch.no_grad():
        ref = m(x.clone())
    m = m.to(DEV)

    def build(gr, ins):
        y = m.lower(gr, ins[0])
        kinds = [k for k, _ in gr.ops]
        assert kinds.count("bottleneck") == (n if c <= 128 else 0)
        gr.to_nchw(y, "y")

    out = _run_graph(build, x, compute="bf16")["y"].float().cpu()
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 2e-2


def test_bottleneck_c32_rejects_in_place():
    lib = _ffi.lib()
    buf = torch.zeros(1, 8, 8, 32, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(128, 288, device=DEV, dtype=torch.bfloat16)
    b = torch.zeros(128, device=DEV)
    c1, c2 = _ffi.ConvDesc(), _ffi.ConvDesc()
    for d, k, p, kp in ((c1, 1, 0, 32), (c2, 3, 1, 288)):
        d.x = d.y = buf.data_ptr()
        d.w, d.bias = w.data_ptr(), b.data_ptr()
        d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo = 1, 8, 8, 32, 32, k, k, 1, p, 8, 8
        d.x_ld = d.y_ld = 32
        d.kp, d.cout_pad, d.act, d.in_dtype, d.out_dtype = kp, 128, _ffi.ACT_SILU, _ffi.BF16, _ffi.BF16
    assert lib.ydet_bottleneck_fused(c1, c2, None) == -1 and b"overlaps" in lib.ydet_last_error()


@pytest.mark.parametrize("c0,c1,cin,k", [(32, 32, 64, 1), (64, 24, 32, 3), (8, 128, 96, 1)])
def test_conv_second_destination(c0, c1, cin, k):
    """One convolution standing for two that read the same input: channels [0, c0) go to one buffer, the rest to a
    channel slice of another (here: the second half of a concat); bit-identical to the single-destination conv."""
    g = _gen(11, f"split{c0}{c1}")
    convs = [nn.Conv2d(cin, c, k, 1, k // 2, bias=True) for c in (c0, c1)]
    for c in convs:
        with torch.no_grad():
            c.weight.copy_(torch.randn(c.weight.shape, generator=g) * (2.0 / (cin * k * k)) ** 0.5)
            c.bias.copy_(torch.randn(c.bias.shape, generator=g) * 0.1)
    x = torch.randn(2, cin, 19, 23, generator=g)

    def joint(gr, ins):
        out, _ = gr.conv_siblings(ins[0], [(c, None) for c in convs], _ffi.ACT_SILU)
        gr.to_nchw(out, "y")

    def separate(gr, ins):
        _, (a, b) = gr.conv_siblings(ins[0], [(c, None) for c in convs], _ffi.ACT_SILU, separate=True)
        cat = gr.cat([gr.maxpool2s1(a), b])          # b is redirected into the concat buffer, a stays on its own
        gr.to_nchw(a, "a")
        gr.to_nchw(gr.slice(cat, c0, c1), "b")

    ref = _run_graph(joint, x, compute="bf16")["y"].float().cpu()
    out = _run_graph(separate, x, compute="bf16")
    assert torch.equal(out["a"].float().cpu(), ref[:, :c0]) and torch.equal(out["b"].float().cpu(), ref[:, c0:])


# ---- evaluation tail on the device -----------------------------------------------------------------------------------
def test_process_batch_golden_and_batched(golden):
    """ydet_process_batch against the reference's own output (fixture G7) and, batched with ragged counts, against the
    oracle: images without labels, without predictions, classes without labels, duplicate predictions that compete for
    one label, a degenerate (NaN IoU) pair."""
    from oracle import metrics as OM
    from yolo_amd.yolov5_official import utils as U
    g = golden("g7_v5_metrics")
    iouv = torch.from_numpy(g["iouv"]).to(DEV)
    got = U.process_batch(torch.from_numpy(g["preds"]).to(DEV), torch.from_numpy(g["labels"]).to(DEV), iouv)
    assert got.is_cuda and got.dtype == torch.bool
    np.testing.assert_array_equal(got.cpu().numpy(), g["correct"])

    gen = _gen(13, "pb")
    B, max_det = 6, 40
    det = torch.zeros(B, max_det, 6)
    counts = [40, 0, 17, 25, 40, 3]
    labels, starts = [], [0]
    for b in range(B):
        nl = [9, 4, 0, 6, 12, 2][b]
        xy = torch.rand(nl, 2, generator=gen) * 300
        lb = torch.cat([torch.randint(0, 3, (nl, 1), generator=gen).float(), xy, xy + torch.rand(nl, 2, generator=gen) * 80 + 5], 1)
        labels.append(lb)
        starts.append(starts[-1] + nl)
        n = counts[b]
        if n and nl:
            src = lb[torch.randint(0, nl, (n,), generator=gen)]
            det[b, :n, :4] = src[:, 1:] + torch.randn(n, 4, generator=gen) * 6
            det[b, :n, 5] = torch.where(torch.rand(n, generator=gen) < 0.8, src[:, 0], torch.full((n,), 4.0))
        elif n:
            det[b, :n, :4] = torch.rand(n, 4, generator=gen) * 300
        det[b, :n, 4] = torch.rand(n, generator=gen)
    det[0, 1] = det[0, 0]                       # an exact duplicate: only the first may claim
    labels[4][0, 1:] = torch.tensor([10., 10., 10., 10.])       # zero-area label and prediction of its class: 0 / 0
    det[4, 0, :4] = torch.tensor([50., 50., 50., 50.])
    det[4, 0, 5] = labels[4][0, 0]
    lab = torch.cat(labels, 0)
    got = ops.process_batch(det.to(DEV), torch.tensor(counts, dtype=torch.int32, device=DEV), lab.to(DEV),
                            torch.tensor(starts, dtype=torch.int32, device=DEV), iouv).cpu().numpy()
    for b in range(B):
        n = counts[b]
        ref = OM.process_batch(det[b, :n].numpy(), labels[b].numpy(), g["iouv"]) if n and len(labels[b]) else np.zeros((n, 10), bool)
        np.testing.assert_array_equal(got[b, :n], ref, err_msg=f"image {b}")
        assert not got[b, n:].any()
    assert got[0, 0].any() and not got[0, 1].any()


# ---- pre-processing: uint8 HWC images -> network input, one launch (ydet_preprocess_u8) ------------------------------
PRE_SHAPES = [(480, 640), (375, 500), (720, 1280), (1280, 1280), (37, 53), (640, 640), (333, 500), (1, 1), (12, 900)]


def _u8_images(shapes, seed=0):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, size=s + (3,), dtype=np.uint8) for s in shapes]


@pytest.mark.parametrize("new_shape", [(640, 640), (416, 416), (96, 160)])
def test_letterbox_batch_matches_oracle(new_shape):
    """Bit-exact against the CPU restatement: same fixed-point resampling, border, BGR->RGB, CHW, IEEE / 255."""
    from oracle import preproc as OP
    from yolo_amd.preprocess import letterbox_batch
    imgs = _u8_images(PRE_SHAPES, seed=new_shape[0])
    x, ratios, pads = letterbox_batch(imgs, new_shape=new_shape, auto=False, device=DEV)
    assert x.shape == (len(imgs), 3) + new_shape and x.dtype == torch.float32
    for i, im in enumerate(imgs):
        np.testing.assert_array_equal(x[i].cpu().numpy(), OP.v5_input(im, new_shape, auto=False), err_msg=str(im.shape))
        _, r, p = OP.letterbox(im, new_shape, auto=False)
        assert tuple(ratios[i]) == tuple(r) and tuple(pads[i]) == tuple(p)
    xb, _, _ = letterbox_batch(imgs[:3], new_shape=new_shape, auto=False, device=DEV, dtype=torch.bfloat16)
    assert torch.equal(xb, x[:3].to(torch.bfloat16))


def test_base_transform_batch_matches_oracle():
    from oracle import preproc as OP
    from yolo_amd.data.transformers import BaseTransform
    imgs = _u8_images(PRE_SHAPES, seed=7)
    for size in (416, 64):
        bt = BaseTransform(size)
        x = bt.batch(imgs, device=DEV)
        assert x.shape == (len(imgs), 3, size, size)
        for i, im in enumerate(imgs):
            np.testing.assert_array_equal(x[i].cpu().numpy(), OP.family_a_input(im, size, bt.mean, bt.std), err_msg=str(im.shape))


def test_preproc_golden(golden):
    """The reference's own letterbox / base_transform outputs (recorded with the restated resize standing in for cv2)."""
    from yolo_amd.data.transformers import BaseTransform
    from yolo_amd.yolov5_official.utils import letterbox
    import json
    g = golden("g10_preproc")
    for i in range(int(g["n_lb"])):
        kw = json.loads(str(g[f"lb{i}_kw"]))
        out, ratio, pad = letterbox(torch.from_numpy(g[f"lb{i}_img"]).to(DEV), tuple(int(v) for v in g[f"lb{i}_new_shape"]), **kw)
        np.testing.assert_array_equal(out.cpu().numpy(), g[f"lb{i}_out"])
        np.testing.assert_array_equal(np.array(ratio, np.float64), g[f"lb{i}_ratio"])
        np.testing.assert_array_equal(np.array(pad, np.float64), g[f"lb{i}_pad"])
    for i in range(int(g["n_bt"])):
        x, b, l = BaseTransform(int(g[f"bt{i}_size"]))(g[f"bt{i}_img"])
        assert b is None and l is None
        np.testing.assert_array_equal(x.cpu().numpy(), g[f"bt{i}_out"])


def test_preproc_strided_source_and_errors():
    from oracle import preproc as OP
    from yolo_amd.preprocess import letterbox_batch, preprocess_u8
    big = torch.from_numpy(_u8_images([(200, 300)], seed=3)[0]).to(DEV)
    crop = big[10:150, 20:260]                      # rows keep the parent's pitch
    x, _, _ = letterbox_batch([crop], new_shape=(128, 128), auto=False)
    np.testing.assert_array_equal(x[0].cpu().numpy(), OP.v5_input(crop.cpu().numpy(), (128, 128), auto=False))
    assert letterbox_batch([], new_shape=(64, 64), device=DEV)[0].shape == (0, 3, 64, 64)
    with pytest.raises(RuntimeError):
        preprocess_u8([big.cpu()], [(8, 8, 0, 0)], (8, 8), device="cpu")
    with pytest.raises(ValueError):
        preprocess_u8([big], [(80, 80, 0, 0)], (64, 64))
    with pytest.raises(TypeError):
        preprocess_u8([big.float()], [(8, 8, 0, 0)], (8, 8))
    with pytest.raises(ValueError):                 # auto=True pads to different canvases
        letterbox_batch(_u8_images([(480, 640), (640, 480)]), new_shape=(640, 640), auto=True, device=DEV)


# ---- VOC metric: the matching loop on the device (ydet_voc_match) ----------------------------------------------------
def _voc_random_case(seed, n_img=40, n_cls=5):
    rng = np.random.default_rng(seed)
    names = [f"img{i:03d}" for i in range(n_img)]
    labelmap = [f"c{j}" for j in range(n_cls)]
    recs, boxes = {}, [[np.empty((0, 5), np.float32) for _ in names] for _ in labelmap]
    for ii, n in enumerate(names):
        objs = []
        for _ in range(int(rng.integers(0, 7))):
            x, y = rng.integers(0, 300, 2)
            w, h = rng.integers(0, 120, 2)          # zero-area ground truth happens
            objs.append({"name": labelmap[int(rng.integers(0, n_cls - 1))], "difficult": int(rng.random() < 0.25),
                         "bbox": [int(x), int(y), int(x + w), int(y + h)]})
        recs[n] = objs
        for ci in range(n_cls - 1):                 # the last class never has detections (-1)
            k = int(rng.integers(0, 25))
            if k == 0:
                continue
            d = np.zeros((k, 5), np.float32)
            src = [o["bbox"] for o in objs if o["name"] == labelmap[ci]]
            for r in range(k):
                if src and rng.random() < 0.7:      # near-duplicates of a ground-truth box: contested claims
                    b = np.array(src[int(rng.integers(0, len(src)))], np.float32) - 1 + rng.normal(0, 4, 4).astype(np.float32)
                else:
                    p = rng.uniform(0, 300, 2)
                    b = np.concatenate([p, p + rng.uniform(0, 100, 2)]).astype(np.float32)
                if rng.random() < 0.05:
                    b[2:] = b[:2]                   # zero-area detection on a zero-area box: 0/0
                d[r, :4] = b
                d[r, 4] = np.float32(int(rng.integers(0, 40)) / 40.0) if rng.random() < 0.5 else rng.random()   # score ties
            boxes[ci][ii] = d
    return names, labelmap, recs, boxes


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_voc_matching_on_device_equals_host_loop(seed):
    from yolo_amd.data import vocapi_evaluator as V
    names, labelmap, recs, boxes = _voc_random_case(seed)
    for use07 in (True, False):
        dev_aps = V.voc_eval_all_device(boxes, names, recs, labelmap, DEV, 0.5, use07)
        host_aps = []
        for ci, cls in enumerate(labelmap):
            gts = {n: (np.array([o["bbox"] for o in recs[n] if o["name"] == cls], dtype=float).reshape(-1, 4),
                       np.array([o["difficult"] for o in recs[n] if o["name"] == cls], dtype=bool)) for n in names}
            dets = {n: V.round_like_results_file(boxes[ci][ii]) for ii, n in enumerate(names) if len(boxes[ci][ii])}
            with np.errstate(all="ignore"):
                host_aps.append(V.voc_eval_class(dets, gts, 0.5, use07)[2])
        assert dev_aps == host_aps
        assert dev_aps[-1] == -1.0


def test_voc_metric_on_device_golden(golden):
    """The reference's own voc_eval outputs (fixture from the imported reference), matching loop on the GPU."""
    from test_host import _voc_fixture
    g, ev, boxes = _voc_fixture(golden)
    ev.device = torch.device(DEV)
    assert abs(ev.evaluate_detections(boxes) - float(g["map"])) < 1e-12
    np.testing.assert_allclose(ev.aps, g["aps"], rtol=0, atol=1e-12)
    ev.evaluate_detections(boxes, use_07=False)
    np.testing.assert_allclose(ev.aps, g["aps_area"], rtol=0, atol=1e-12)


# ---- nearest upsample + concat folded into the 1x1 convolution that reads them ---------------------------------------
@pytest.mark.parametrize("compute", ["bf16", "fp32"])
@pytest.mark.parametrize("shape", [(2, 64, 128, 10, 12, 128), (3, 128, 64, 20, 20, 256), (1, 256, 256, 7, 9, 64),
                                   # at least 65536 pixels and 128 / 256 channels in all: the streaming 1x1 kernel (bf16)
                                   (12, 128, 128, 40, 40, 128), (11, 64, 64, 39, 41, 200)])
def test_conv_reads_upsampled_source(compute, shape):
    """cat(upsample(a), b) -> 1x1 conv with the upsample never materialised equals the three separate ops bit for bit
    (same operands, same accumulation order), and matches torch."""
    N, C1, C2, h, w, Cout = shape
    gen = _gen(11, f"up{shape}")
    a = torch.randn(N, C1, h, w, generator=gen)
    b = torch.randn(N, C2, 2 * h, 2 * w, generator=gen)
    conv = nn.Conv2d(C1 + C2, Cout, 1).requires_grad_(False)
    conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * 0.05)
    conv2 = nn.Conv2d(C1 + C2, 72, 1).requires_grad_(False)      # a second reader of the same concat

    def build(gr, ins):
        ta, tb = ins
        # the sources pass through a convolution-free identity copy so that they are NHWC arena tensors
        cat = gr.cat([gr.upsample2x(ta, 0), tb])
        gr.to_nchw(gr.conv(cat, conv, act=_ffi.ACT_LEAKY, slope=0.1), "y")
        gr.to_nchw(gr.conv(cat, conv2), "y2")

    outs = {}
    for fused in (True, False):
        engine.set_upsample_fusion(fused)
        try:
            g = engine.Graph(DEV, compute)
            build(g, [g.input_nchw(*a.shape), g.input_nchw(*b.shape)])
            plan = g.finalize()
            kinds = [k for k, _ in g.ops]
            assert kinds.count("upsample") == (0 if fused else 1)
            outs[fused] = plan.run(a.to(DEV), b.to(DEV))
        finally:
            engine.set_upsample_fusion(True)
    for k in ("y", "y2"):
        assert torch.equal(outs[True][k], outs[False][k]), k
    ref = F.leaky_relu(conv(torch.cat([F.interpolate(a, scale_factor=2, mode="nearest"), b], 1)), 0.1)
    assert rel_err(outs[True]["y"].float().cpu(), ref) < (1e-3 if compute == "fp32" else 2e-2)


def test_upsample_is_kept_when_another_op_reads_it():
    a = torch.randn(1, 32, 6, 6, generator=_gen(12, "keep"))
    b = torch.randn(1, 32, 12, 12, generator=_gen(12, "keep2"))
    conv = nn.Conv2d(64, 64, 1).requires_grad_(False)
    conv3 = nn.Conv2d(64, 64, 3, padding=1).requires_grad_(False)

    def build(gr, ins):
        cat = gr.cat([gr.upsample2x(ins[0], 0), ins[1]])
        gr.to_nchw(gr.conv(cat, conv), "y")
        gr.to_nchw(gr.conv(cat, conv3), "y3")      # a 3x3 reader needs the materialised map

    g = engine.Graph(DEV, "fp32")
    build(g, [g.input_nchw(*a.shape), g.input_nchw(*b.shape)])
    out = g.finalize().run(a.to(DEV), b.to(DEV))
    assert [k for k, _ in g.ops].count("upsample") == 1
    cat = torch.cat([F.interpolate(a, scale_factor=2, mode="nearest"), b], 1)
    assert rel_err(out["y"].cpu(), conv(cat)) < 1e-3 and rel_err(out["y3"].cpu(), conv3(cat)) < 1e-3


# ---- a 1x1 convolution run on the tile of the 64-channel convolution in front of it -------------------------------
@pytest.mark.parametrize("geom", [(3, 2, 1, 32, 64), (3, 1, 1, 64, 48), (1, 1, 0, 96, 64)])
def test_conv_with_pointwise_follower(geom):
    """conv (k, s, p, Cin -> 64) + SiLU followed by a 1x1 conv (64 -> Cout2) + SiLU as one launch: equal to the two
    launches up to the rounding of a different summation order inside a K step, and to torch within the bf16 bar."""
    k, s, p, cin, cout2 = geom
    gen = _gen(13, f"pw{geom}")
    x = torch.randn(2, cin, 22, 26, generator=gen).bfloat16().float()
    c1 = nn.Conv2d(cin, 64, k, s, p).requires_grad_(False)
    c2 = nn.Conv2d(64, cout2, 1).requires_grad_(False)
    c1.weight.copy_(torch.randn(c1.weight.shape, generator=gen) * (2.0 / (cin * k * k)) ** 0.5)
    c2.weight.copy_(torch.randn(c2.weight.shape, generator=gen) * 0.15)

    def build(gr, ins):
        t = gr.conv(ins[0], c1, act=_ffi.ACT_SILU)
        gr.to_nchw(gr.conv(t, c2, act=_ffi.ACT_SILU), "y")

    outs = {}
    for fused in (True, False):
        engine.set_follower_fusion(fused)
        try:
            g = engine.Graph(DEV, "bf16")
            build(g, [g.input_nchw(*x.shape)])
            plan = g.finalize()
            assert [k_ for k_, _ in g.ops].count("conv") == (1 if fused else 2)
            outs[fused] = plan.run(x.to(DEV))["y"].float().cpu()
        finally:
            engine.set_follower_fusion(True)
    assert rel_err(outs[True].numpy(), outs[False].numpy()) < 1e-2      # bf16 outputs: a few values differ by one ulp
    assert (outs[True] != outs[False]).float().mean() < 0.02
    ref = F.silu(c2(F.silu(c1(x))))
    assert rel_err(outs[True].numpy(), ref.numpy()) < 2e-2

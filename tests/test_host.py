"""CPU-side checks of the product: the C ABI surface, the planner, the host-side metric code against the
golden vectors, state_dict structure, and the 2-rank (gloo) data-parallel exchange.  No GPU needed."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import FA_SPECS, ROOT, build_fa, build_v5s

HEADER = os.path.join(ROOT, "include", "ydet.h")


# ---- C ABI ---------------------------------------------------------------------------------------------------
def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ydet_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from yolo_amd import _ffi
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert sorted(_ffi.SYMBOLS) == declared, "the ctypes table and include/ydet.h disagree"
    lib = _ffi.lib()   # dlopen + dlsym of every symbol; raises if one is missing
    for name in declared:
        assert getattr(lib, name) is not None
    header_version = int(re.search(r"#define YDET_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert lib.ydet_abi_version() == header_version == _ffi.ABI_VERSION
    assert lib.ydet_last_error() == b""


def test_ctypes_structs_match_header_layout():
    """sizeof() of the ctypes mirrors against a C compile of the header (catches field drift)."""
    from yolo_amd import _ffi
    src = '#include <stdio.h>\n#include "ydet.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(ydet_conv_desc),' \
          'sizeof(ydet_slice_desc), sizeof(ydet_v5_level), sizeof(ydet_fa_level), sizeof(ydet_op), sizeof(ydet_head_desc),' \
          'sizeof(ydet_u8_image), sizeof(ydet_preproc_desc));return 0;}'
    exe = os.path.join(ROOT, "build", "abi_sizes")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    sizes = [int(v) for v in subprocess.run([exe], capture_output=True, check=True).stdout.split()]
    import ctypes as C
    assert sizes == [C.sizeof(_ffi.ConvDesc), C.sizeof(_ffi.SliceDesc), C.sizeof(_ffi.V5Level), C.sizeof(_ffi.FALevel),
                     C.sizeof(_ffi.Op), C.sizeof(_ffi.HeadDesc), C.sizeof(_ffi.U8Image), C.sizeof(_ffi.PreprocDesc)]


def test_no_gpu_means_loud_failure():
    from yolo_amd.yolov5_official.utils import non_max_suppression
    with pytest.raises(RuntimeError):
        non_max_suppression(torch.zeros(1, 8, 85))
    net = build_fa("v3tiny", 64)
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 64, 64))


def test_product_does_not_import_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "yolo_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "oracle" not in open(os.path.join(dirpath, f)).read(), f


# ---- structure ---------------------------------------------------------------------------------------------------
def test_state_dict_structure_matches_reference():
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "g8_structure.json")))
    keys = lambda m: [[k, list(v.shape)] for k, v in m.state_dict().items()]
    assert keys(build_v5s(80)) == ref["yolov5s_nc80"]
    for tag, size in (("v3tiny", 416), ("v2d19", 416), ("v2r50", 416), ("v2slim", 416), ("v3spp", 608), ("v3", 416),
                      ("v4", 608)):
        assert keys(build_fa(tag, size)) == ref[tag], tag


def test_own_v5_structure_and_torch_face(golden):
    """models/v5.py and models/v5v2.py mirrors: state_dict keys/shapes and the stock-torch forward."""
    from yolo_amd.synthetic import images, init_deterministic
    from yolo_amd.models.v5 import YOLOv5 as V5
    from yolo_amd.models.v5v2 import YOLOv5 as V5v2
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "g9_structure.json")))
    anchors9 = [[10, 13], [16, 30], [33, 23], [30, 61], [62, 45], [59, 119], [116, 90], [156, 198], [373, 326]]
    for tag, net in (("v5", V5(nc=20)), ("v5v2", V5v2(nc=20, anchors=anchors9, input_size=128))):
        assert [[k, list(v.shape)] for k, v in net.state_dict().items()] == ref[tag], tag
        init_deterministic(net, seed=0)
        net.eval()
        g = golden(f"g9_{tag}_128")
        with torch.no_grad():
            outs = net(images(2, 128, seed=129, kind="randn"))
        for o, name in zip(outs, ("ys", "ym", "yl")):
            np.testing.assert_allclose(o.numpy(), g[name], rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError):    # the reference's default anchor list cannot be reshaped either
        V5v2(nc=20)


def test_planner_own_v5_on_cpu():
    from yolo_amd import engine
    from yolo_amd.models._anchor_detector import _HeadLowerOps
    from yolo_amd.models.v5 import YOLOv5 as V5
    net = V5(nc=20).eval()
    g = engine.Graph("cpu", "bf16")
    heads = net._heads(_HeadLowerOps(g), g.input_nchw(1, 3, 128, 128, focus=True))
    kinds = [k for k, _ in g.ops]
    assert "copy" not in kinds and kinds.count("spp") == 1 and kinds.count("upsample") == 2
    assert sum(1 for k, a in g.ops if k == "conv" and a["post"] is not None) == 8   # one per CSPModule
    assert [h.C for h in heads] == [75, 75, 75]
    from yolo_amd.models.v5v2 import YOLOv5 as V5v2
    anchors9 = [[10, 13], [16, 30], [33, 23], [30, 61], [62, 45], [59, 119], [116, 90], [156, 198], [373, 326]]
    net = V5v2(nc=20, anchors=anchors9, input_size=128).eval()
    g = engine.Graph("cpu", "fp32")
    heads = net._heads(_HeadLowerOps(g), g.input_nchw(1, 3, 128, 128, focus=True))
    kinds = [k for k, _ in g.ops]
    assert "copy" not in kinds and kinds.count("spp") == 1 and [h.C for h in heads] == [75, 75, 75]
    assert sum(1 for k, a in g.ops if k == "conv" and a["post"] is not None) == 3       # the three CSPModule1


def test_v5s_torch_face_matches_golden(golden):
    """CPU tensors / training mode run stock torch ops with the reference's module semantics."""
    from yolo_amd.synthetic import images
    g = golden("g6_v5s_nc20_64")
    m = build_v5s(20)
    with torch.no_grad():
        z, xs = m(images(2, 64, seed=64))
    np.testing.assert_allclose(z.numpy(), g["z"], rtol=1e-5, atol=1e-5)
    m.train()
    out = m(images(2, 64, seed=64))
    assert isinstance(out, list) and tuple(out[0].shape) == (2, 3, 8, 8, 25)


def test_planner_v5s_on_cpu():
    """The launch list can be planned without a GPU: op mix, concat by redirection, FLOP bookkeeping."""
    from yolo_amd import engine
    m = build_v5s(80)
    g = engine.Graph("cpu", "bf16")
    m._lower(g, 2, 3, 640, 640)
    plan = g.finalize()
    kinds = [k for k, _ in g.ops]
    # 62 convolutions: 8 C3 (cv1, cv2) pairs fused, the first one fused with the Focus slicing and the NCHW edge, the
    # 32- and 64-channel bottlenecks (1x1 + 3x3: one in the first C3, three in the second, one in the last neck C3 at
    # 80 x 80) one kernel each, the three prediction convs decode in place
    # (and the Focus layer takes that first pair into its own kernel: 37 plain convolution launches remain)
    assert kinds[0] == "focus_pair" and kinds.count("conv") == 37 and kinds.count("bottleneck") == 4 and kinds.count("c3tail") == 1
    assert kinds.count("head_decode") == 3 and "copy" not in kinds
    assert kinds.count("spp") == 1 and kinds.count("upsample") == 0      # both upsamples ride in the 1x1 convs behind them
    tail = [op for op in plan.ops if op.kind == 12][0]        # first C3: bottleneck + concat + cv3 in one kernel
    assert (tail.conv.x_ld, tail.conv.x_coff, tail.conv2.res_ld, tail.conv3.x_ld, tail.conv3.x_coff, tail.conv3.Cin,
            tail.conv3.y_ld, tail.conv3.y_coff) == (32, 0, 32, 32, 0, 64, 64, 0)
    stem = plan.ops[0]       # Focus + Conv(32, 64, 3, 2) with the C3's sibling pair (cv1 | cv2) run on its tile
    assert stem.kind == 14 and (stem.conv.Cin, stem.conv.Cout, stem.conv.kp, stem.conv.H, stem.conv.y) == (3, 32, 160, 640, None)
    sib = stem.conv2
    assert (sib.KH, sib.stride, sib.Cin, sib.Cout, sib.pw_cout, sib.pw_kp, sib.H, sib.Ho) == (3, 2, 32, 64, 64, 64, 320, 160)
    assert (sib.y_split, sib.y_ld, sib.y2_ld, sib.y2_coff) == (32, 32, 32, 0) and sib.pw_w and not sib.x
    bn = [op for op in plan.ops if op.kind == 11][2]          # second C3 (64 channels): its last bottleneck writes concat half 1
    assert (bn.conv.x_ld, bn.conv.x_coff, bn.conv2.y_ld, bn.conv2.y_coff, bn.conv2.res_ld) == (64, 0, 128, 0, 64)
    gflop = sum(i["flops"] for i in plan.op_info) / 2 / 1e9
    assert abs(gflop - 16.96) < 0.01          # conv FLOPs per 640x640 image of YOLOv5s
    g32 = engine.Graph("cpu", "fp32")      # the fp32 path keeps the layout kernel + a regular first convolution
    m._lower(g32, 2, 3, 640, 640)
    assert [k for k, _ in g32.ops][:2] == ["nchw2nhwc", "conv"]
    heads = [op for op in plan.ops if op.kind == 10]     # prediction convs with the decode in their epilogue
    assert len(heads) == 3 and all(h.conv.Cout == 255 and h.conv.cout_store == 256 and h.conv.y_ld == 256 for h in heads)
    assert [h.head.level_row_offset for h in heads] == [0, 19200, 24000] and heads[0].head.rows_per_image == 25200
    assert set(g.outputs) == {"z", "row_max", "head0", "head1", "head2"}


@pytest.mark.parametrize("tag,size,convs", [("v3tiny", 416, 13), ("v2d19", 416, 23), ("v3spp", 608, 75), ("v4", 608, 120),
                                          ("v2r50", 416, 26), ("v2slim", 416, 15)])
def test_planner_family_a_on_cpu(tag, size, convs):
    from yolo_amd import engine
    from yolo_amd.models._anchor_detector import _HeadLowerOps
    net = build_fa(tag, size)
    g = engine.Graph("cpu", "fp32")
    heads = net._heads(_HeadLowerOps(g), g.input_nchw(1, 3, size, size))
    kinds = [k for k, _ in g.ops]
    assert kinds.count("conv") == convs and "copy" not in kinds
    assert len(heads) == len(FA_SPECS[tag][3])


# ---- host-side metric code against the golden vectors -------------------------------------------------------------
def test_v5_metric_helpers(golden):
    from yolo_amd.yolov5_official import utils as U
    g = golden("g7_v5_metrics")
    np.testing.assert_allclose(U.box_iou(torch.from_numpy(g["box1"]), torch.from_numpy(g["box2"])).numpy(), g["iou"], rtol=1e-6)
    np.testing.assert_array_equal(U.xywh2xyxy(torch.from_numpy(g["xywh"])).numpy(), g["xyxy"])
    np.testing.assert_allclose(U.xyxy2xywh(torch.from_numpy(g["xyxy"])).numpy(), g["xywh_back"], rtol=1e-6)
    np.testing.assert_allclose(U.scale_coords((640, 640), torch.from_numpy(g["coords"]).clone(), (480, 360)).numpy(), g["scaled1"], rtol=1e-6)
    np.testing.assert_allclose(U.scale_coords((384, 640), torch.from_numpy(g["coords"]).clone(), (720, 1280),
                                              ((0.5, 0.5), (0.0, 12.0))).numpy(), g["scaled2"], rtol=1e-6)
    correct = U.process_batch(torch.from_numpy(g["preds"]), torch.from_numpy(g["labels"]), torch.from_numpy(g["iouv"]))
    np.testing.assert_array_equal(correct.numpy(), g["correct"])
    p, r, ap, f1, cls = U.ap_per_class(g["tp"], g["conf"], g["pcls"], g["tcls"])
    np.testing.assert_allclose(ap, g["ap"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(p, g["p"], rtol=1e-12)
    np.testing.assert_allclose(r, g["r"], rtol=1e-12)
    np.testing.assert_allclose(f1, g["f1"], rtol=1e-12)
    np.testing.assert_array_equal(cls, g["ap_class"])
    one = U.compute_ap(g["rec"], g["prec"])
    np.testing.assert_allclose(one[0], g["ap_one"], rtol=1e-12)
    np.testing.assert_allclose(one[1], g["mpre"], rtol=1e-12)


def _voc_fixture(golden):
    from yolo_amd.data.vocapi_evaluator import VOCAPIEvaluator
    g = golden("g7_voc")
    names = [str(n) for n in g["names"]]
    labelmap = [str(c) for c in g["labelmap"]]
    gts = json.loads(str(g["gts"]))
    ds = type("DS", (), {"ids": [("", n) for n in names], "__len__": lambda self: len(names)})()
    ev = VOCAPIEvaluator("/nonexistent/", 416, None, labelmap, dataset=ds, annotations=gts)
    boxes = [[g[f"boxes_{ci}_{ii}"] for ii in range(len(names))] for ci in range(len(labelmap))]
    return g, ev, boxes


def test_voc_metric(golden):
    from yolo_amd.data.vocapi_evaluator import voc_ap
    g, ev, boxes = _voc_fixture(golden)
    pr = golden("g7_v5_metrics")
    assert abs(voc_ap(pr["rec"], pr["prec"], True) - float(g["voc_ap07"])) < 1e-12
    assert abs(voc_ap(pr["rec"], pr["prec"], False) - float(g["voc_ap_area"])) < 1e-12
    assert abs(ev.evaluate_detections(boxes) - float(g["map"])) < 1e-12
    np.testing.assert_allclose(ev.aps, g["aps"], rtol=0, atol=1e-12)
    assert ev.aps[3] == -1.0
    ev.evaluate_detections(boxes, use_07=False)
    np.testing.assert_allclose(ev.aps, g["aps_area"], rtol=0, atol=1e-12)


# ---- data-parallel exchange: 2 ranks over gloo ---------------------------------------------------------------------
WORKER = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from yolo_amd import distributed as yd
rank, world, _ = yd.init_from_env("cpu")
assert world == 2 and yd.world() == (rank, 2)
# shard_range partitions without gaps
n = 37
lo, hi = yd.shard_range(n, rank, world)
parts = [None, None]; dist.all_gather_object(parts, (lo, hi))
assert parts[0][0] == 0 and parts[0][1] == parts[1][0] and parts[1][1] == n
# ragged all-gather of per-image detections
g = torch.Generator().manual_seed(5)
det_all = torch.rand(n, 300, 6, generator=g); cnt_all = torch.randint(0, 300, (n,), generator=g, dtype=torch.int32)
ids = torch.arange(lo, hi)
det, cnt, iid = yd.all_gather_detections(det_all[lo:hi], cnt_all[lo:hi], ids)
assert torch.equal(det, det_all) and torch.equal(cnt, cnt_all) and torch.equal(iid, torch.arange(n))
# empty shard on one rank
d2 = yd.all_gather_rows(det_all[:5] if rank == 0 else det_all[:0])
assert torch.equal(d2, det_all[:5])
# VOC evaluator: evaluate() shards the image list, detects rank-locally, all-gathers, AP identical to one process
from conftest import load_golden
from test_host import _voc_fixture
gold, ev, boxes = _voc_fixture(load_golden)
class DS:
    ids = ev.dataset.ids
    def __len__(self): return len(self.ids)
    def pull_item(self, i): return torch.full((3, 4, 4), float(i)), None, 1, 1
ev.dataset = DS()
class FakeNet:
    seen = []
    def eval(self): return self
    def detect_batch(self, x):
        idx = [int(v) for v in x[:, 0, 0, 0].tolist()]
        FakeNet.seen += idx
        rows = []
        for i in idx:
            r = [np.concatenate([boxes[ci][i], np.full((len(boxes[ci][i]), 1), ci, np.float32)], 1)
                 for ci in range(len(ev.labelmap)) if len(boxes[ci][i])]
            rows.append(np.concatenate(r, 0) if r else np.zeros((0, 6), np.float32))
        k = max(max(len(r) for r in rows), 1)
        d = torch.zeros(len(idx), k, 6); keep = torch.zeros(len(idx), k, dtype=torch.int32)
        for j, r in enumerate(rows):
            d[j, :len(r)] = torch.from_numpy(r); keep[j, :len(r)] = 1
        return d[..., :4], keep, d[..., 5].long(), d[..., 4]
mAP = ev.evaluate(FakeNet(), batch_size=2, device="cpu")
lo, hi = yd.shard_range(len(DS.ids), rank, world)
assert FakeNet.seen == list(range(lo, hi)), (FakeNet.seen, lo, hi)      # each rank detected only its shard
assert abs(mAP - float(gold["map"])) < 1e-12
np.testing.assert_allclose(ev.aps, gold["aps"], rtol=0, atol=1e-12)
dist.barrier(); dist.destroy_process_group()
open(os.path.join(sys.argv[2], f"ok{rank}"), "w").write("ok")
'''


def test_two_rank_gloo_exchange(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", str(script), ROOT, str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


VAL_WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch
from oracle import postproc as O
from yolo_amd import distributed as yd
from yolo_amd.synthetic import clustered_predictions
from yolo_amd.yolov5_official import val as V

rank, world, _ = yd.init_from_env("cpu")
N_IMG, BS, NC = 10, 2, 4
pred_all = clustered_predictions(N_IMG, 300, NC, seed=7, img_size=64.0, obj_pass=0.2)


class FakeModel(torch.nn.Module):       # "forward" looks the decoded predictions up by the image index in pixel 0
    names = [str(i) for i in range(NC)]
    seen = []
    def forward(self, img, augment=False):
        idx = (img[:, 0, 0, 0] * 255.0).round().long()
        FakeModel.seen += idx.tolist()
        return pred_all[idx], None


def cpu_nms_padded(pred, conf, iou, labels=(), multi_label=False, agnostic=False, max_det=300):   # the GPU operator's stand-in
    out = O.non_max_suppression(pred.numpy(), conf, iou, multi_label=multi_label, agnostic=agnostic, max_det=max_det)
    det = torch.zeros(len(out), max_det, 6); cnt = torch.zeros(len(out), dtype=torch.int32)
    for i, o in enumerate(out):
        det[i, :len(o)] = torch.from_numpy(o); cnt[i] = len(o)
    return det, cnt


V.non_max_suppression_padded = cpu_nms_padded
g = torch.Generator().manual_seed(3)
batches = []
for b in range(N_IMG // BS):
    img = torch.zeros(BS, 3, 64, 64, dtype=torch.uint8)
    tg = []
    for k in range(BS):
        i = b * BS + k
        img[k, :, 0, 0] = i
        top = pred_all[i][pred_all[i][:, 4] > 0.5][:3]           # labels near some confident predictions
        for r in top:
            tg.append([k, float(r[5:].argmax()), *(r[:4] / 64.0).tolist()])
    batches.append((img, torch.tensor(tg).reshape(-1, 6), [f"im{b}_{k}.jpg" for k in range(BS)], [((64, 64), ((1.0, 1.0), (0.0, 0.0)))] * BS))
mode = sys.argv[3] if len(sys.argv) > 3 else "list"
loader = batches
if mode == "dataloader":      # a torch DataLoader over an indexable dataset: re-built per rank over its own image range
    class DS(torch.utils.data.Dataset):
        touched = []
        def __len__(self):
            return N_IMG
        def __getitem__(self, i):
            DS.touched.append(i)
            b, k = divmod(i, BS)
            img, tg, paths, shapes = batches[b]
            return img[k], tg[tg[:, 0] == k, 1:], paths[k], shapes[k]
    def collate(items):
        imgs, tgs, paths, shapes = zip(*items)
        tg = torch.cat([torch.cat([torch.full((len(t), 1), float(k)), t], 1) for k, t in enumerate(tgs)], 0)
        return torch.stack(imgs), tg, list(paths), list(shapes)
    loader = torch.utils.data.DataLoader(DS(), batch_size=BS, collate_fn=collate)
elif mode == "generator":
    loader = (b for b in batches)
res, maps, _ = V.run(dict(nc=NC), model=FakeModel(), dataloader=loader, device="cpu", half=False, conf_thres=0.05, iou_thres=0.5)
own = list(range(*[x * BS for x in yd.shard_range(len(batches), rank, world)]))
if mode == "dataloader":
    assert sorted(set(DS.touched)) == own, (DS.touched, own)     # a rank never loads an image it does not own
assert FakeModel.seen == own, (FakeModel.seen, own)           # a rank runs only its contiguous range of images
json.dump(dict(res=[float(v) for v in res], maps=[float(v) for v in maps]), open(os.path.join(sys.argv[2], f"val{rank}_{world}_{mode}.json"), "w"))
if world > 1:
    torch.distributed.barrier(); torch.distributed.destroy_process_group()
'''


def test_two_rank_gloo_val_run_equals_one_process(tmp_path):
    """val.run sharded over two gloo ranks (contiguous image ranges, one packed detection all-gather) returns the
    metrics of the single-process run, on every rank."""
    script = tmp_path / "val_worker.py"
    script.write_text(VAL_WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r1 = subprocess.run([sys.executable, str(script), ROOT, str(tmp_path)], capture_output=True, text=True, env=env, timeout=240)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    import json as _json
    one = _json.load(open(tmp_path / "val0_1_list.json"))
    assert one["res"][2] > 0.0, "the synthetic set must produce a non-trivial mAP"
    # a list of batches (walked lazily), a torch DataLoader over an indexable dataset (re-built per rank over its own image
    # range; every rank knows the shard sizes, so the gather needs no size exchange) and a generator (drawn first)
    for port, mode in ((29613, "list"), (29615, "dataloader"), (29617, "generator")):
        r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                             "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, str(tmp_path), mode],
                            capture_output=True, text=True, env=dict(env, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)), timeout=240)
        assert r2.returncode == 0, mode + r2.stdout[-2000:] + r2.stderr[-4000:]
        for rank in (0, 1):
            two = _json.load(open(tmp_path / f"val{rank}_2_{mode}.json"))
            assert two == one, mode


def test_bench_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks, which rendezvous (gloo here), run the
    detection all-gather on their own detections and print ONE line from rank 0 with n_gpus = 2 and both ranks seen in
    the gathered image ids; a WORLD_SIZE that contradicts --gpus fails loudly.  (--mock-gpu: this machine has no GPU,
    the step itself is a stand-in.)"""
    import json as _json
    env = dict(os.environ, YDET_DIST_BACKEND="gloo", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mock-gpu", "--batch", "6", "--steps", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = _json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["launcher"] == "self" and j["mock"] is True
    ex = j["exchange"]
    assert ex["backend"] == "gloo" and ex["ranks_seen"] == 2 and ex["images"] == 12 and 0.0 < ex["map50"] <= 1.0
    bad = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, WORLD_SIZE="3", RANK="0"), timeout=60)
    assert bad.returncode != 0 and "WORLD_SIZE=3" in bad.stderr
    one = subprocess.run(cmd[:2] + ["--gpus", "1", "--mock-gpu", "--batch", "4", "--steps", "1"], capture_output=True, text=True,
                         env=env, timeout=120)
    assert one.returncode == 0 and _json.loads(one.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_results_file_rounding_without_strings():
    """The vectorised rounding equals the reference's write-then-parse round trip ('{:.3f}' score, '{:.1f}' of x + 1),
    including exact halves (x.25 -> half to even on the exact value)."""
    from yolo_amd.data.vocapi_evaluator import round_like_results_file, round_like_results_file_fast
    rng = np.random.default_rng(0)
    d = np.concatenate([rng.uniform(-5, 700, size=(4000, 5)), rng.integers(-8, 2800, size=(1000, 5)) / 4.0,
                        rng.integers(0, 4000, size=(1000, 5)) / 2000.0 + 0.0005]).astype(np.float32)
    d[:, 4] = np.abs(d[:, 4]) % 1.0
    np.testing.assert_array_equal(round_like_results_file_fast(d), round_like_results_file(d))


def test_planner_folds_nearest_upsamples_into_pointwise_convs():
    from yolo_amd import engine
    m = build_v5s(80)
    for comp in ("bf16", "fp32"):
        g = engine.Graph("cpu", comp)
        m._lower(g, 2, 3, 640, 640)
        assert [k for k, _ in g.ops].count("upsample") == 2
        g.fuse_upsamples()
        kinds = [k for k, _ in g.ops]
        assert kinds.count("upsample") == 0
        ups = [(a["x"].C, a["up"][1], a["up"][0].H * 2 == a["x"].H) for k, a in g.ops if k == "conv" and a.get("up")]
        assert ups == [(512, 256, True), (256, 128, True)]
        assert [g.ops[i][0] for i in g.ext_in] in (["focus_conv"], ["nchw2nhwc"])


def test_v5s_fuse_and_augment_torch_face_match_golden(golden):
    """Model.fuse() (yolov5_official/model.py:258-265) and forward(augment=True) (:208-251) on CPU tensors against the
    reference's own outputs; a fused model still plans (every block reads its BatchNorm through a guard)."""
    from yolo_amd import engine
    from yolo_amd.yolov5_official.modules import Conv
    g = golden("g11_v5s_fuse_augment")
    m = build_v5s(20)
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        np.testing.assert_allclose(m(x)[0].numpy(), g["z_plain"], rtol=1e-4, atol=1e-4)
        za, none = m(x.clone(), augment=True)
        assert none is None
        np.testing.assert_allclose(za.numpy(), g["z_aug"], rtol=1e-4, atol=2e-4)
        m.fuse()
        assert not any(hasattr(mod, "bn") for mod in m.modules() if type(mod) is Conv)
        np.testing.assert_allclose(m(x)[0].numpy(), g["z_fused"], rtol=1e-4, atol=2e-4)
    for comp in ("bf16", "fp32"):
        gr = engine.Graph("cpu", comp)
        m._lower(gr, 2, 3, 96, 128)
        kinds = [k for k, _ in gr.finalize().g.ops]
        assert kinds.count("head_decode") == 3 and (comp == "fp32" or kinds.count("c3tail") == 1)


def test_kernel_labels_mirror_the_dispatcher(monkeypatch):
    """engine.conv_kernel_label names the kernel the library picks for a shape (the benchmark groups its per-op times by
    it): streaming 1x1 kernel for 128 / 256 input channels over at least 65536 pixels (bf16, no residual, stride 1),
    256 x 256 tile for wide layers with a long K and about a tile per CU, 128-byte-row form of the 128 x 128 tile for few
    tiles and a long K, wide tile where the grid fits, else 128 x 128."""
    from yolo_amd import _ffi, engine
    B, F = _ffi.BF16, _ffi.F32
    lab = engine.conv_kernel_label
    monkeypatch.delenv("YDET_PW", raising=False)
    monkeypatch.delenv("YDET_CONV_K8", raising=False)
    assert lab(128, 128, 1, 1, 0, B, B, M=409600) == "conv_pw<k128>"
    assert lab(256, 256, 1, 1, 0, B, B, M=102400) == "conv_pw<k256>"
    assert lab(256, 256, 1, 1, 0, B, B, M=25600).startswith("conv_dma<")          # too few pixels
    assert lab(128, 128, 1, 1, 0, B, B, M=409600, res=True).startswith("conv_dma<")
    assert lab(128, 128, 1, 1, 0, B, B, M=409600, stride=2).startswith("conv_dma<")
    assert lab(128, 64, 1, 1, 0, B, B, M=409600).startswith("conv_dma<64x128")     # narrow outputs keep their tile
    assert lab(128, 128, 1, 1, 0, F, F, M=409600).startswith("conv_dma<")          # fp32 plans
    assert lab(512, 256, 1, 1, 0, B, B, M=102400).startswith("conv_dma<")          # 512 input channels
    monkeypatch.setenv("YDET_PW", "0")
    assert lab(128, 128, 1, 1, 0, B, B, M=409600).startswith("conv_dma<")
    monkeypatch.delenv("YDET_PW")
    # 3x3, 256 -> 512, stride 2 at 40x40 (25600 output pixels, K = 2304): 200 tiles of 256 x 256 -> the sixteen-wave tile;
    # without it 800 tiles of 128 x 128 and a long K -> 128-byte rows; without either the wide tile
    monkeypatch.delenv("YDET_CONV_256", raising=False)
    assert lab(256, 512, 3, 3, 1, B, B, M=25600, stride=2) == "conv_dma<256x256,bf16->bf16>"
    assert lab(256, 256, 3, 3, 1, B, B, M=25600, res=True) == "conv_dma<128x128,bf16->bf16>"      # 100 tiles: too few
    assert lab(128, 128, 3, 3, 1, B, B, M=102400, res=True) == "conv_dma<128x256,bf16->bf16>"
    monkeypatch.setenv("YDET_CONV_256", "0")
    assert lab(256, 512, 3, 3, 1, B, B, M=25600, stride=2) == "conv_dma<128x128,bf16->bf16>"
    monkeypatch.setenv("YDET_CONV_K8", "0")
    assert lab(256, 512, 3, 3, 1, B, B, M=25600, stride=2) == "conv_dma<128x256,bf16->bf16>"

def test_kernel_label_mirror_equals_library_dispatch():
    """engine.conv_kernel_label (what bench.py keys its per-kernel times and its `roofline` by) against the library's own
    answer, ydet_conv_kernel_name -- the dispatch of csrc/conv.hip / conv_pw.hip / conv_patch.hip run dry, host arithmetic
    only -- over a grid of layer shapes: batch 1 to 64, maps of 13 to 160 pixels, 16 to 1024 channels, 1x1 / 3x3 / stride 2,
    with and without a residual, the three dtype pairs."""
    import ctypes as C
    from yolo_amd import _ffi, engine
    lib = _ffi.lib()

    def library_label(N, S, cin, cout, k, s, res, in_dt, out_dt):
        d = _ffi.ConvDesc()
        p = k // 2
        Ho = (S + 2 * p - k) // s + 1
        d.x = d.w = d.bias = d.y = 4096          # (never dereferenced: nothing is launched)
        d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo = N, S, S, cin, cout, k, k, s, p, Ho, Ho
        vec = 8 if out_dt == _ffi.BF16 else 4
        bk = 32 if in_dt == _ffi.BF16 else 16
        d.x_ld, d.y_ld = cin, (cout + vec - 1) // vec * vec
        d.kp, d.cout_pad = (k * k * cin + bk - 1) // bk * bk, (cout + 127) // 128 * 128
        d.act, d.in_dtype, d.out_dtype = _ffi.ACT_SILU, in_dt, out_dt
        if cout % vec:
            d.cout_store = d.y_ld
        if res:
            d.res, d.res_ld, d.res_mode = 4096, d.y_ld, _ffi.RES_AFTER_ACT
        buf = C.create_string_buffer(96)
        _ffi.check(lib.ydet_conv_kernel_name(C.byref(d), buf, 96), "ydet_conv_kernel_name")
        return buf.value.decode(), Ho

    n = 0
    for N in (1, 32, 64):
        for S in (13, 19, 20, 38, 40, 76, 80, 152):
            for cin in (32, 64, 96, 128, 256, 512, 1024):
                for cout in (32, 64, 75, 128, 200, 256, 512, 1024):
                    for k, s in ((1, 1), (3, 1), (3, 2)):
                        for res in (False, True):
                            for in_dt, out_dt in ((_ffi.BF16, _ffi.BF16), (_ffi.BF16, _ffi.F32), (_ffi.F32, _ffi.F32)):
                                if N * S * S * max(cin, cout) > 3e8 or (res and (s != 1 or out_dt != in_dt)):
                                    continue
                                got, Ho = library_label(N, S, cin, cout, k, s, res, in_dt, out_dt)
                                want = engine.conv_kernel_label(cin, cout, k, k, k // 2, in_dt, out_dt, M=N * Ho * Ho, stride=s,
                                                                res=res, geom=(N, S, S))
                                assert got == want, (N, S, cin, cout, k, s, res, in_dt, out_dt, got, want)
                                n += 1
    assert n > 5000

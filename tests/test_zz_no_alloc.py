"""The NMS entry point that allocates nothing (ydet_nms_v5_static: workspace = a device array built into the library)
against the caller-workspace entry point ydet_nms_v5, through the C ABI: bit-identical outputs on two calls in a row, the
second with a smaller input (the shared array keeps the first call's contents), and a loud error beyond its size."""
import pytest
import torch

from yolo_amd import _ffi, ops
from yolo_amd.synthetic import clustered_predictions

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _static_nms(pred, conf, iou, multi_label, max_det=300, classes=None, row_max=None):
    lib = _ffi.lib()
    N, nboxes, no = pred.shape
    det = torch.full((N, max_det, 6), -7.0, device=pred.device)          # outputs are the caller's; poisoned on purpose
    cnt = torch.full((N,), -7, device=pred.device, dtype=torch.int32)
    mask = None
    if classes is not None:
        mask = torch.zeros(no - 5, dtype=torch.uint8)
        mask[list(classes)] = 1
        mask = mask.to(pred.device)
    with torch.cuda.device(pred.device):
        rc = lib.ydet_nms_v5_static(pred.data_ptr(), N, nboxes, no, conf, iou, int(multi_label), 0, max_det, max_det, 4096.0,
                                    mask.data_ptr() if mask is not None else None,
                                    row_max.data_ptr() if row_max is not None else None,
                                    det.data_ptr(), cnt.data_ptr(), _ffi.current_stream())
    return rc, det, cnt


def test_static_scratch_nms_equals_workspace_nms():
    big = clustered_predictions(64, 25200, 80, seed=31, obj_pass=0.3).to(DEV)       # the benchmark's shape: fills the array
    small = clustered_predictions(3, 1000, 20, seed=32, obj_pass=0.5).to(DEV)
    tiny = clustered_predictions(1, 64, 1, seed=33, obj_pass=0.5).to(DEV)
    for pred, kw in ((big, dict(conf=0.001, iou=0.6, multi_label=True)),
                     (small, dict(conf=0.25, iou=0.45, multi_label=False)),        # right after the big one: stale scratch
                     (small, dict(conf=0.05, iou=0.45, multi_label=True, classes=[1, 4])),
                     (tiny, dict(conf=0.001, iou=0.6, multi_label=True)),
                     (big, dict(conf=0.25, iou=0.45, multi_label=False))):
        before = torch.cuda.memory_allocated(DEV)
        rc, det, cnt = _static_nms(pred, **kw)
        assert rc == 0, _ffi.lib().ydet_last_error()
        ref_det, ref_cnt = ops.nms_v5(pred, kw["conf"], kw["iou"], multi_label=kw["multi_label"], classes=kw.get("classes"),
                                      row_max=None)
        assert torch.equal(cnt, ref_cnt) and torch.equal(det, ref_det)
        assert int(cnt.min()) >= 0 and int(cnt.max()) > 0
        del det, cnt, ref_det, ref_cnt
        assert torch.cuda.memory_allocated(DEV) == before
    # "allocates nothing" at the level of the driver, not just of torch's caching allocator (which never sees a hipMalloc
    # made inside the library): with the outputs allocated beforehand, the free device memory the driver reports is the
    # same before and after calls of either size
    for pred, kw in ((big, dict(conf=0.001, iou=0.6, multi_label=True)), (small, dict(conf=0.25, iou=0.45, multi_label=False))):
        det = torch.empty(pred.shape[0], 300, 6, device=DEV)
        cnt = torch.empty(pred.shape[0], device=DEV, dtype=torch.int32)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(DEV)[0]
        with torch.cuda.device(DEV):
            rc = _ffi.lib().ydet_nms_v5_static(pred.data_ptr(), pred.shape[0], pred.shape[1], pred.shape[2], kw["conf"], kw["iou"],
                                               int(kw["multi_label"]), 0, 300, 300, 4096.0, None, None, det.data_ptr(),
                                               cnt.data_ptr(), _ffi.current_stream())
        torch.cuda.synchronize()
        assert rc == 0 and torch.cuda.mem_get_info(DEV)[0] == free0


def test_static_scratch_nms_rejects_oversize_and_bad_arguments():
    lib = _ffi.lib()
    x = torch.zeros(1, 8, 85, device=DEV)
    out, cnt = torch.zeros(1, 300, 6, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    # 256 images x 25 200 boxes would need 58 MB of workspace: beyond the built-in 16 MiB (pointers are never touched)
    rc = lib.ydet_nms_v5_static(x.data_ptr(), 256, 25200, 85, 0.001, 0.6, 1, 0, 300, 300, 4096.0, None, None,
                                out.data_ptr(), cnt.data_ptr(), _ffi.current_stream())
    assert rc == -3 and b"built-in buffer" in lib.ydet_last_error()
    assert lib.ydet_nms_v5_workspace_bytes(256, 25200, 85) > (16 << 20) >= lib.ydet_nms_v5_workspace_bytes(64, 25200, 85)
    rc = lib.ydet_nms_v5_static(None, 1, 8, 85, 0.001, 0.6, 1, 0, 300, 300, 4096.0, None, None,
                                out.data_ptr(), cnt.data_ptr(), _ffi.current_stream())
    assert rc == -1

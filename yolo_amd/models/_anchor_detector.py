"""Shared inference tail of the v2/v3/v4 detectors.

The reference repeats the same eval branch in every model file (models/v3tiny.py:88-154,
models/v2d19.py:75-120, models/v3spp.py:121-188, models/v4.py:185-259): split each head map into
[A conf | A*nc cls | A*4 box], sigmoid/softmax/exp decode against a (transposed) cell grid with a
fixed multiplier of 32, clamp to [0,1], then argmax-class + per-class greedy NMS of image 0 on the host.
Here the backbone/neck/head convs run as one HIP launch list, the split+decode is one fused kernel on
the NHWC head maps, and the NMS is ydet_nms_perclass -- for every image of the batch.
"""
import torch
import torch.nn as nn

from .. import _ffi, engine, ops
from ..utils.utils import select_kept


class AnchorDetector(engine.AcceleratedModule):
    decode_v2 = True      # utils/utils.py decode_boxes(v2=...) default used by v2d19 / v3spp / v4
    box_multiplier = 32.0  # the reference passes no stride, so every scale is multiplied by 32

    # subclasses: self.input_size, num_classes, num_anchors, conf_thresh, nms_thresh, trainable,
    #             _heads(F, x) -> raw head maps fine -> coarse, _level_anchors() -> per-level flat lists

    def forward(self, x, target=None, eva=False):
        if self.trainable and not eva:
            raise NotImplementedError(
                "training losses are outside this package's scope (inference hot path); "
                "construct with trainable=False or call with eva=True")
        boxes, keep, cls, best = self.detect_batch(x)
        return select_kept(boxes, keep, cls, best, image=0)  # the reference post-processes image 0 only

    @torch.no_grad()
    def detect_batch(self, x):
        """x [N,3,S,S] CUDA -> device tensors (boxes [N,n,4], keep [N,n] u8, cls [N,n] i32, score [N,n]).
        Image i's detections are boxes[i][keep[i]] etc., in box-index order."""
        boxes, scores = self.decode_batch(x)
        keep, cls, best = ops.nms_perclass(boxes, scores, self.conf_thresh, self.nms_thresh)
        return boxes, keep, cls, best

    @torch.no_grad()
    def decode_batch(self, x):
        """x [N,3,S,S] CUDA -> (boxes [N,n,4] normalised + clamped xyxy, scores [N,n,nc] = softmax(cls)*sigmoid(conf))
        for every image: what the reference hands to postprocess() for image 0."""
        if not x.is_cuda:
            raise RuntimeError(f"{type(self).__name__}: inference runs on the GPU only (got a CPU tensor)")
        if self.training:
            raise RuntimeError(f"{type(self).__name__}: call .eval() first (BatchNorm is folded from running stats)")
        return ops.decode_v2v3(self.head_maps_nhwc(x), self._level_anchors(), self.num_anchors, self.num_classes,
                               mode=0 if self.decode_v2 else 1, input_size=float(self.input_size),
                               mult=self.box_multiplier, grid_transposed=True)

    def head_maps_nhwc(self, x):
        """Raw prediction maps as fp32 NHWC views [N, H, W, A*(5+nc)], fine -> coarse."""
        compute = engine.compute_dtype_for(self)
        key = (tuple(x.shape), compute, x.device.index)

        def lower(g, n):
            t = g.input_nchw(n, *x.shape[1:])
            for i, h in enumerate(self._heads(_HeadLowerOps(g), t)):
                g.output(f"head{i}", h)

        def build():
            return engine.build_plan(x.device, compute, x.shape[0], lower)

        outs = self._plans().get(key, build).run(x)
        return [outs[f"head{i}"] for i in range(len(outs))]

    def head_maps(self, x):
        """Raw prediction maps in the reference's NCHW layout (what forward() starts from)."""
        if x.is_cuda and not self.training:
            return [h.permute(0, 3, 1, 2) for h in self.head_maps_nhwc(x)]
        return self._heads(engine.TorchOps, x)


class _HeadLowerOps(engine.LowerOps):
    """LowerOps whose bare nn.Conv2d layers (the prediction convs) write fp32 buffers owned by the caller."""

    def run(self, m, x):
        if isinstance(m, nn.Conv2d):
            return self.g.conv(x, m, out_dtype=_ffi.F32, external=True)
        return super().run(m, x)

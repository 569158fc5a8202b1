"""Inference face of the detectors whose forward() returns the raw prediction maps (the author's own
YOLOv5 re-creations, models/v5.py and models/v5v2.py in the reference): the whole network is one HIP
launch list; the maps come back as NCHW-shaped views of the fp32 NHWC head buffers."""
import torch

from .. import engine
from ._anchor_detector import _HeadLowerOps


class RawHeadDetector(engine.AcceleratedModule):
    focus_input = True   # the stem is a Focus layer: slice while converting NCHW -> NHWC

    # subclasses: _heads(F, x) -> tuple of head maps in the order forward() returns them

    def forward(self, x):
        if not x.is_cuda or self.training:
            return self._heads(engine.TorchOps, x)      # stock torch ops: training and CPU tensors
        compute = engine.compute_dtype_for(self)
        key = (tuple(x.shape), compute, x.device.index)

        def lower(g, n):
            t = g.input_nchw(n, *x.shape[1:], focus=self.focus_input)
            for i, h in enumerate(self._heads(_HeadLowerOps(g), t)):
                g.output(f"head{i}", h)

        def build():
            return engine.build_plan(x.device, compute, x.shape[0], lower)

        with torch.no_grad():
            outs = self._plans().get(key, build).run(x)
        return tuple(outs[f"head{i}"].permute(0, 3, 1, 2) for i in range(len(outs)))

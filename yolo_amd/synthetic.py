"""Deterministic synthetic weights and inputs.

No trained weights exist for the reference models, so parity fixtures, tests and benchmarks all use the
same seeded initialiser: the tensors depend only on (seed, name of the owning module, shape), are drawn with a CPU generator and are therefore identical in every process and on every box.
BatchNorm running statistics are randomised on purpose (so BN folding is exercised) and the objectness
bias of the detection heads is shifted so that the two NMS regimes of the reference are both met:
about 2 % of the boxes pass conf 0.25 and most pass conf 0.001.
"""
import math
import zlib

import torch
import torch.nn as nn


# Fixtures, tests and benchmarks of the residual-trunk detectors scale their prediction convs' weights by these after
# init_deterministic: a residual trunk amplifies activations under random BatchNorm gains (the unscaled head maps of
# the Darknet-53 models reach 1e6), which saturates most scores to exactly 1.0 -- ties whose visiting order numpy's
# argsort leaves open.  tag -> (names of the prediction convs, factor).
FA_HEAD_SCALE = {
    "v2r50": (("pred",), 0.05),
    "v3spp": (("pred_1", "pred_2", "pred_3"), 1e-6),
    "v3": (("pred_1", "pred_2", "pred_3"), 2e-6),
    "v4": (("head_det_1", "head_det_2", "head_det_3"), 4e-6),
}
V2R50_HEAD_SCALE = FA_HEAD_SCALE["v2r50"][1]


@torch.no_grad()
def scale_fa_heads(net, tag):
    """Apply FA_HEAD_SCALE[tag] to `net` (a no-op for detectors without an entry)."""
    names, factor = FA_HEAD_SCALE.get(tag, ((), 1.0))
    for n in names:
        getattr(net, n).weight.mul_(factor)
    return net


def _gen(seed, key):
    """CPU generator seeded by (seed, key); `key` is an int or a name (hashed with crc32)."""
    if isinstance(key, str):
        key = zlib.crc32(key.encode())
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + int(key)) % (2 ** 63))
    return g


@torch.no_grad()
def init_deterministic(model, seed=0, head_convs=None, anchors_per_head=3, obj_index=4, obj_logit_mean=-4.2,
                       obj_logit_std=1.5):
    """Fill every Conv2d / BatchNorm2d parameter and buffer of `model` in place.  Each tensor is seeded by
    its module name, so two models with the same state_dict keys get the same values whatever their
    attribute order.  `head_convs`: bias-only prediction convs laid out per anchor with the objectness
    logit at `obj_index` (YOLOv5 heads): their objectness bias / weights are re-scaled so that the logits
    are roughly N(obj_logit_mean, obj_logit_std)."""
    for name, mod in model.named_modules():
        if isinstance(mod, nn.Conv2d):
            fan_in = mod.in_channels // mod.groups * mod.kernel_size[0] * mod.kernel_size[1]
            mod.weight.copy_(torch.randn(mod.weight.shape, generator=_gen(seed, name + ".weight")) * math.sqrt(2.0 / fan_in))
            if mod.bias is not None:
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=_gen(seed, name + ".bias")) * 0.1)
        elif isinstance(mod, nn.BatchNorm2d):
            c = mod.num_features
            mod.weight.copy_(torch.rand(c, generator=_gen(seed, name + ".weight")) + 0.5)
            mod.bias.copy_(torch.randn(c, generator=_gen(seed, name + ".bias")) * 0.1)
            mod.running_mean.copy_(torch.randn(c, generator=_gen(seed, name + ".running_mean")) * 0.1)
            mod.running_var.copy_(torch.rand(c, generator=_gen(seed, name + ".running_var")) + 0.5)
    for h, conv in enumerate(head_convs or ()):
        no = conv.out_channels // anchors_per_head
        b = conv.bias.view(anchors_per_head, no)
        b[:, obj_index] = obj_logit_mean + torch.randn(anchors_per_head, generator=_gen(seed, 900000 + h)) * 0.1
        w = conv.weight.view(anchors_per_head, no, -1)
        w[:, obj_index] *= obj_logit_std / max(float(w[:, obj_index].norm(dim=1).mean()), 1e-6)
    return model


def images(n, size, seed=0, kind="rand", channels=3):
    """Synthetic image batch [n, channels, size, size] fp32: 'rand' in [0,1) (YOLOv5 callers divide uint8 by
    255), 'randn' standard normal (the v2/v3/v4 callers feed mean/std-normalised images)."""
    g = _gen(seed, 7777)
    shape = (n, channels, size, size)
    return torch.rand(shape, generator=g) if kind == "rand" else torch.randn(shape, generator=g)


def clustered_predictions(n_img, n_boxes, nc, seed=0, clusters=40, img_size=640.0, obj_pass=0.3):
    """[n_img, n_boxes, 5+nc] YOLOv5-style decoded predictions with overlapping box clusters: what an NMS
    benchmark needs (many near-duplicates per object, a controllable fraction above the obj threshold)."""
    g = _gen(seed, 4242)
    cx = torch.rand(n_img, clusters, 1, generator=g) * img_size
    cy = torch.rand(n_img, clusters, 1, generator=g) * img_size
    w = torch.rand(n_img, clusters, 1, generator=g) * 150 + 20
    h = torch.rand(n_img, clusters, 1, generator=g) * 150 + 20
    per = (n_boxes + clusters - 1) // clusters
    jit = lambda s: torch.randn(n_img, clusters, per, generator=g) * s
    box = torch.stack([cx + jit(6.0), cy + jit(6.0), (w + jit(8.0)).abs() + 2, (h + jit(8.0)).abs() + 2], -1)
    box = box.view(n_img, clusters * per, 4)[:, :n_boxes]
    obj = torch.rand(n_img, n_boxes, 1, generator=g)
    obj = torch.where(obj < obj_pass, 0.3 + 0.7 * torch.rand(n_img, n_boxes, 1, generator=g), obj * 0.001)
    cls = torch.rand(n_img, n_boxes, nc, generator=g) ** 4
    return torch.cat([box, obj, cls], -1).float()

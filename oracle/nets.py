"""Functional (state_dict -> outputs) restatement of the reference networks on stock torch CPU ops
(test infrastructure; see oracle/__init__.py).  Nothing here uses the yolo_amd module classes: the
networks are re-derived from the reference's constructors / forward() bodies and read their weights by
state_dict key, so a key or wiring mistake in the product cannot cancel out.
"""
import contextlib
import math

import torch
import torch.nn.functional as F

# ---- bf16 storage floor ------------------------------------------------------------------------------
# Inside `with bf16_storage():` the networks below run as a layer-by-layer bf16 inference pipeline stores them: BatchNorm
# is folded into the convolution weights (w * gamma / sqrt(var + eps), the rest a fp32 bias), the folded weights are
# rounded to bf16, and every tensor a layer hands to the next one (a convolution's activated output, a residual sum, an
# interpolated map) is rounded to bf16 once; sums, BatchNorm shifts and activations are computed in fp32 and the
# prediction maps stay fp32.  The distance of these head maps from the plain fp32 ones is what ANY implementation with
# bf16 weights and activations loses on that network and input -- the floor the bf16 GPU path is measured against.
# (Which weights get rounded matters: rounding the raw weights instead of the folded ones moves YOLOv5s's 640 x 640 head
# maps from 1.3-1.7 % to 2.3-3.2 % relative L2, the error being dominated by a few coherent directions.)
_BF16 = False


@contextlib.contextmanager
def bf16_storage():
    global _BF16
    old, _BF16 = _BF16, True
    try:
        yield
    finally:
        _BF16 = old


def _q(t):
    return t.bfloat16().float() if _BF16 else t


def _conv2d(x, w, b=None, **kw):
    return F.conv2d(_q(x), _q(w), b, **kw)


# ---- building blocks -------------------------------------------------------------------------------
def _bn(x, sd, p, eps):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, eps)


def _conv_bn(x, w, b, sd, bn, eps, **kw):
    """BatchNorm(conv(x) + b); with bf16 storage: the folded form, folded weights and input rounded to bf16."""
    if not _BF16:
        return _bn(F.conv2d(x, w, b, **kw), sd, bn, eps)
    scale = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + eps)
    shift = sd[bn + ".bias"] - sd[bn + ".running_mean"] * scale
    if b is not None:
        shift = shift + b * scale
    return F.conv2d(_q(x), _q(w * scale.view(-1, 1, 1, 1)), shift, **kw)


def conv_a(x, sd, p, s=1, act=True, store=True):
    """utils/moduleUtils.py:8-24: Conv2d(bias) + BN(eps 1e-5) + LeakyReLU(0.1); padding 1 for 3x3, 0 for 1x1.
    store=False: the caller adds something before the tensor is handed on (no bf16 rounding here)."""
    w = sd[p + ".conv.0.weight"]
    y = _conv_bn(x, w, sd[p + ".conv.0.bias"], sd, p + ".conv.1", 1e-5, stride=s, padding=w.shape[-1] // 2)
    y = F.leaky_relu(y, 0.1) if act else y
    return _q(y) if store else y


def conv_b(x, sd, p, s=1, act="silu", eps=1e-3, store=True):
    """yolov5_official/modules.py:24-41: Conv2d(no bias, autopad) + BN(eps 1e-3 after initialize_weights) + SiLU;
    models/common.py:22-37 uses LeakyReLU(0.01) instead."""
    w = sd[p + ".conv.weight"]
    y = _conv_bn(x, w, None, sd, p + ".bn", eps, stride=s, padding=w.shape[-1] // 2)
    if act == "silu":
        y = F.silu(y)
    elif act == "leaky001":
        y = F.leaky_relu(y, 0.01)
    return _q(y) if store else y


def spp_cat(x):
    """utils/moduleUtils.py:63-75 / yolov5_official/modules.py:278-282."""
    return torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)], 1)


# ---- YOLOv5s (yolov5_official/model.py:88-265 + yamls/yolov5s.yaml) -----------------------------------
def _bottleneck_b(x, sd, p, add, act):
    y = conv_b(conv_b(x, sd, p + ".cv1", act=act), sd, p + ".cv2", act=act, store=not add)
    return _q(x + y) if add else y                                                        # modules.py:218


def _c3(x, sd, p, n, shortcut, act):
    y = conv_b(x, sd, p + ".cv1", act=act)
    for i in range(n):
        y = _bottleneck_b(y, sd, f"{p}.m.{i}", shortcut, act)
    return conv_b(torch.cat((y, conv_b(x, sd, p + ".cv2", act=act)), 1), sd, p + ".cv3", act=act)  # :253


def yolov5_heads(sd, x, cfg, act="silu"):
    """Raw Detect conv outputs [bs, na*no, ny, nx] for P3, P4, P5 of a yolov5s-style yaml dict `cfg`."""
    gd = cfg["depth_multiple"]
    rows = cfg["backbone"] + cfg["head"]
    outs = []
    for i, (f, n, m, args) in enumerate(rows):
        if isinstance(f, int):
            inp = x if f == -1 else outs[f]
        else:
            inp = [x if j == -1 else outs[j] for j in f]
        n = max(round(n * gd), 1) if n > 1 else n                                     # model.py:504
        p = f"model.{i}"
        if m == "Focus":                                                              # modules.py:293
            t = torch.cat([inp[..., ::2, ::2], inp[..., 1::2, ::2], inp[..., ::2, 1::2], inp[..., 1::2, 1::2]], 1)
            x = conv_b(t, sd, p + ".conv", act=act)
        elif m == "Conv":
            x = conv_b(inp, sd, p, s=args[2] if len(args) > 2 else 1, act=act)
        elif m == "C3":
            x = _c3(inp, sd, p, n, args[1] if len(args) > 1 else True, act)
        elif m == "SPP":
            x = conv_b(spp_cat(conv_b(inp, sd, p + ".cv1", act=act)), sd, p + ".cv2", act=act)
        elif m == "nn.Upsample":
            x = F.interpolate(inp, scale_factor=2, mode="nearest")
        elif m == "Concat":
            x = torch.cat(inp, 1)
        elif m == "Detect":
            return [_conv2d(t, sd[f"{p}.m.{k}.weight"], sd[f"{p}.m.{k}.bias"]) for k, t in enumerate(inp)]
        else:
            raise NotImplementedError(m)
        outs.append(x)
    raise ValueError("no Detect row")


# ---- family A backbones (models/basemodel/d19.py) ---------------------------------------------------
def _seq_a(x, sd, p, spec):
    """spec: list of ('c', stride) conv entries and 'p' 2x2/2 max-pools, indexed like the nn.Sequential."""
    for i, e in enumerate(spec):
        x = F.max_pool2d(x, 2, 2) if e == "p" else conv_a(x, sd, f"{p}.{i}", s=e[1])
    return x


def dark19(sd, x, p="base"):                                                          # d19.py:17-79
    c = ("c", 1)
    x = _seq_a(x, sd, p + ".conv1", [c, "p"])
    x = _seq_a(x, sd, p + ".conv2", [c, "p"])
    x = _seq_a(x, sd, p + ".conv3", [c, c, c, "p"])
    x1 = _seq_a(x, sd, p + ".conv4", [c, c, c])
    x2 = _seq_a(F.max_pool2d(x1, 2, 2), sd, p + ".conv5", [c] * 5)
    x3 = _seq_a(F.max_pool2d(x2, 2, 2), sd, p + ".conv6", [c] * 5)
    return x1, x2, x3


def _resblock(x, sd, p, n):                                                           # moduleUtils.py:45-59
    for i in range(n):
        x = _q(conv_a(conv_a(x, sd, f"{p}.module_list.{i}.0"), sd, f"{p}.module_list.{i}.1", store=False) + x)
    return x


def darknet53(sd, x, p="base"):                                                       # d19.py:83-123
    x = conv_a(x, sd, p + ".layer_1.0")
    x = _resblock(conv_a(x, sd, p + ".layer_1.1", s=2), sd, p + ".layer_1.2", 1)
    x = _resblock(conv_a(x, sd, p + ".layer_2.0", s=2), sd, p + ".layer_2.1", 2)
    c3 = _resblock(conv_a(x, sd, p + ".layer_3.0", s=2), sd, p + ".layer_3.1", 8)
    c4 = _resblock(conv_a(c3, sd, p + ".layer_4.0", s=2), sd, p + ".layer_4.1", 8)
    c5 = _resblock(conv_a(c4, sd, p + ".layer_5.0", s=2), sd, p + ".layer_5.1", 4)
    return c3, c4, c5


def darknet_light(sd, x, p="base"):                                                   # d19.py:170-212
    for i in range(1, 5):
        x = F.max_pool2d(conv_a(x, sd, f"{p}.conv_{i}"), 2, 2)
    c4 = conv_a(x, sd, p + ".conv_5")
    x = conv_a(F.max_pool2d(c4, 2, 2), sd, p + ".conv_6")
    x = F.max_pool2d(F.pad(x, (0, 1, 0, 1)), 2, 1)                                    # :190-193 zero pad
    return c4, conv_a(x, sd, p + ".conv_7")


def _csp_stage(x, sd, p, n):                                                          # moduleUtils.py:79-107
    y1 = conv_a(x, sd, p + ".conv1")
    y2 = conv_a(x, sd, p + ".conv2")
    for i in range(n):
        q = f"{p}.resblock.{i}"
        y2 = _q(F.leaky_relu(y2 + conv_a(conv_a(y2, sd, q + ".conv1"), sd, q + ".conv2", act=False, store=False), 0.1))
    return conv_a(torch.cat([y1, y2], 1), sd, p + ".conv3")


def cspdarknet(sd, x, p="base"):                                                      # d19.py:215-265
    x = conv_a(x, sd, p + ".layer1.0")
    x = _csp_stage(conv_a(x, sd, p + ".layer1.1", s=2), sd, p + ".layer1.2", 1)
    x = _csp_stage(conv_a(x, sd, p + ".layer2.0", s=2), sd, p + ".layer2.1", 2)
    c3 = _csp_stage(conv_a(x, sd, p + ".layer3.0", s=2), sd, p + ".layer3.1", 8)
    c4 = _csp_stage(conv_a(c3, sd, p + ".layer4.0", s=2), sd, p + ".layer4.1", 8)
    c5 = _csp_stage(conv_a(c4, sd, p + ".layer5.0", s=2), sd, p + ".layer5.1", 4)
    return c3, c4, c5


# ---- family A detectors: raw head maps, fine -> coarse --------------------------------------------
def _pred(x, sd, p):
    return _conv2d(x, sd[p + ".weight"], sd[p + ".bias"])


def _up_bilinear(x):
    return _q(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True))


def v3tiny_heads(sd, x):                                                              # models/v3tiny.py:68-80
    c4, c5 = darknet_light(sd, x)
    y1 = conv_a(c5, sd, "conv_set_2")
    up = _up_bilinear(conv_a(y1, sd, "conv_1x1_2"))
    y2 = conv_a(torch.cat([c4, up], 1), sd, "conv_set_1")
    return [_pred(y2, sd, "pred_1"), _pred(conv_a(y1, sd, "extra_conv_2"), sd, "pred_2")]


def v2d19_heads(sd, x):                                                               # models/v2d19.py:60-72
    _, c5, c6 = dark19(sd, x)
    p6 = conv_a(conv_a(c6, sd, "convsets1.0"), sd, "convsets1.1")
    r = conv_a(c5, sd, "route_layer")
    b, c, h, w = r.shape                                                              # reorg, moduleUtils.py:33-41
    r = r.view(b, c, h // 2, 2, w // 2, 2).transpose(3, 4).contiguous()
    r = r.view(b, c, (h // 2) * (w // 2), 4).transpose(2, 3).contiguous()
    r = r.view(b, c, 4, (h // 2) * (w // 2)).transpose(1, 2).contiguous().view(b, -1, h // 2, w // 2)
    return [_pred(conv_a(torch.cat([r, p6], 1), sd, "convsets2"), sd, "pred")]


def _reorg2(r):                                                                       # moduleUtils.py:33-41
    b, c, h, w = r.shape
    r = r.view(b, c, h // 2, 2, w // 2, 2).transpose(3, 4).contiguous()
    r = r.view(b, c, (h // 2) * (w // 2), 4).transpose(2, 3).contiguous()
    return r.view(b, c, 4, (h // 2) * (w // 2)).transpose(1, 2).contiguous().view(b, -1, h // 2, w // 2)


def _cbr(x, sd, conv, bn, s=1, relu=True, store=True):
    """Conv2d(no bias) + BN(eps 1e-5) [+ ReLU] as the ResNet trunk composes them (basemodel/resnet.py:13-20)."""
    w = sd[conv + ".weight"]
    y = _conv_bn(x, w, None, sd, bn, 1e-5, stride=s, padding=w.shape[-1] // 2)
    y = F.relu(y) if relu else y
    return _q(y) if store else y


def _basic_block(x, sd, p, s):                                                        # basemodel/resnet.py:24-54
    out = _cbr(_cbr(x, sd, p + ".conv1", p + ".bn1", s=s), sd, p + ".conv2", p + ".bn2", relu=False, store=False)
    if p + ".downsample.0.weight" in sd:
        x = _cbr(x, sd, p + ".downsample.0", p + ".downsample.1", s=s, relu=False)
    return _q(F.relu(out + x))


def resnet18_trunk(sd, x, p="base"):                                                  # basemodel/resnet.py:96-155
    """ResNet([2,2,2,2]) as models/v2r50.py:39 builds it; forward() skips self.maxpool (:148-155)."""
    y = _cbr(x, sd, p + ".first_layer.0", p + ".first_layer.1", s=2)
    outs = []
    for li, s in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for b in range(2):
            y = _basic_block(y, sd, f"{p}.layer{li}.{b}", s if b == 0 else 1)
        outs.append(y)
    return outs[1], outs[2], outs[3]


def v2r50_heads(sd, x):                                                               # models/v2r50.py:68-82
    _, c5, c6 = resnet18_trunk(sd, x)
    p6 = conv_a(conv_a(conv_a(c6, sd, "convset1.0"), sd, "convset1.1"), sd, "convset1.2")
    p5 = _reorg2(conv_a(c5, sd, "route_layer"))
    return [_pred(conv_a(torch.cat([p5, p6], 1), sd, "convset2"), sd, "pred")]


def darknet_tiny(sd, x, p="base"):                                                    # d19.py:127-166
    outs = []
    for i in range(1, 6):
        x = conv_a(conv_a(x, sd, f"{p}.conv_{i}.0"), sd, f"{p}.conv_{i}.1", s=2)
        outs.append(x)
    return outs[2], outs[3], outs[4]


def v2slim_heads(sd, x):                                                              # models/v2slim.py:48-60
    _, c5, c6 = darknet_tiny(sd, x)
    p6 = conv_a(conv_a(c6, sd, "convset1.0"), sd, "convset1.1")
    p5 = _reorg2(conv_a(c5, sd, "route_layer"))
    return [_pred(conv_a(torch.cat([p5, p6], 1), sd, "convset2"), sd, "pred")]


def _set5(x, sd, p, first=0):
    for i in range(first, first + 5):
        x = conv_a(x, sd, f"{p}.{i}")
    return x


def v3_heads(sd, x, spp):                                                             # models/v3.py:146-172, v3spp.py:88-113
    c3, c4, c5 = darknet53(sd, x)
    y1 = _set5(spp_cat(c5), sd, "conv_set_3", 1) if spp else _set5(c5, sd, "conv_set_3")
    o1 = _pred(conv_a(y1, sd, "extra_conv_3"), sd, "pred_3")
    y2 = _set5(torch.cat([c4, _up_bilinear(conv_a(y1, sd, "conv_1x1_3"))], 1), sd, "conv_set_2")
    o2 = _pred(conv_a(y2, sd, "extra_conv_2"), sd, "pred_2")
    y3 = _set5(torch.cat([c3, _up_bilinear(conv_a(y2, sd, "conv_1x1_2"))], 1), sd, "conv_set_1")
    o3 = _pred(conv_a(y3, sd, "extra_conv_1"), sd, "pred_1")
    return [o3, o2, o1]


def _csp_bottleneck(x, sd, p, n=3):                                                   # moduleUtils.py:138-171 (shortcut=False)
    y1 = conv_a(x, sd, p + ".conv1")
    for i in range(n):
        y1 = conv_a(conv_a(y1, sd, f"{p}.bottleblock.{i}.conv1"), sd, f"{p}.bottleblock.{i}.conv2")
    w3, w2 = sd[p + ".conv3.weight"], sd[p + ".conv2.weight"]
    if _BF16:      # the BatchNorm behind the concatenation is per channel: each half folds into the convolution producing it
        scale = sd[p + ".bn.weight"] / torch.sqrt(sd[p + ".bn.running_var"] + 1e-5)
        shift = sd[p + ".bn.bias"] - sd[p + ".bn.running_mean"] * scale
        c = w3.shape[0]
        y1 = F.conv2d(_q(y1), _q(w3 * scale[:c].view(-1, 1, 1, 1)), shift[:c])
        y2 = F.conv2d(_q(x), _q(w2 * scale[c:].view(-1, 1, 1, 1)), shift[c:])
        y = _q(F.leaky_relu(torch.cat([y1, y2], 1), 0.1))
    else:
        y = F.leaky_relu(_bn(torch.cat([_conv2d(y1, w3), _conv2d(x, w2)], 1), sd, p + ".bn", 1e-5), 0.1)
    return conv_a(y, sd, p + ".conv4")


def v4_heads(sd, x):                                                                  # models/v4.py:150-179
    c3, c4, c5 = cspdarknet(sd, x)
    c5 = _csp_bottleneck(spp_cat(conv_a(c5, sd, "spp.0")), sd, "spp.2")
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    y0 = conv_a(c5, sd, "head_conv_0")
    p4 = _csp_bottleneck(torch.cat([up(y0), c4], 1), sd, "head_csp_0")
    y1 = conv_a(p4, sd, "head_conv_1")
    p3 = _csp_bottleneck(torch.cat([up(y1), c3], 1), sd, "head_csp_1")
    n4 = _csp_bottleneck(torch.cat([conv_a(p3, sd, "head_conv_2", s=2), y1], 1), sd, "head_csp_2")
    n5 = _csp_bottleneck(torch.cat([conv_a(n4, sd, "head_conv_3", s=2), y0], 1), sd, "head_csp_3")
    return [_pred(p3, sd, "head_det_1"), _pred(n4, sd, "head_det_2"), _pred(n5, sd, "head_det_3")]

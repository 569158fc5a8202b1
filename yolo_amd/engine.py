"""Launch-list planner for the detection forward pass.

A model is lowered ONCE per (batch, height, width, compute dtype) into a flat list of `ydet_op`
records over NHWC device buffers; running the model is then a single C call (`ydet_run_ops`).

Layout decisions
  * activations are NHWC slices `(storage, channel offset, channels)`; a `torch.cat(dim=1)` in the
    reference becomes "producers write into channel slices of one storage" (`Graph.cat` redirects
    the producers' storages, no copy kernel unless a tensor is concatenated twice);
  * BatchNorm is folded into packed weights `[cout_pad][kp]` (k = (kh*KW+kw)*Cin + c) and an fp32
    bias; post-concat BatchNorms (CSP blocks) are folded per half into the producing convs;
  * compute dtype "bf16": bf16 activations/weights, fp32 accumulate, fp32 detection-head outputs;
    compute dtype "fp32": everything fp32 (exact-fp32 MFMA).
"""
import ctypes as C
import math
import os
import threading

import torch

from . import _ffi

_TORCH_DT = {_ffi.F32: torch.float32, _ffi.BF16: torch.bfloat16}


def _round_up(v, m):
    return (v + m - 1) // m * m


class Storage:
    """One NHWC device buffer [N, H, W, ld]; may be redirected into a channel range of another."""

    def __init__(self, N, H, W, ld, dtype, external=False):
        self.N, self.H, self.W, self.ld, self.dtype = N, H, W, ld, dtype
        self.external = external  # pointer supplied per call
        self.parent = None
        self.parent_off = 0
        self.tensor = None

    def root(self):
        st, off = self, 0
        while st.parent is not None:
            off += st.parent_off
            st = st.parent
        return st, off


class T:
    """A channel slice of a storage: `C` meaningful channels starting at `coff`."""

    def __init__(self, st, coff, C):
        self.st, self.coff, self.C = st, coff, C

    N = property(lambda s: s.st.N)
    H = property(lambda s: s.st.H)
    W = property(lambda s: s.st.W)
    dtype = property(lambda s: s.st.dtype)

    def resolve(self):
        root, off = self.st.root()
        return root, off + self.coff

    def whole(self):
        return self.coff == 0 and self.C == self.st.ld


class PackedConv:
    """Device-resident packed weights + bias of one (conv [+ BatchNorm]) for one input layout."""

    def __init__(self, w, b, Cin, Cout, KH, KW, kp, cout_pad, cin_real=None, groups=1, dilation=1):
        self.w, self.b, self.Cin, self.Cout, self.KH, self.KW, self.kp, self.cout_pad = w, b, Cin, Cout, KH, KW, kp, cout_pad
        self.cin_real = cin_real or Cin      # input channels one output channel reads (per group)
        self.groups, self.dilation = groups, dilation

    @property
    def plain(self):
        """One group, no dilation: the shapes the MFMA kernels and every fusion pass handle."""
        return self.groups == 1 and self.dilation == 1


def fold_conv(conv, bn=None, bn_slice=None, device=None):
    """(w [Cout, KH, KW, Cin / groups] fp32, b [Cout] fp32) of `conv` with `bn` (optionally the channel range `bn_slice` of a
    wider BatchNorm) folded in: y = gamma * (conv(x) + bias - mean) / sqrt(var + eps) + beta, as nn.BatchNorm2d
    computes in eval mode."""
    if (conv.kernel_size[0] != conv.kernel_size[1] or conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1]
            or conv.dilation[0] != conv.dilation[1]):
        raise NotImplementedError("only square kernels / strides / paddings / dilations are used by the reference models")
    device = device or conv.weight.device
    w = conv.weight.detach().to(device=device, dtype=torch.float32)
    Cout = w.shape[0]
    b = conv.bias.detach().to(device=device, dtype=torch.float32) if conv.bias is not None else torch.zeros(Cout, device=device)
    if bn is not None:
        sl = slice(*bn_slice) if bn_slice is not None else slice(None)
        g = bn.weight.detach().to(device=device, dtype=torch.float32)[sl]
        beta = bn.bias.detach().to(device=device, dtype=torch.float32)[sl]
        mean = bn.running_mean.detach().to(device=device, dtype=torch.float32)[sl]
        var = bn.running_var.detach().to(device=device, dtype=torch.float32)[sl]
        scale = g / torch.sqrt(var + bn.eps)
        w = w * scale.view(-1, 1, 1, 1)
        b = (b - mean) * scale + beta
    return w.permute(0, 2, 3, 1), b


def pack_folded(w, b, cin_phys=None, dtype=_ffi.BF16, groups=1, dilation=1):
    """Pack folded weights [Cout, KH, KW, Cin] + bias into the kernel's [cout_pad][kp] layout.  With groups > 1 the
    last axis is the group's channels and is packed as it is (k = tap * cin_group + c)."""
    Cout, KH, KW, Cin = w.shape
    epc = 8 if dtype == _ffi.BF16 else 4
    if groups > 1:
        cin_phys = Cin
    cin_phys = cin_phys or Cin
    if cin_phys % epc and groups == 1:
        raise ValueError(f"input channel count {cin_phys} must be a multiple of {epc}")
    if cin_phys > Cin:
        w = torch.nn.functional.pad(w, (0, cin_phys - Cin))
    K = KH * KW * cin_phys
    kp = _round_up(K, 4 * epc)
    cout_pad = _round_up(Cout, 128)
    wp = torch.zeros(cout_pad, kp, device=w.device, dtype=torch.float32)
    wp[:Cout, :K] = w.reshape(Cout, K)
    bp = torch.zeros(cout_pad, device=w.device, dtype=torch.float32)
    bp[:Cout] = b
    return PackedConv(wp.to(_TORCH_DT[dtype]).contiguous(), bp, cin_phys, Cout, KH, KW, kp, cout_pad, cin_real=Cin,
                      groups=groups, dilation=dilation)


def pack_conv(conv, bn=None, bn_slice=None, cin_phys=None, dtype=_ffi.BF16, device=None):
    """Fold `bn` into `conv` and pack (see fold_conv / pack_folded)."""
    w, b = fold_conv(conv, bn, bn_slice, device)
    return pack_folded(w, b, cin_phys, dtype, groups=conv.groups, dilation=conv.dilation[0])


def is_plain(conv):
    """nn.Conv2d with one group and no dilation."""
    return conv.groups == 1 and tuple(conv.dilation) == (1, 1)


class Graph:
    def __init__(self, device, compute="bf16"):
        if compute not in ("bf16", "fp32"):
            raise ValueError("compute dtype must be 'bf16' or 'fp32'")
        self.device = torch.device(device)
        self.dt = _ffi.BF16 if compute == "bf16" else _ffi.F32
        self.vec = 8 if self.dt == _ffi.BF16 else 4
        self.ops = []        # (kind, dict) records, resolved in finalize()
        self.storages = []
        self.keep = []       # tensors the plan must keep alive (packed weights)
        self.ext_in = []     # op indices of the NCHW sources, in argument order
        self.outputs = {}    # name -> T of external outputs

    # ---- storage helpers ---------------------------------------------------------------------
    def new(self, N, H, W, Cch, dtype=None, external=False, ld=None):
        dtype = self.dt if dtype is None else dtype
        st = Storage(N, H, W, ld or _round_up(Cch, 4 if dtype == _ffi.F32 else 8), dtype, external)
        self.storages.append(st)
        return T(st, 0, Cch)

    def input_nchw(self, N, Cch, H, W, focus=False, src_dtype=_ffi.F32):
        """API edge: NCHW source -> NHWC (through the Focus slicing if asked), channels zero-padded."""
        c_out = 4 * Cch if focus else Cch
        Ho, Wo = (H // 2, W // 2) if focus else (H, W)
        t = self.new(N, Ho, Wo, _round_up(c_out, self.vec))
        self.ops.append(("nchw2nhwc", dict(y=t, N=N, C=Cch, H=H, W=W, focus=int(focus), c_pad=t.C, src_dtype=src_dtype)))
        self.ext_in.append(len(self.ops) - 1)
        t.focused = bool(focus)
        return t

    def focus_stem(self, N, Cch, H, W, conv, bn=None, act=_ffi.ACT_NONE, slope=0.0, src_dtype=_ffi.F32):
        """API edge + first layer: NCHW image -> Focus slicing -> 3x3 convolution (+ BatchNorm + activation) -> NHWC.
        One fused kernel on the bf16 path when the convolution qualifies (3 image channels, 3x3 / stride 1 / pad 1,
        at most 64 output channels); otherwise the layout kernel (with the slicing folded in) followed by a
        regular convolution."""
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        fused = (self.dt == _ffi.BF16 and Cch == 3 and (k, s, p) == (3, 1, 1) and conv.groups == 1 and H % 2 == 0
                 and W % 2 == 0 and conv.out_channels <= 64 and conv.out_channels % 4 == 0)
        if not fused:
            return self.conv(self.input_nchw(N, Cch, H, W, focus=True, src_dtype=src_dtype), conv, bn, act=act, slope=slope)
        pc = pack_conv(conv, bn, None, cin_phys=16, dtype=self.dt, device=self.device)
        self.keep.append(pc)
        out = self.new(N, H // 2, W // 2, pc.Cout)
        self.ops.append(("focus_conv", dict(y=out, pc=pc, N=N, C=Cch, H=H, W=W, act=act, slope=float(slope), src_dtype=src_dtype)))
        self.ext_in.append(len(self.ops) - 1)
        return out

    # ---- operators -----------------------------------------------------------------------------
    def slice(self, t, off, C):
        """Channels [off, off + C) of `t`."""
        assert 0 <= off and off + C <= t.C
        return T(t.st, t.coff + off, C)

    def conv_siblings(self, x, pairs, act=_ffi.ACT_NONE, slope=0.0, separate=False):
        """Several (conv, bn) pairs that read the same input with the same geometry and activation, run as ONE
        convolution whose output channels are the pairs' outputs side by side (the input is read once).
        Returns (buffer, list of the pairs' channel slices in order); they share one storage.  With `separate` (two
        pairs, LDS-DMA kernel shapes only) each pair gets a storage of its own (the convolution's second destination)
        and the buffer returned is None."""
        assert all(is_plain(c) for c, _ in pairs), "grouped / dilated convolutions are not merged with their siblings"
        folded = [fold_conv(c, b, None, self.device) for c, b in pairs]
        first = pairs[0][0]
        for c, _ in pairs[1:]:
            assert (c.kernel_size, c.stride, c.padding, c.in_channels) == (first.kernel_size, first.stride, first.padding,
                                                                        first.in_channels), "sibling convs differ"
        pc = pack_folded(torch.cat([w for w, _ in folded], 0), torch.cat([b for _, b in folded], 0), cin_phys=x.C, dtype=self.dt)
        self.keep.append(pc)
        s, p = first.stride[0], first.padding[0]
        Ho = (x.H + 2 * p - pc.KH) // s + 1
        Wo = (x.W + 2 * p - pc.KW) // s + 1
        if separate:
            c0, c1 = folded[0][0].shape[0], folded[1][0].shape[0]
            bk = 32 if self.dt == _ffi.BF16 else 16
            assert len(pairs) == 2 and c0 % self.vec == 0 and c1 % self.vec == 0 and c0 + c1 > 32 and x.C % bk == 0, \
                "separate sibling outputs need the LDS-DMA convolution"
            y, y2 = self.new(x.N, Ho, Wo, c0), self.new(x.N, Ho, Wo, c1)
            self.ops.append(("conv", dict(x=x, y=y, y2=y2, res=None, pc=pc, stride=s, pad=p, act=act,
                                          slope=float(slope), res_mode=_ffi.RES_NONE, post=None)))
            self._tag_up(self.ops[-1][1])
            return None, [y, y2]
        out = self.new(x.N, Ho, Wo, pc.Cout)
        self.ops.append(("conv", dict(x=x, y=out, res=None, pc=pc, stride=s, pad=p, act=act, slope=float(slope),
                                      res_mode=_ffi.RES_NONE, post=None)))
        self._tag_up(self.ops[-1][1])
        parts, off = [], 0
        for w, _ in folded:
            parts.append(T(out.st, off, w.shape[0]))
            off += w.shape[0]
        return out, parts

    def conv(self, x, conv, bn=None, bn_slice=None, act=_ffi.ACT_NONE, slope=0.0, res=None,
             res_mode=_ffi.RES_NONE, out=None, out_dtype=None, external=False, post=None):
        """post = (bn, channel range or None, act, slope): a second BatchNorm + activation applied to the
        finished value (the normalisation that follows a concatenation of already-activated branches)."""
        pc = pack_conv(conv, bn, bn_slice, cin_phys=x.C, dtype=self.dt, device=self.device)
        self.keep.append(pc)
        post_rec = None
        if post is not None:
            pbn, psl, pact, pslope = post
            sl = slice(*psl) if psl is not None else slice(None)
            f32 = lambda t: t.detach().to(device=self.device, dtype=torch.float32)[sl]
            scale = f32(pbn.weight) / torch.sqrt(f32(pbn.running_var) + pbn.eps)
            shift = f32(pbn.bias) - f32(pbn.running_mean) * scale
            assert scale.numel() == pc.Cout, "post-BatchNorm slice does not match the conv's output channels"
            ps = torch.zeros(pc.cout_pad, device=self.device)
            pb = torch.zeros(pc.cout_pad, device=self.device)
            ps[:pc.Cout], pb[:pc.Cout] = scale, shift
            self.keep.append((ps, pb))
            post_rec = (ps, pb, pact, float(pslope))
        s, p, dil = conv.stride[0], conv.padding[0], pc.dilation
        if pc.groups > 1:
            assert x.C >= conv.in_channels and pc.cin_real * pc.groups == conv.in_channels
        Ho = (x.H + 2 * p - dil * (pc.KH - 1) - 1) // s + 1
        Wo = (x.W + 2 * p - dil * (pc.KW - 1) - 1) // s + 1
        if out is None:
            out = self.new(x.N, Ho, Wo, pc.Cout, dtype=out_dtype, external=external)
        assert (out.N, out.H, out.W, out.C) == (x.N, Ho, Wo, pc.Cout), "conv output shape mismatch"
        if res is not None:
            assert (res.N, res.H, res.W, res.C) == (out.N, out.H, out.W, out.C), "residual shape mismatch"
        self.ops.append(("conv", dict(x=x, y=out, res=res, pc=pc, stride=s, pad=p, act=act, slope=float(slope),
                                      res_mode=res_mode if res is not None else _ffi.RES_NONE, post=post_rec)))
        if pc.plain:
            self._tag_up(self.ops[-1][1])
        return out

    def bottleneck(self, x, cv1, cv2, act=_ffi.ACT_NONE, slope=0.0, add=True, out=None, res_mode=_ffi.RES_AFTER_ACT):
        """x + cv2(cv1(x)) (or cv2(cv1(x)) without `add`) for cv1 = (1x1 conv, bn), cv2 = (3x3 conv, bn) sharing one
        activation; with res_mode = RES_BEFORE_ACT the shortcut goes in before cv2's activation: act(x + bn2(conv2(cv1(x)))).
        One fused kernel when every channel count is 32 (or 64), or the channels go 64 -> 32 -> 64, on the bf16 path and
        `out` does not alias `x` (the intermediate map then never exists in HBM); otherwise two convolutions."""
        (c1, bn1), (c2, bn2) = cv1, cv2
        geom = lambda c: (c.kernel_size[0], c.stride[0], c.padding[0], c.groups if is_plain(c) else -1)
        Cc = x.C
        Cm = c1.out_channels                  # channels between the stages: Cc, or 32 under a 64-channel block
        fused = (self.dt == _ffi.BF16 and ((Cc in (32, 64) and Cm == Cc) or (Cc == 64 and Cm == 32)) and c2.out_channels == Cc
                 and c1.in_channels == Cc and c2.in_channels == Cm
                 and geom(c1) == (1, 1, 0, 1) and geom(c2) == (3, 1, 1, 1) and x.dtype == _ffi.BF16
                 and (out is None or out.resolve()[0] is not x.resolve()[0] or out.resolve()[1] + Cc <= x.resolve()[1]
                      or x.resolve()[1] + Cc <= out.resolve()[1]))
        if not fused:
            t = self.conv(x, c1, bn1, act=act, slope=slope)
            return self.conv(t, c2, bn2, act=act, slope=slope, res=x if add else None, res_mode=res_mode, out=out)
        pc1 = pack_conv(c1, bn1, None, cin_phys=Cc, dtype=self.dt, device=self.device)
        pc2 = pack_conv(c2, bn2, None, cin_phys=Cm, dtype=self.dt, device=self.device)
        self.keep += [pc1, pc2]
        if out is None:
            out = self.new(x.N, x.H, x.W, Cc)
        assert (out.N, out.H, out.W, out.C, out.dtype) == (x.N, x.H, x.W, Cc, _ffi.BF16), "bottleneck output shape mismatch"
        common = dict(x=x, y=out, act=act, slope=float(slope), post=None)
        self.ops.append(("bottleneck", dict(
            x=x, y=out, add=bool(add), pc=pc1, pc2=pc2,
            cv1=dict(common, res=None, pc=pc1, stride=1, pad=0, res_mode=_ffi.RES_NONE),
            cv2=dict(common, res=x if add else None, pc=pc2, stride=1, pad=1,
                     res_mode=res_mode if add else _ffi.RES_NONE))))
        return out

    def c3_tail(self, x, cv1, cv2, other, cv3, act=_ffi.ACT_NONE, slope=0.0, add=True):
        """cv3(cat(bottleneck(x), other)) as ONE kernel for a 32-channel bottleneck (cv1 1x1, cv2 3x3, optional shortcut)
        and cv3 = (1x1 conv 64 -> 64, bn) with the same activation: neither the bottleneck's output nor the concat is
        written.  Returns None when the shapes do not qualify (the caller then lowers the pieces separately)."""
        (c1, bn1), (c2, bn2), (c3, bn3) = cv1, cv2, cv3
        geom = lambda c: (c.kernel_size[0], c.stride[0], c.padding[0], c.groups if is_plain(c) else -1)
        ok = (self.dt == _ffi.BF16 and x.C == 32 and other.C == 32 and x.dtype == _ffi.BF16 and other.dtype == _ffi.BF16
              and c1.out_channels == 32 and c2.out_channels == 32 and geom(c1) == (1, 1, 0, 1) and geom(c2) == (3, 1, 1, 1)
              and geom(c3) == (1, 1, 0, 1) and c3.in_channels == 64 and c3.out_channels == 64
              and (other.N, other.H, other.W) == (x.N, x.H, x.W))
        if not ok:
            return None
        pc1 = pack_conv(c1, bn1, None, cin_phys=32, dtype=self.dt, device=self.device)
        pc2 = pack_conv(c2, bn2, None, cin_phys=32, dtype=self.dt, device=self.device)
        pc3 = pack_conv(c3, bn3, None, cin_phys=64, dtype=self.dt, device=self.device)
        self.keep += [pc1, pc2, pc3]
        out = self.new(x.N, x.H, x.W, 64)
        common = dict(act=act, slope=float(slope), post=None, res_mode=_ffi.RES_NONE, res=None)
        self.ops.append(("c3tail", dict(
            x=x, y=out, other=other, add=bool(add), pc=pc1, pc2=pc2, pc3=pc3,
            cv1=dict(common, x=x, y=out, pc=pc1, stride=1, pad=0),
            cv2=dict(common, x=x, y=out, pc=pc2, stride=1, pad=1, res=x if add else None,
                     res_mode=_ffi.RES_AFTER_ACT if add else _ffi.RES_NONE),
            cv3=dict(common, x=other, y=out, pc=pc3, stride=1, pad=0))))
        return out

    def head_decode(self, x, conv, z, row_max, level_off, na, no, stride, anchors, transposed):
        """1x1 prediction conv of one Detect level with the decode in its epilogue.  z / row_max: external storages
        [N,1,rows,no] / [N,1,rows,1] shared by the levels; returns the raw head map (external, 256-padded): fp32, or
        in the plan's compute dtype when set_head_raw_dtype("compute") is in force (the default; a `model.half()`
        reference returns half-precision x[i] as well).  z is decoded from the fp32 accumulators either way."""
        pc = pack_conv(conv, None, None, cin_phys=x.C, dtype=self.dt, device=self.device)
        if pc.cout_pad < 256:      # the kernel's weight tile is 256 rows
            wp = torch.zeros(256, pc.kp, device=self.device, dtype=pc.w.dtype)
            wp[:pc.cout_pad] = pc.w
            bp = torch.zeros(256, device=self.device)
            bp[:pc.cout_pad] = pc.b
            pc = PackedConv(wp, bp, pc.Cin, pc.Cout, pc.KH, pc.KW, pc.kp, 256, cin_real=pc.cin_real)
        self.keep.append(pc)
        out = self.new(x.N, x.H, x.W, pc.Cout, dtype=self.dt if _head_raw_compute else _ffi.F32, external=True)
        self.ops.append(("head_decode", dict(x=x, y=out, res=None, pc=pc, stride=1, pad=0, act=_ffi.ACT_NONE, slope=0.0,
                                             res_mode=_ffi.RES_NONE, post=None, z=z, row_max=row_max, level_off=level_off,
                                             na=na, no=no, det_stride=float(stride), anchors=[float(v) for v in anchors],
                                             transposed=int(bool(transposed)))))
        return out

    def _slice_op(self, kind, x, out, iarg=0):
        self.ops.append((kind, dict(x=x, y=out, iarg=iarg)))
        return out

    def maxpool2(self, x):
        return self._slice_op("maxpool2", x, self.new(x.N, x.H // 2, x.W // 2, x.C))

    def maxpool2s1(self, x):
        return self._slice_op("maxpool2s1", x, self.new(x.N, x.H, x.W, x.C))

    def spp(self, x):
        """x, pool5(x), pool9(x), pool13(x) concatenated on channels; x is redirected into slice 0."""
        out = self.new(x.N, x.H, x.W, 4 * x.C)
        x = self._place(x, out, 0)
        self._slice_op("spp", x, T(out.st, x.C, 3 * x.C))
        return out

    def upsample2x(self, x, mode):
        out = self._slice_op("upsample", x, self.new(x.N, 2 * x.H, 2 * x.W, x.C), iarg=mode)
        if mode == 0:
            out.up_of = (x, self.ops[-1][1])   # nearest: fuse_upsamples() may fold it into the 1x1 convs that read it
        return out

    def space_to_depth(self, x, mode):
        return self._slice_op("space2depth", x, self.new(x.N, x.H // 2, x.W // 2, 4 * x.C), iarg=mode)

    def _place(self, t, dst, coff):
        """Make `t` live at channels [coff, coff+C) of dst's storage: redirect if possible, else copy."""
        st = t.st
        if st.parent is None and not st.external and t.whole() and st.ld == t.C and st is not dst.st:
            st.parent, st.parent_off = dst.st, dst.coff + coff
            return t
        piece = T(dst.st, dst.coff + coff, t.C)
        self._slice_op("copy", t, piece)
        return piece

    def cat(self, ts):
        N, H, W = ts[0].N, ts[0].H, ts[0].W
        for t in ts:
            assert (t.N, t.H, t.W) == (N, H, W), "cat: spatial mismatch"
            if t.C % self.vec:
                raise NotImplementedError("cat of channel counts that are not vector multiples")
        out = self.new(N, H, W, sum(t.C for t in ts))
        off = 0
        for t in ts:
            placed = self._place(t, out, off)
            if off == 0 and placed is t and getattr(t, "up_of", None) is not None:
                out.up_head = (t.C,) + t.up_of     # the concat starts with a nearest-upsampled map (redirected, not copied)
            off += t.C
        return out

    def to_nchw(self, x, name, dst_dtype=_ffi.F32):
        """API edge: an NHWC slice returned to the caller as a fresh NCHW tensor."""
        self.ops.append(("nhwc2nchw", dict(x=x, name=name, dst_dtype=dst_dtype)))

    def output(self, name, t):
        self.outputs[name] = t

    # ---- nearest upsample + concat folded into the 1x1 convolutions that read them ----------------------------------
    def _tag_up(self, a):
        """Remember on a conv record that it could read the head of its input through the half-size source."""
        x, pc = a["x"], a["pc"]
        head = getattr(x, "up_head", None)
        bk = 32 if self.dt == _ffi.BF16 else 16
        if (head is not None and (pc.KH, pc.KW, a["stride"], a["pad"]) == (1, 1, 1, 0) and x.C % bk == 0 and head[0] % bk == 0
                and pc.Cout > 32 and x.H % 2 == 0 and x.W % 2 == 0 and x.dtype == self.dt):
            a["up_cand"] = head

    def fuse_upsamples(self):
        """For every nearest x2 upsample whose output is the head of a concat read ONLY by 1x1 convolutions: drop the
        upsample op and let those convolutions fetch the channels from the half-size map at (h/2, w/2) (the conv
        kernel's second source).  The concat buffer keeps the (now unwritten) channels; no other op may read them."""
        if not _fuse_upsample:
            return

        def reads(a):
            return [a[k] for k in ("x", "res") if isinstance(a.get(k), T)]
        if not any(kind == "conv" and a.get("up_cand") is not None and a.get("up") is None for kind, a in self.ops):
            return
        out_roots = [v.resolve()[0] for v in self.outputs.values() if isinstance(v, T)]
        done = set()
        for kind, a in list(self.ops):
            cand = a.get("up_cand") if kind == "conv" else None
            if cand is None or a.get("up") is not None or id(cand[2]) in done:
                continue
            C_up, src, up_rec = cand
            done.add(id(up_rec))
            uroot, uoff = up_rec["y"].resolve()
            readers, ok = [], not any(r is uroot for r in out_roots)
            for k2, a2 in self.ops:
                if a2 is up_rec:
                    continue
                for t in (reads(a2) + ([a2["cv1"]["x"]] if k2 == "bottleneck" else []) + ([a2["other"]] if k2 == "c3tail" else [])):
                    root, off = t.resolve()
                    if root is uroot and off < uoff + C_up and uoff < off + t.C:
                        fits = (k2 == "conv" and t is a2["x"] and off == uoff and a2.get("up_cand") is not None
                                and a2["up_cand"][2] is up_rec)
                        ok = ok and fits
                        readers.append(a2)
            if not ok or not readers:
                continue
            for a2 in readers:
                a2["up"] = (src, C_up)
            idx = next(i for i, (_, r) in enumerate(self.ops) if r is up_rec)
            del self.ops[idx]
            self.ext_in = [i - 1 if i > idx else i for i in self.ext_in]

    # ---- a 64-channel convolution followed by a 1x1 convolution: the second runs on the first one's tile --------------
    def fuse_pointwise_followers(self):
        """conv A (bf16, 64 output channels, LDS-DMA shapes, nothing else fused in) whose output is read by exactly one
        op, a 1x1 / stride 1 conv B with 33..64 output channels and the same activation: B takes over A's input and
        geometry (`pre`), A disappears, and the kernel runs B on A's finished tile in registers."""
        if not _fuse_followers or self.dt != _ffi.BF16:
            return

        def inputs(k, a):
            ts = [a[key] for key in ("x", "res", "other") if isinstance(a.get(key), T)]
            if k in ("bottleneck", "c3tail"):
                ts.append(a["cv1"]["x"])
            if a.get("up") is not None:
                ts.append(a["up"][0])
            return ts

        out_roots = [v.resolve()[0] for v in self.outputs.values() if isinstance(v, T)]
        i = 0
        while i < len(self.ops):
            ka, A = self.ops[i]
            i += 1
            if ka != "conv" or any(A.get(k) is not None for k in ("res", "post", "y2", "up", "pre")):
                continue
            pa, xa, ya = A["pc"], A["x"], A["y"]
            ra, oa = ya.resolve()
            if not (pa.plain and pa.Cout == 64 and xa.C % 32 == 0 and pa.KH * pa.KW <= 32 and xa.dtype == _ffi.BF16 and ya.dtype == _ffi.BF16
                    and not ra.external and ya.whole() and ra is ya.st and all(r is not ra for r in out_roots)):
                continue
            readers = [(k2, a2) for k2, a2 in self.ops if a2 is not A and any(t.resolve()[0] is ra for t in inputs(k2, a2))]
            if len(readers) != 1 or readers[0][0] != "conv":
                continue
            B = readers[0][1]
            pb = B["pc"]
            if not (pb.plain and B["x"].resolve() == (ra, oa) and B["x"].C == 64 and (pb.KH, pb.KW, B["stride"], B["pad"]) == (1, 1, 1, 0)
                    and 32 < pb.Cout <= 64 and pb.kp == 64 and B["y"].dtype == _ffi.BF16
                    and all(B.get(k) is None for k in ("res", "post", "up", "pre"))
                    and (B["act"], B["slope"]) == (A["act"], A["slope"])):
                continue
            B["pre"] = dict(pc=pa, x=xa, stride=A["stride"], pad=A["pad"])
            idx = i - 1
            del self.ops[idx]
            self.ext_in = [j - 1 if j > idx else j for j in self.ext_in]
            i = idx

    # ---- NCHW edge + first 3x3 convolution over the 3 image channels as one kernel --------------------------------------
    def fuse_rgb_stem(self):
        """`input_nchw` (3 channels, no Focus) read only by a 3x3 / stride 1 / pad 1 convolution with at most 64 output
        channels (bf16): both become one "rgb_conv" op that reads the NCHW image directly."""
        if self.dt != _ffi.BF16:
            return
        for kind, e in list(self.ops):       # (positions shift as ops are merged: an op is found again by identity)
            if kind != "nchw2nhwc" or e["focus"] or e["C"] != 3:
                continue
            root = e["y"].resolve()[0]
            readers = []
            for k2, a2 in self.ops:
                ts = [a2[key] for key in ("x", "res", "other") if isinstance(a2.get(key), T)]
                if k2 in ("bottleneck", "c3tail"):
                    ts.append(a2["cv1"]["x"])
                if a2 is not e and any(t.resolve()[0] is root for t in ts):
                    readers.append((k2, a2))
            if len(readers) != 1 or readers[0][0] != "conv" or any(v.resolve()[0] is root for v in self.outputs.values() if isinstance(v, T)):
                continue
            A = readers[0][1]
            pc, y = A["pc"], A["y"]
            if not (pc.plain and (pc.KH, pc.KW, A["stride"], A["pad"]) == (3, 3, 1, 1) and pc.Cout % 8 == 0 and pc.Cout <= 64 and y.dtype == _ffi.BF16
                    and all(A.get(k) is None for k in ("res", "post", "y2", "up", "pre")) and not y.resolve()[0].external):
                continue
            cin = pc.Cin                                   # channels the weights were packed for (3 padded to the vector)
            w = pc.w[:, :9 * cin].reshape(pc.cout_pad, 9, cin)[:, :, :4].reshape(pc.cout_pad, 36)
            w4 = torch.zeros(max(pc.cout_pad, 64), 64, device=w.device, dtype=w.dtype)
            w4[:pc.cout_pad, :36] = w
            b4 = torch.zeros(max(pc.cout_pad, 64), device=pc.b.device)
            b4[:pc.cout_pad] = pc.b
            npc = PackedConv(w4.contiguous(), b4, 4, pc.Cout, 3, 3, 64, w4.shape[0], cin_real=3)
            self.keep.append(npc)
            idx = next(i for i, (_, r) in enumerate(self.ops) if r is e)
            self.ops[idx] = ("rgb_conv", dict(y=y, pc=npc, N=e["N"], C=3, H=e["H"], W=e["W"], act=A["act"], slope=A["slope"],
                                              src_dtype=e["src_dtype"], pc_src=pc))
            j = next(i for i, (_, r) in enumerate(self.ops) if r is A)
            del self.ops[j]
            self.ext_in = [i - 1 if i > j else i for i in self.ext_in]

    # ---- 3-channel first layer + the 2x2 max pool behind it ---------------------------------------------------------------
    def fuse_rgb_pool(self):
        """An "rgb_conv" op whose map (even height and width) is read by exactly one op, a 2x2 / stride 2 max pool: the pool
        is taken in the convolution's epilogue and the full-size map is never written."""
        if not _fuse_rgb_pool or self.dt != _ffi.BF16:
            return
        for kind, e in list(self.ops):
            if kind != "rgb_conv" or e["H"] % 2 or e["W"] % 2 or e.get("pool"):
                continue
            root, off = e["y"].resolve()
            if root.external or not e["y"].whole() or any(v.resolve()[0] is root for v in self.outputs.values() if isinstance(v, T)):
                continue
            readers = []
            for k2, a2 in self.ops:
                ts = [a2[key] for key in ("x", "res", "other") if isinstance(a2.get(key), T)]
                if k2 in ("bottleneck", "c3tail"):
                    ts.append(a2["cv1"]["x"])
                if a2.get("up") is not None:
                    ts.append(a2["up"][0])
                if a2.get("pre") is not None:
                    ts.append(a2["pre"]["x"])
                if a2 is not e and any(t.resolve()[0] is root for t in ts):
                    readers.append((k2, a2))
            if len(readers) != 1 or readers[0][0] != "maxpool2":
                continue
            P = readers[0][1]
            if P["x"].resolve() != (root, off) or P["x"].C != e["y"].C or P["y"].dtype != _ffi.BF16:
                continue
            e["y"], e["pool"] = P["y"], True
            root.dead = True
            j = next(i for i, (_, r) in enumerate(self.ops) if r is P)
            del self.ops[j]
            self.ext_in = [i - 1 if i > j else i for i in self.ext_in]

    # ---- Focus layer + the stride-2 convolution behind it as one kernel ---------------------------------------------------
    def fuse_focus_pair(self):
        """A fused Focus layer with 32 output channels whose map is read by exactly one op, a plain 3x3 / stride 2 / pad 1
        convolution to 64 channels (on its own or carrying a 1x1 follower): both become one "focus_pair" op and the
        32-channel map in between -- the largest tensor of a YOLOv5 forward pass -- is never written.  Likewise the fused
        3-channel first layer of the Darknet-53 / CSPDarknet-53 trunks (32 output channels at full image size) and the
        stride-2 convolution behind it (no follower there)."""
        if not _fuse_focus_pair or self.dt != _ffi.BF16:
            return

        def inputs(k, a):
            if k in ("focus_conv", "rgb_conv", "nchw2nhwc"):
                return []
            if k == "nhwc2nchw":
                return [a["x"]]
            ts = [a[key] for key in ("x", "res", "other") if isinstance(a.get(key), T)]
            if k in ("bottleneck", "c3tail"):
                ts.append(a["cv1"]["x"])
            if a.get("up") is not None:
                ts.append(a["up"][0])
            if a.get("pre") is not None:
                ts = [t for t in ts if t is not a["x"]] + [a["pre"]["x"]]   # (the follower's own input no longer exists)
            return ts

        for kind, e0 in list(self.ops):      # (positions shift as ops are merged: an op is found again by identity)
            e = e0
            if kind not in ("focus_conv", "rgb_conv") or e["pc"].Cout != 32:
                continue
            rgb = kind == "rgb_conv"
            if rgb and e.get("pool"):
                continue
            root, off = e["y"].resolve()
            if root.external or not e["y"].whole() or any(v.resolve()[0] is root for v in self.outputs.values() if isinstance(v, T)):
                continue
            readers = [(k2, a2) for k2, a2 in self.ops if a2 is not e and any(t.resolve()[0] is root for t in inputs(k2, a2))]
            if len(readers) != 1 or readers[0][0] != "conv":
                continue
            B = readers[0][1]
            pre = B.get("pre")
            pa = pre["pc"] if pre else B["pc"]
            xa = pre["x"] if pre else B["x"]
            sa, pda = (pre["stride"], pre["pad"]) if pre else (B["stride"], B["pad"])
            if not (pa.plain and (pa.KH, pa.KW, sa, pda) == (3, 3, 2, 1) and pa.Cout == 64 and pa.kp == 288 and xa.resolve() == (root, off)
                    and xa.C == 32 and B["y"].dtype == _ffi.BF16 and all(B.get(k) is None for k in ("res", "post", "up"))):
                continue
            if pre is None and (B.get("y2") is not None or B["y"].C != 64):
                continue
            if pre is not None and not (B["pc"].Cout % 8 == 0 and B["pc"].Cout <= 64 and B["pc"].kp == 64):
                continue
            if rgb and e["act"] != B["act"]:
                continue
            # (rgb: the pair kernel reads the weights as the stand-alone first-layer kernel does, k = tap * 4 + channel, kp = 64)
            idx = next(i for i, (_, r) in enumerate(self.ops) if r is e0)
            self.ops[idx] = ("focus_pair", dict(focus=e, conv=B, rgb=rgb))
            j = next(i for i, (_, r) in enumerate(self.ops) if r is B)
            root.dead = True          # never written, never read: no memory behind it
            del self.ops[j]
            self.ext_in = [i - 1 if i > j else i for i in self.ext_in]

    # ---- finalize ------------------------------------------------------------------------------
    def finalize(self, replicas=1):
        return Plan(self, replicas)


class Plan:
    """Resolved launch list(s) + the buffers they run on.

    With `replicas` R > 1 the graph describes 1/R of the batch: the plan holds R copies of the launch list, each
    with its own activation arena, working on consecutive slices of the caller's batch (inputs and external
    outputs are contiguous in the batch dimension, so a slice is a pointer offset).  The copies run side by side
    (ydet_run_op_lists), which fills the GPU when single layers do not."""

    def __init__(self, g, replicas=1):
        self.device = g.device
        self.g = g
        self.R = replicas
        g.fuse_upsamples()
        g.fuse_pointwise_followers()
        g.fuse_rgb_stem()
        g.fuse_rgb_pool()
        g.fuse_focus_pair()
        n = len(g.ops)
        self.patch_in = list(g.ext_in)  # op indices whose source is a caller-supplied NCHW input
        self.patch_out = []       # (op index, field, storage) of external NHWC buffers
        self.nchw_out = []        # (op index, name, per-replica shape, dtype)
        self._ext_storages = [st for st in g.storages if st.external]
        self.op_info = [_op_info(kind, a) for kind, a in g.ops]   # per op: label, algorithmic flops and bytes
        self.arenas, self.op_lists = [], []
        for r in range(replicas):
            arena = {}
            for st in g.storages:
                root, _ = st.root()
                if root is st and not st.external and not getattr(st, "dead", False):
                    arena[st] = torch.empty(st.N, st.H, st.W, st.ld, device=g.device, dtype=_TORCH_DT[st.dtype])
            self.arenas.append(arena)
            ops = (_ffi.Op * n)()
            for i, (kind, a) in enumerate(g.ops):
                self._fill(ops[i], i, kind, a, arena, record=(r == 0))
            self.op_lists.append(ops)
        # K split of the few-tile / long-K convolutions (set_split_k): one workspace per launch list, sized for the largest
        # taker (the ops of a list run one after the other)
        self.splitk_ws = []
        if _split_k and g.device != "cpu" and torch.device(g.device).type == "cuda":
            lib = _ffi.lib()
            def ws_bytes(op):
                # (a caller-owned input / output / residual is patched in per call and still null here: the sizing query
                # only validates pointers, so a stand-in address goes in for its duration)
                d = op.conv
                saved = (d.x, d.y, d.res)
                probe = 256
                d.x, d.y = d.x or probe, d.y or probe
                if d.res_mode != _ffi.RES_NONE:
                    d.res = d.res or probe
                n = int(lib.ydet_conv_splitk_workspace_bytes(C.byref(d)))
                d.x, d.y, d.res = saved
                return n

            for ops in self.op_lists:
                need = [ws_bytes(op) if op.kind == _ffi.OP_CONV else 0 for op in ops]
                ws = torch.empty(max(need), dtype=torch.uint8, device=g.device) if max(need, default=0) > 0 else None
                self.splitk_ws.append(ws)
                for op, nb in zip(ops, need):
                    if nb > 0:
                        op.conv.splitk_ws, op.conv.splitk_ws_bytes = ws.data_ptr(), ws.numel()
        self.ops = self.op_lists[0]
        # run() writes the caller's pointers into the shared launch records and then enqueues them: one host thread at a
        # time per plan (the enqueue copies the records into kernel arguments, so the lock ends with the call)
        self._run_lock = threading.Lock()
        self._lists = (C.POINTER(_ffi.Op) * replicas)(*[C.cast(o, C.POINTER(_ffi.Op)) for o in self.op_lists])
        self._counts = (C.c_int * replicas)(*([n] * replicas))

    # ---- op records ------------------------------------------------------------------------------------
    def _ptr(self, root, arena, i, field, record):
        if root.external:
            if record:
                self.patch_out.append((i, field, root))
            return 0
        return arena[root].data_ptr()

    def _fill(self, op, i, kind, a, arena, record, with_y=True):
        if kind == "conv":
            self._fill_conv(op, i, a, arena, record)
        elif kind == "bottleneck":
            assert not a["x"].resolve()[0].external and not a["y"].resolve()[0].external
            self._fill_conv(op, i, a["cv1"], arena, record)
            self._fill_conv(op, i, a["cv2"], arena, record, d=op.conv2)
            op.conv2.Cin = a["pc"].Cout           # (the record's x is the block's input: the second stage reads the first's output)
            op.kind = _ffi.OP_BOTTLENECK
        elif kind == "c3tail":
            assert not any(a[k].resolve()[0].external for k in ("x", "y", "other"))
            self._fill_conv(op, i, a["cv1"], arena, record)
            self._fill_conv(op, i, a["cv2"], arena, record, d=op.conv2)
            self._fill_conv(op, i, a["cv3"], arena, record, d=op.conv3)
            op.conv.Cout = op.conv2.Cin = op.conv2.Cout = 32           # (their y fields point at the 64-wide output)
            op.conv3.Cin, op.conv3.x_ld, op.conv3.x_coff = 64, a["other"].resolve()[0].ld, a["other"].resolve()[1]
            op.conv2.y = 0
            op.kind = _ffi.OP_C3_TAIL
        elif kind == "head_decode":
            self._fill_conv(op, i, a, arena, record)
            op.kind = _ffi.OP_HEAD_DECODE
            hd = op.head
            zroot, _ = a["z"].resolve()
            hd.z = self._ptr(zroot, arena, i, "hz", record)
            if a["row_max"] is not None:
                rroot, _ = a["row_max"].resolve()
                hd.row_max = self._ptr(rroot, arena, i, "hr", record)
            hd.rows_per_image, hd.level_row_offset = zroot.W, a["level_off"]
            hd.na, hd.no, hd.ny, hd.nx = a["na"], a["no"], a["y"].H, a["y"].W
            hd.grid_transposed, hd.stride = a["transposed"], a["det_stride"]
            for k, v in enumerate(a["anchors"]):
                hd.anchor_wh[k] = v
        elif kind == "focus_pair":
            assert not a["conv"]["y"].resolve()[0].external
            self._fill_conv(op, i, a["conv"], arena, record, d=op.conv2, skip_x=True)
            self._fill(op, i, "rgb_conv" if a["rgb"] else "focus_conv", a["focus"], arena, record, with_y=False)
            op.kind = _ffi.OP_RGB_PAIR if a["rgb"] else _ffi.OP_FOCUS_PAIR
        elif kind in ("focus_conv", "rgb_conv"):
            op.kind = _ffi.OP_FOCUS_CONV if kind == "focus_conv" else _ffi.OP_RGB_CONV
            d, pc = op.conv, a["pc"]
            root, off = a["y"].resolve()
            d.y = self._ptr(root, arena, i, "cy", record) if with_y else 0
            d.w, d.bias = pc.w.data_ptr(), pc.b.data_ptr()
            d.N, d.H, d.W, d.Cin = a["N"], a["H"], a["W"], a["C"]
            d.Cout, d.KH, d.KW, d.stride, d.pad = pc.Cout, 3, 3, 1, 1
            d.Ho, d.Wo = (a["H"] // 2, a["W"] // 2) if kind == "focus_conv" or a.get("pool") else (a["H"], a["W"])
            d.y_ld, d.y_coff, d.kp, d.cout_pad = root.ld, off, pc.kp, pc.cout_pad
            d.act, d.slope, d.in_dtype, d.out_dtype = a["act"], a["slope"], _ffi.BF16, root.dtype
            op.iarg3 = a["src_dtype"]
        elif kind == "nchw2nhwc":
            op.kind = _ffi.OP_NCHW2NHWC
            root, off = a["y"].resolve()
            sl = op.slice
            sl.y = self._ptr(root, arena, i, "y", record)
            sl.N, sl.C, sl.H, sl.W = a["N"], a["C"], a["H"], a["W"]
            sl.y_ld, sl.y_coff, sl.dtype = root.ld, off, root.dtype
            op.iarg, op.iarg2, op.iarg3 = a["focus"], a["c_pad"], a["src_dtype"]
        elif kind == "nhwc2nchw":
            op.kind = _ffi.OP_NHWC2NCHW
            x = a["x"]
            root, off = x.resolve()
            sl = op.slice
            sl.x = self._ptr(root, arena, i, "x", record)
            sl.N, sl.C, sl.H, sl.W = x.N, x.C, x.H, x.W
            sl.x_ld, sl.x_coff, sl.dtype = root.ld, off, root.dtype
            op.iarg3 = a["dst_dtype"]
            if record:
                self.nchw_out.append((i, a["name"], (x.N, x.C, x.H, x.W), _TORCH_DT[a["dst_dtype"]]))
        else:
            op.kind = {"maxpool2": _ffi.OP_MAXPOOL2, "maxpool2s1": _ffi.OP_MAXPOOL2S1, "spp": _ffi.OP_SPP,
                       "upsample": _ffi.OP_UPSAMPLE, "space2depth": _ffi.OP_SPACE2DEPTH,
                       "copy": _ffi.OP_COPY}[kind]
            op.iarg = a["iarg"]
            x, y = a["x"], a["y"]
            rx, ox = x.resolve()
            ry, oy = y.resolve()
            assert rx.dtype == ry.dtype
            sl = op.slice
            sl.x, sl.y = self._ptr(rx, arena, i, "x", record), self._ptr(ry, arena, i, "y", record)
            sl.N, sl.H, sl.W, sl.C = x.N, x.H, x.W, x.C
            sl.x_ld, sl.x_coff, sl.y_ld, sl.y_coff, sl.dtype = rx.ld, ox, ry.ld, oy, rx.dtype

    def _fill_conv(self, op, i, a, arena, record, d=None, skip_x=False):
        op.kind = _ffi.OP_CONV
        d = op.conv if d is None else d
        x, y, res, pc = a["x"], a["y"], a["res"], a["pc"]
        rx, ox = x.resolve()
        ry, oy = y.resolve()
        d.x = 0 if skip_x else self._ptr(rx, arena, i, "cx", record)
        d.y = self._ptr(ry, arena, i, "cy", record)
        d.w, d.bias = pc.w.data_ptr(), pc.b.data_ptr()
        d.N, d.H, d.W, d.Cin = x.N, x.H, x.W, x.C
        d.Cout, d.KH, d.KW, d.stride, d.pad = pc.Cout, pc.KH, pc.KW, a["stride"], a["pad"]
        d.Ho, d.Wo = y.H, y.W
        d.x_ld, d.x_coff, d.y_ld, d.y_coff = rx.ld, ox, ry.ld, oy
        if res is not None:
            rr, orr = res.resolve()
            assert rr.dtype == rx.dtype
            d.res, d.res_ld, d.res_coff = self._ptr(rr, arena, i, "cres", record), rr.ld, orr
        d.kp, d.cout_pad = pc.kp, pc.cout_pad
        d.groups, d.dilation, d.cin_group = pc.groups, pc.dilation, (pc.cin_real if pc.groups > 1 else 0)
        d.act, d.slope, d.res_mode = a["act"], a["slope"], a["res_mode"]
        d.in_dtype, d.out_dtype = rx.dtype, ry.dtype
        if a.get("y2") is not None:
            r2, o2 = a["y2"].resolve()
            assert r2.dtype == ry.dtype and not r2.external
            d.y2, d.y2_ld, d.y2_coff, d.y_split = arena[r2].data_ptr(), r2.ld, o2, y.C
        if a.get("pre") is not None:     # this record is the 1x1 follower: the launch is the convolution in front of it
            pre = a["pre"]
            p1, x1 = pre["pc"], pre["x"]
            r1, o1 = x1.resolve()
            d.x = 0 if skip_x else self._ptr(r1, arena, i, "cx", record)
            d.w, d.bias = p1.w.data_ptr(), p1.b.data_ptr()
            d.H, d.W, d.Cin, d.x_ld, d.x_coff = x1.H, x1.W, x1.C, r1.ld, o1
            d.Cout, d.KH, d.KW, d.stride, d.pad = p1.Cout, p1.KH, p1.KW, pre["stride"], pre["pad"]
            d.kp, d.cout_pad, d.in_dtype = p1.kp, p1.cout_pad, r1.dtype
            d.pw_w, d.pw_bias, d.pw_cout, d.pw_kp = pc.w.data_ptr(), pc.b.data_ptr(), pc.Cout, pc.kp
            d.cout_store = 0
        if a.get("up") is not None:
            src, C_up = a["up"]
            ru, ou = src.resolve()
            assert ru.dtype == rx.dtype and not ru.external and (src.H * 2, src.W * 2, src.C) == (x.H, x.W, C_up)
            d.x_up, d.x_up_ld, d.x_up_coff, d.x_up_C = arena[ru].data_ptr(), ru.ld, ou, C_up
        if a.get("post") is not None:
            ps, pb, pact, pslope = a["post"]
            d.post_scale, d.post_shift, d.post_act, d.post_slope = ps.data_ptr(), pb.data_ptr(), pact, pslope
        # a stand-alone output buffer whose channel count is not a vector multiple (75 / 125 / 255-channel heads)
        # is stored with its zero pad channels, so that every store is a whole 16-byte piece
        vec = 4 if ry.dtype == _ffi.F32 else 8
        if pc.plain and y.st is ry and y.coff == 0 and pc.Cout % vec and _round_up(pc.Cout, vec) <= min(ry.ld, pc.cout_pad):
            d.cout_store = _round_up(pc.Cout, vec)

    # ---- execution -------------------------------------------------------------------------------------
    def run(self, *xs, timer=None, first_mark=0):
        """xs: the NCHW CUDA inputs (fp32 or bf16; other dtypes are converted to fp32), in the order
        they were declared.  Returns {name: tensor} of the plan's outputs.  `timer` (a ydet_timer handle, plans with
        one replica only): record an event before the first and after every op into marks first_mark.."""
        with self._run_lock:
            return self._run_locked(xs, timer, first_mark)

    def _run_locked(self, xs, timer, first_mark):
        lib = _ffi.lib()
        R = self.R
        assert len(xs) == len(self.patch_in), f"plan takes {len(self.patch_in)} inputs"
        assert timer is None or R == 1, "per-op timing needs an unsplit plan"
        outs = {}
        held = []
        ext = {st: torch.empty(st.N * R, st.H, st.W, st.ld, device=self.device, dtype=_TORCH_DT[st.dtype])
               for st in self._ext_storages}
        for i, x in zip(self.patch_in, xs):
            op0 = self.op_lists[0][i]
            focus = op0.kind in (_ffi.OP_FOCUS_CONV, _ffi.OP_RGB_CONV, _ffi.OP_FOCUS_PAIR, _ffi.OP_RGB_PAIR)   # first layers that read the image themselves
            if x.dtype == torch.uint8:
                # raw pixel values: the fused Focus layer scales them by 1/255 while loading (val.py:68-69 folded in);
                # any other first layer gets the scaled fp32 image
                if op0.kind not in (_ffi.OP_FOCUS_CONV, _ffi.OP_FOCUS_PAIR):
                    x = x.float() / 255.0
            elif x.dtype not in (torch.float32, torch.bfloat16):
                x = x.float()
            x = x.contiguous()
            held.append(x)
            per = (op0.conv.N, op0.conv.Cin, op0.conv.H, op0.conv.W) if focus else (op0.slice.N, op0.slice.C, op0.slice.H, op0.slice.W)
            want = (per[0] * R,) + per[1:]
            assert tuple(x.shape) == want, f"plan built for {want}, got {tuple(x.shape)}"
            step = per[0] * per[1] * per[2] * per[3] * x.element_size()
            for r in range(R):
                op = self.op_lists[r][i]
                if focus:
                    op.conv.x = x.data_ptr() + r * step
                else:
                    op.slice.x = x.data_ptr() + r * step
                op.iarg3 = {torch.bfloat16: _ffi.BF16, torch.uint8: _ffi.U8}.get(x.dtype, _ffi.F32)
        for i, field, root in self.patch_out:
            t = ext[root]
            step = root.N * root.H * root.W * root.ld * t.element_size()
            for r in range(R):
                ptr = t.data_ptr() + r * step
                op = self.op_lists[r][i]
                if field == "x":
                    op.slice.x = ptr
                elif field == "y":
                    op.slice.y = ptr
                elif field == "cx":
                    op.conv.x = ptr
                elif field == "cy":
                    op.conv.y = ptr
                elif field == "hz":
                    op.head.z = ptr
                elif field == "hr":
                    op.head.row_max = ptr
                else:
                    op.conv.res = ptr
        for i, name, shape, dt in self.nchw_out:
            t = torch.empty((shape[0] * R,) + shape[1:], device=self.device, dtype=dt)
            step = shape[0] * shape[1] * shape[2] * shape[3] * t.element_size()
            for r in range(R):
                self.op_lists[r][i].slice.y = t.data_ptr() + r * step
            outs[name] = t
        with torch.cuda.device(self.device):
            if timer is not None:
                _ffi.check(lib.ydet_run_ops_timed(self.ops, len(self.ops), _ffi.current_stream(), timer, first_mark),
                           "ydet_run_ops_timed")
            elif R == 1:
                _ffi.check(lib.ydet_run_ops(self.ops, len(self.ops), _ffi.current_stream()), "ydet_run_ops")
            else:
                _ffi.check(lib.ydet_run_op_lists(self._lists, self._counts, R, _ffi.current_stream()), "ydet_run_op_lists")
        # Every tensor handed back is the caller's: external storages are allocated per call above, and an output that
        # lives in the plan's activation arena (which the next run on this plan overwrites) is copied out.
        for name, t in self.g.outputs.items():
            root, off = t.resolve()
            if root.external:
                outs[name] = ext[root][..., off:off + t.C]
            elif R == 1:
                outs[name] = self.arenas[0][root][..., off:off + t.C].clone()
            else:
                outs[name] = torch.cat([ar[root][..., off:off + t.C] for ar in self.arenas], 0)
        return outs


_batch_split = 1
_fuse_upsample = True
_fuse_followers = True
_fuse_focus_pair = True
_fuse_rgb_pool = True
_head_raw_compute = True
_split_k = False


def set_focus_pair_fusion(flag):
    """Whether plans run the Focus layer and the stride-2 convolution behind it as one kernel (default on).  Off: two
    launches with the 32-channel half-size map written and read back (ablation / comparison)."""
    global _fuse_focus_pair
    _fuse_focus_pair = bool(flag)


def set_rgb_pool_fusion(flag):
    """Whether plans take a 2x2 / stride 2 max pool that directly follows the 3-channel first layer in that layer's
    epilogue (default on).  Off: two launches (ablation / comparison)."""
    global _fuse_rgb_pool
    _fuse_rgb_pool = bool(flag)


def set_split_k(flag):
    """Whether plans spread the K of few-tile / long-K convolutions (single images, the deepest layers) over several
    workgroups (default off).  On: several times faster for such launches; the order of the fp32 additions then depends
    on the launch shape, so an image's result may differ in the last bits between batch sizes -- off, every image of a
    batch equals its own single-image run bit for bit."""
    global _split_k
    _split_k = bool(flag)


def set_head_raw_dtype(mode):
    """Dtype of the raw prediction maps a fused Detect level returns next to the decoded rows: "compute" (default; bf16
    plans return bf16 maps, half the bytes of the largest stores of the forward pass) or "fp32"."""
    global _head_raw_compute
    assert mode in ("compute", "fp32")
    _head_raw_compute = mode == "compute"


def set_follower_fusion(flag):
    """Whether plans run a 1x1 convolution that is the only reader of a 64-channel convolution inside that
    convolution's kernel (default on).  Off: two launches with the map in between written and read back."""
    global _fuse_followers
    _fuse_followers = bool(flag)



def set_upsample_fusion(flag):
    """Whether plans fold `nearest x2 upsample -> concat -> 1x1 conv` into the convolution (default on).  Off: the
    upsampled map is written and read back as its own op (ablation / comparison)."""
    global _fuse_upsample
    _fuse_upsample = bool(flag)



def set_batch_split(n):
    """How many side-by-side launch lists a forward pass is split into along the batch (1 = off, the default).
    A split is only applied to batches of at least 8 images per part.  Two half-batch lists on two streams ran the
    YOLOv5s 640x640 batch-64 conv stack about 8 % faster than one list (their kernels fill each other's partial
    waves of workgroups); per-kernel timings are then no longer those of a kernel that owns the GPU."""
    global _batch_split
    if n not in (1, 2, 4):
        raise ValueError("batch split must be 1, 2 or 4")
    _batch_split = n


def build_plan(device, compute, N, lower):
    """Plan a forward pass for a batch of N: lower(g, n) emits the network for a batch of n into graph g."""
    R = _batch_split
    while R > 1 and (N % R or N // R < 8):
        R //= 2
    g = Graph(device, compute)
    lower(g, N // R)
    return Plan(g, R)


def conv_tile(cout):
    """Output tile (channels x pixels) the conv dispatcher picks; mirrors launch_dtype() in csrc/conv.hip."""
    return (32, 256) if cout <= 32 else (64, 128) if cout <= 64 else (128, 128)


def _wide_tile_pays(M, cout, K):
    """Mirror of wide_tile_pays() in csrc/conv.hip."""
    tn = (cout + 127) // 128
    b128, b256 = (M + 127) // 128 * tn, (M + 255) // 256 * tn
    if b256 < 256:
        return False
    w128, w256 = b128 / 768.0, b256 / 512.0
    r128, r256 = math.ceil(w128) / w128, math.ceil(w256) / w256
    return r256 * (0.85 if K >= 256 else 1.0) < r128 - 0.02


def _patch_kernel_takes(N, H, W, cin, cout):
    """conv_patch_takes() / plan_patch() of csrc/conv_patch.hip for a plain 3x3 / stride 1 / pad 1 bf16 layer: whether a
    tile of whole rows fits the LDS of a CU and the launch has a tile per CU."""
    if cin not in (64, 128) or cout < 96 or W > 224 or os.environ.get("YDET_PATCH", "1")[:1] == "0":
        return False
    best = None
    for fmw in (7, 5):
        th = min(32 * fmw // W, H)
        while th >= 1:
            npad = ((th + 2) * (W + 2) + 15) // 16 * 16
            lds = 3 * 128 * 128 + 2 * 3 * 128 * 4 + cin // 8 * npad * 16
            if lds <= 160 * 1024 and (npad + 63) // 64 * (cin // 64) <= 12:
                tiles = N * ((H + th - 1) // th) * ((cout + 127) // 128)
                cost = (tiles + 255) // 256 * fmw
                if best is None or cost < best[0]:
                    best = (cost, tiles)
                break
            th -= 1
    return best is not None and best[1] >= 256


def conv_kernel_label(cin_phys, cout, kh, kw, pad, in_dt, out_dt, M=0, plain=True, stride=1, res=False, geom=None, pipe=True):
    """Name of the kernel instantiation ydet_conv2d_nhwc launches for this shape (mirrors launch_dtype() / launch_tile()
    in csrc/conv.hip, conv_pw_takes() in csrc/conv_pw.hip and conv_patch_takes() in csrc/conv_patch.hip), so that timings
    keyed by it line up with a profiler's per-symbol statistics.  geom = (N, H, W) of the input map of a layer with no
    fused upsampling, second output or 1x1 stage (the patch kernel takes whole rows of a map); pipe: the layer has no fused
    upsampling or 1x1 stage (the software-pipelined tiles have no path for them)."""
    if not plain:
        return "conv_direct"
    if (geom is not None and kh == 3 and kw == 3 and pad == 1 and stride == 1 and in_dt == _ffi.BF16 and out_dt == _ffi.BF16
            and _patch_kernel_takes(geom[0], geom[1], geom[2], cin_phys, cout)):
        return f"conv_patch<c{cin_phys}>"
    if (kh == 1 and kw == 1 and pad == 0 and stride == 1 and not res and in_dt == _ffi.BF16 and out_dt == _ffi.BF16
            and cin_phys in (128, 256) and cout > 64 and M >= 65536 and os.environ.get("YDET_PW", "1")[:1] != "0"):
        return f"conv_pw<k{cin_phys}>"
    bn, bm = conv_tile(cout)
    bk = 32 if in_dt == _ffi.BF16 else 16
    dts = f"{'bf16' if in_dt == _ffi.BF16 else 'f32'}->{'bf16' if out_dt == _ffi.BF16 else 'f32'}"
    dma = cin_phys % bk == 0 and kh * kw <= 32
    K = kh * kw * cin_phys
    tiles = (M + 127) // 128 * ((cout + 127) // 128)
    # few tiles and a long K: the 128-byte-row variants of the 128 x 128 tile (same symbol family, same label)
    k8 = (bn == 128 and in_dt == _ffi.BF16 and out_dt == _ffi.BF16 and cin_phys % 64 == 0 and kh * kw <= 32 and K >= 256
          and (tiles <= 512 or (tiles <= 1024 and K >= 2048)) and os.environ.get("YDET_CONV_K8", "1") != "0")
    rem256 = cout % 256
    tn = (cout + 255) // 256
    piped = pipe and cin_phys % 64 == 0 and os.environ.get("YDET_CONV_PIPE", "1") != "0"
    if (bn == 128 and in_dt == _ffi.BF16 and out_dt == _ffi.BF16 and cout >= 256 and (rem256 == 0 or rem256 > 128)
            and cin_phys % 32 == 0 and kh * kw <= 32 and K >= 512 and (M + (191 if piped else 255)) // (192 if piped else 256) * tn >= 192
            and os.environ.get("YDET_CONV_256", "1") != "0"):
        # (the pipelined form picks 192-pixel tiles where they fill the last round of 256 workgroups markedly better)
        r256 = ((M + 255) // 256 * tn + 255) // 256 * 256
        r192 = ((M + 191) // 192 * tn + 255) // 256 * 192
        if piped and r192 * 100 < r256 * 92:
            return f"conv_dma<256x192,{dts}>"
        return f"conv_dma<256x256,{dts}>"
    if bn == 128 and dma and out_dt == _ffi.BF16 and not k8 and _wide_tile_pays(M, cout, K):
        bm = 256
    if bn >= 64 and dma:
        return f"conv_dma<{bn}x{bm},{dts}>"
    one = kh == 1 and kw == 1 and pad == 0
    return f"conv_generic<{bn}x{bm},{'1x1' if one else 'kxk'},{dts}>"


def _op_info(kind, a):
    esz = lambda t: 2 if t.dtype == _ffi.BF16 else 4
    if kind == "head_decode":
        x, y, pc = a["x"], a["y"], a["pc"]
        flops = 2.0 * y.N * y.H * y.W * pc.Cout * pc.cin_real
        nbytes = (x.N * x.H * x.W * x.C * esz(x) + pc.Cout * x.C * esz(x) + y.N * y.H * y.W * 256 * esz(y)
                  + y.N * y.H * y.W * pc.Cout * 4)
        return dict(kind="conv", label="conv_head_decode", flops=flops, bytes=nbytes,
                    shape=(x.N, x.H, x.W, x.C, pc.Cout, 1, 1))
    if kind == "conv" and a.get("pre") is not None:
        pre, y, pc = a["pre"], a["y"], a["pc"]
        x, p1 = pre["x"], pre["pc"]
        px = y.N * y.H * y.W
        flops = 2.0 * px * (p1.Cout * p1.cin_real * p1.KH * p1.KW + pc.Cout * pc.cin_real)
        nbytes = (x.N * x.H * x.W * x.C * esz(x) + px * pc.Cout * esz(y) + p1.Cout * p1.KH * p1.KW * x.C * esz(x)
                  + pc.Cout * 64 * esz(x))
        return dict(kind="conv", label=conv_kernel_label(x.C, p1.Cout, p1.KH, p1.KW, pre["pad"], x.dtype, y.dtype, M=px),
                    flops=flops, bytes=nbytes, shape=(x.N, x.H, x.W, x.C, pc.Cout, p1.KH, pre["stride"]))
    if kind == "conv":
        x, y, pc = a["x"], a["y"], a["pc"]
        flops = 2.0 * y.N * y.H * y.W * pc.Cout * pc.cin_real * pc.KH * pc.KW
        nbytes = x.N * x.H * x.W * x.C * esz(x) + y.N * y.H * y.W * pc.Cout * esz(y) + pc.Cout * pc.KH * pc.KW * x.C * esz(x)
        if a["res"] is not None:
            nbytes += y.N * y.H * y.W * y.C * esz(x)
        if a.get("up") is not None:     # the first C_up channels come from a map of a quarter of the pixels
            nbytes -= x.N * x.H * x.W * a["up"][1] * esz(x) * 3 // 4
        whole = a.get("up") is None and a.get("y2") is None and x.C == pc.cin_real and pc.groups == 1 and pc.dilation == 1
        return dict(kind="conv", label=conv_kernel_label(x.C, pc.Cout, pc.KH, pc.KW, a["pad"], x.dtype, y.dtype, M=y.N * y.H * y.W,
                                                         plain=pc.plain, stride=a["stride"], res=a["res"] is not None,
                                                         geom=(x.N, x.H, x.W) if whole else None, pipe=a.get("up") is None),
                    flops=flops,
                    bytes=nbytes,
                    shape=(x.N, x.H, x.W, x.C, pc.Cout, pc.KH, a["stride"]))
    if kind == "c3tail":
        x, y = a["x"], a["y"]
        px = y.N * y.H * y.W
        return dict(kind="conv", label="conv_c3_tail_c32", flops=2.0 * px * (32 * 32 * 10 + 64 * 64),
                    bytes=px * (32 + 32 + 64) * 2 + (10 * 32 * 32 + 64 * 64) * 2, shape=(x.N, x.H, x.W, 64, 64, 3, 1))
    if kind == "bottleneck":
        x, y = a["x"], a["y"]
        px = y.N * y.H * y.W
        Cc, Cm = x.C, a["pc"].Cout
        return dict(kind="conv", label=f"conv_bottleneck_c{Cc}" if Cm == Cc else f"conv_resblock_c{Cc}", flops=2.0 * px * Cc * Cm * 10,
                    bytes=px * Cc * 2 * 2 + 10 * Cc * Cm * 2, shape=(x.N, x.H, x.W, Cc, Cc, 3, 1))
    if kind == "rgb_conv":
        y, pc = a["y"], a["pc"]
        return dict(kind="conv", label="conv_rgb3x3", flops=2.0 * a["N"] * a["H"] * a["W"] * pc.Cout * 27,
                    bytes=a["N"] * 3 * a["H"] * a["W"] * 4 + y.N * y.H * y.W * y.C * esz(y),
                    shape=(a["N"], a["H"], a["W"], 3, pc.Cout, 3, 1))
    if kind == "focus_pair":
        f, B = a["focus"], a["conv"]
        pre, y = B.get("pre"), B["y"]
        pa = pre["pc"] if pre else B["pc"]
        rgb = a.get("rgb", False)
        px1, px2 = (f["N"] * f["H"] * f["W"] if rgb else f["N"] * (f["H"] // 2) * (f["W"] // 2)), y.N * y.H * y.W
        cout = B["pc"].Cout
        flops = 2.0 * px1 * 32 * (1 if rgb else 4) * f["C"] * 9 + 2.0 * px2 * 64 * 32 * 9 + (2.0 * px2 * cout * 64 if pre else 0.0)
        nbytes = f["N"] * f["C"] * f["H"] * f["W"] * 4 + px2 * cout * 2 + (32 * 160 + 64 * 288 + (cout * 64 if pre else 0)) * 2
        return dict(kind="conv", label="conv_rgb_pair" if rgb else "conv_focus_pair", flops=flops, bytes=nbytes,
                    shape=(f["N"], f["H"], f["W"], f["C"], cout, 3, 2 if rgb else 4))
    if kind == "focus_conv":
        y, pc = a["y"], a["pc"]
        return dict(kind="conv", label="conv_focus3x3", flops=2.0 * y.N * y.H * y.W * pc.Cout * 4 * a["C"] * 9,
                    bytes=a["N"] * a["C"] * a["H"] * a["W"] * 4 + y.N * y.H * y.W * y.C * esz(y),
                    shape=(a["N"], a["H"], a["W"], a["C"], pc.Cout, 3, 1))
    if kind == "nchw2nhwc":
        y = a["y"]
        n = a["N"] * a["C"] * a["H"] * a["W"]
        return dict(kind=kind, label=kind, flops=0.0, bytes=n * 4 + y.N * y.H * y.W * y.C * esz(y))
    if kind == "nhwc2nchw":
        x = a["x"]
        n = x.N * x.H * x.W * x.C
        return dict(kind=kind, label=kind, flops=0.0, bytes=n * esz(x) + n * 4)
    x, y = a["x"], a["y"]
    return dict(kind=kind, label=kind, flops=0.0,
                bytes=x.N * x.H * x.W * x.C * esz(x) + y.N * y.H * y.W * y.C * esz(y))


class PlanCache:
    """Per-model cache of plans keyed by (input shape, compute dtype).  A plan holds folded / packed copies of
    the weights, so the cache also remembers a version signature of the owning module's parameters and
    buffers and drops every plan when any of them was modified in place (optimizer step, copy_, ...)."""

    def __init__(self, owner=None):
        self.plans = {}
        self.owner = owner
        self.tensors = None
        self.signature = None

    def clear(self):
        self.plans.clear()
        self.tensors = None

    def _sig(self):
        if self.tensors is None:
            self.tensors = list(self.owner.parameters()) + list(self.owner.buffers()) if self.owner is not None else []
        return sum(t._version for t in self.tensors)

    def get(self, key, build):
        sig = self._sig()
        if sig != self.signature:
            self.plans.clear()
            self.signature = sig
        # a plan owns its activation arena: forward passes issued on different streams may overlap on the GPU and must
        # not share one, so the stream is part of the key (callers that pipeline batches over two streams get two plans)
        # the planner switches (fusions on / off, dtype of the raw head maps, batch split) shape the launch list: a
        # plan built under other settings is not reused
        key = (key, _fuse_upsample, _fuse_followers, _fuse_focus_pair, _fuse_rgb_pool, _head_raw_compute, _batch_split, _split_k)
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            key = (key, torch.cuda.current_stream().cuda_stream)
        p = self.plans.get(key)
        if p is None:
            p = self.plans[key] = build()
        return p


class GraphReplay:
    """fn(*device tensors) -> device tensors, replayed from a captured HIP graph: for callers whose host thread cannot
    keep up with enqueuing a step's launches (a Python loop doing other work between steps).  Measured on MI355X with
    YOLOv3-tiny at batch 1 (forty launches, 0.24 ms of host time per eager step): 0.298 ms per replayed step against
    0.290 ms eager with an otherwise idle host -- the step is bound by the dependency gaps between its short kernels on
    the GPU, which a graph does not close, so nothing here uses it by default.

    The first call per input signature (shapes, dtypes, device) runs fn eagerly on a side stream -- plans are built, the
    kernels' one-time attribute calls made -- and then once more under capture on that same stream (plans are cached per
    stream).  Later calls copy their inputs into the captured input buffers and launch the graph on the current stream.
    The tensors returned are the graph's own output buffers: valid until the next call with the same signature.  fn must
    not synchronise with the host or read device values (the launch lists, the decode and the padded NMS entry points
    of this package do not)."""

    def __init__(self, fn, warmup=2):
        self.fn, self.warmup, self.cache = fn, warmup, {}

    def __call__(self, *xs):
        key = tuple((tuple(x.shape), x.dtype, x.device) for x in xs)
        ent = self.cache.get(key)
        if ent is None:
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream(device=xs[0].device)
            static_in = [torch.empty_like(x) for x in xs]
            for b, x in zip(static_in, xs):
                b.copy_(x)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    self.fn(*static_in)
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                out = self.fn(*static_in)
            ent = self.cache[key] = (graph, static_in, out)
        graph, static_in, out = ent
        for b, x in zip(static_in, xs):
            b.copy_(x, non_blocking=True)
        graph.replay()
        return out


_compute_override = None


def set_compute_dtype(name):
    """Force the compute dtype of the accelerated path: 'bf16', 'fp32' or None (follow the weights)."""
    global _compute_override
    if name not in (None, "bf16", "fp32"):
        raise ValueError(name)
    _compute_override = name


def compute_dtype_for(module):
    """fp32 parameters -> exact-fp32 MFMA path; half/bf16 parameters -> bf16 path."""
    if _compute_override is not None:
        return _compute_override
    p = next(module.parameters())
    return "fp32" if p.dtype == torch.float32 else "bf16"


class AcceleratedModule(torch.nn.Module):
    """Mixin for the top-level detectors: owns the plan cache and invalidates it when weights change."""

    def _plans(self):
        pc = self.__dict__.get("_ydet_plan_cache")
        if pc is None:
            pc = self.__dict__["_ydet_plan_cache"] = PlanCache(self)
        return pc

    def invalidate_plans(self):
        """Drop every cached plan (folded / packed weights, launch lists, activation arenas).  Plans are rebuilt by
        themselves when a parameter or buffer changes through an in-place torch operation or is replaced (their version
        counters are part of the cache key); writes through `.data` (`p.data.mul_(...)`) do not move those counters:
        call this after such an edit."""
        self._plans().clear()

    def _apply(self, fn, *a, **k):
        self.invalidate_plans()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate_plans()
        return super().load_state_dict(*a, **k)

    def train(self, mode=True):
        # (plans describe the eval-mode network: a change of mode drops them; `model.eval()` on a model that already is
        # in eval mode -- val.run does that on every call -- keeps them)
        if bool(mode) != self.training:
            self.invalidate_plans()
        return super().train(mode)


# ---- lowering of plain torch.nn containers / layers ---------------------------------------------
def lower_any(g, m, x):
    """Lower a module onto graph `g`: blocks with a `lower` method, nn.Sequential chains,
    MaxPool2d(2,2), ZeroPad2d((0,1,0,1)) + MaxPool2d(2,1), nn.Upsample(x2) and bare nn.Conv2d."""
    nn = torch.nn
    if hasattr(m, "lower"):
        return m.lower(g, x)
    if isinstance(m, nn.Sequential):
        mods = list(m)
        i = 0
        while i < len(mods):
            cur = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(cur, nn.ZeroPad2d) and tuple(cur.padding) == (0, 1, 0, 1) and _is_pool(nxt, 2, 1):
                x = g.maxpool2s1(x)
                i += 2
                continue
            x = lower_any(g, cur, x)
            i += 1
        return x
    if _is_pool(m, 2, 2):
        return g.maxpool2(x)
    if isinstance(m, nn.Upsample):
        from .yolov5_official.modules import lower_upsample
        return lower_upsample(g, m, x)
    if isinstance(m, nn.Conv2d):
        return g.conv(x, m)
    if isinstance(m, nn.Identity):
        return x
    raise NotImplementedError(f"{type(m).__name__} has no HIP lowering")


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def _is_pool(m, k, s):
    return (isinstance(m, torch.nn.MaxPool2d) and _pair(m.kernel_size) == (k, k) and _pair(m.stride) == (s, s)
            and _pair(m.padding) == (0, 0) and _pair(m.dilation) == (1, 1) and not m.ceil_mode)


class TorchOps:
    """Eager face of a network wiring written against (run, cat)."""
    @staticmethod
    def run(m, x):
        return m(x)

    @staticmethod
    def cat(xs):
        return torch.cat(xs, 1)


class LowerOps:
    """Graph-emitting face of the same wiring."""

    def __init__(self, g):
        self.g = g

    def run(self, m, x):
        return lower_any(self.g, m, x)

    def cat(self, xs):
        return self.g.cat(list(xs))

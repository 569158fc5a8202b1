#!/usr/bin/env python3
"""Headline benchmark: images/sec of YOLOv5s 640x640 forward + Detect decode + NMS, batch 64 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--compute bf16|fp32] [--batch 64] [--no-cpu]

One step = one batch of synthetic images (already resident in HBM) through
    Model.forward (HIP launch list + fused decode)  ->  non_max_suppression (validation regime of the
    reference's val.run: conf 0.001, iou 0.6, multi_label)  ->  async D2H of the padded detections.
N > 1: one process per GPU (torch.distributed / RCCL); images are independent, so each rank owns its own
batch (weak scaling) and the data path has no collective.  Timing: barrier + synchronize on both sides of
exactly K steps, max over ranks; rank 0 prints ONE JSON line.

roofline: the conv kernels are MFMA-bound work; hipEvents recorded between the ops of the launch list during
the timed steps give each kernel's device time, and achieved = algorithmic conv FLOPs / that time
(peak: 2.5 PFLOP/s dense bf16, 157.3 TFLOP/s fp32).  cpu_baseline: the oracle's CPU restatement of the
same path (stock torch CPU ops + numpy decode / NMS) on a bounded sample, rank 0 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}
HBM_PEAK_GBS = 8000.0
GFLOP_PER_IMAGE = 16.96   # YOLOv5s nc=80 at 640x640, conv MACs x 2 (measured on the reference's nn.Conv2d calls)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeat", type=int, default=0, help="diagnostic: time the K steps this many more times (reported as repeat_ms)")
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--compute", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-vendor", action="store_true", help="skip the vendor-GEMM reference of the dominant kernel's launches")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the other configurations (YOLOv3-tiny 416 bs1 with its CPU leg, YOLOv2-D19 416 bs32, YOLOv3-SPP "
                         "608 bs64, YOLOv4 608 bs64), which are timed after the headline region on rank 0 of a 1-GPU run")
    ap.add_argument("--per-op", default=None, help="write a per-op timing table (tsv) to this path")
    ap.add_argument("--split", type=int, default=1, choices=[1, 2, 4],
                    help="run the forward pass as this many side-by-side launch lists over slices of the batch; with a "
                         "split the per-kernel events are taken in a second, unsplit loop outside the timed region")
    ap.add_argument("--event-every", type=int, default=0,
                    help="per-op events (kernel durations for the roofline object) are recorded on every E-th timed "
                         "step; 0 (default): on the last step of the timed region.  A step with events runs alone on "
                         "the GPU (so that a kernel's interval holds that kernel only), which drains the pipeline of "
                         "batches on the other streams, and its enqueue is host-bound (an event record per kernel): "
                         "as the last step it is enqueued while the GPU still works through the steps before it and "
                         "the pipeline is about to drain anyway; anywhere else it costs one to two steps")
    ap.add_argument("--streams", type=int, default=3,
                    help="batches are issued round-robin on this many HIP streams (each with its own plan / activation "
                         "arena), so the tail of one batch's kernels and its small-grid NMS overlap the next batch; the "
                         "steps that carry per-op events run alone")
    args = ap.parse_args()

    import yolo_amd
    from yolo_amd import _ffi, distributed as ydist, engine
    from yolo_amd.synthetic import images, init_deterministic
    from yolo_amd.yolov5_official.model import Model
    from yolo_amd.yolov5_official.utils import non_max_suppression_padded

    rank, world, local = ydist.init_from_env("cuda")
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", local if local < torch.cuda.device_count() else 0)   # (fewer GPUs than ranks: rehearsal only)
    torch.cuda.set_device(dev)
    lib = _ffi.lib()

    engine.set_compute_dtype(args.compute)
    engine.set_batch_split(args.split)
    model = Model()
    init_deterministic(model, seed=0, head_convs=list(model.model[-1].m))
    model = model.to(dev).eval()
    x = images(args.batch, args.size, seed=100 + rank).to(dev)
    S = max(1, args.streams) if args.split == 1 else 1
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    pinned_det = [torch.empty(args.batch, 300, 6, pin_memory=True) for _ in range(S)]
    pinned_cnt = [torch.empty(args.batch, dtype=torch.int32, pin_memory=True) for _ in range(S)]

    plan_holder = {}

    def step(si=0, timer=None, mark=0, ev=None):
        """One batch, start to finish, on stream si (the plan cache hands every stream its own plan)."""
        with torch.cuda.stream(streams[si]):
            compute = engine.compute_dtype_for(model)
            key = (tuple(x.shape), compute, x.device.index)
            plan = model._plans().get(key, lambda: _build(model, x, compute))
            plan_holder["plan"] = plan
            outs = plan.run(x, timer=timer, first_mark=mark)
            det_head = model.model[-1]
            z, _ = det_head.finish(outs)
            if ev is not None:
                ev[1].record()
            det, cnt = non_max_suppression_padded(z, 0.001, 0.6, multi_label=True)
            if ev is not None:
                ev[2].record()
            pinned_det[si].copy_(det, non_blocking=True)
            pinned_cnt[si].copy_(cnt, non_blocking=True)
        return cnt

    def step_alone(*a, **k):
        """A step on stream 0 with nothing else on the GPU: the other streams are drained before and held after."""
        for o in streams[1:]:
            streams[0].wait_stream(o)
        r = step(0, *a, **k)
        for o in streams[1:]:
            o.wait_stream(streams[0])
        return r

    def _build(m, xx, compute):
        return engine.build_plan(xx.device, compute, xx.shape[0], lambda g, n: m._lower(g, n, *xx.shape[1:]))

    with torch.no_grad():
        for k in range(max(args.warmup, S)):
            step(k % S)
        torch.cuda.synchronize()
        n_ops = len(plan_holder["plan"].ops)
        # What an event adds to the stream, measured before the timed region: the same step alone with and without
        # its events (back-to-back events overstate it: a kernel in between hides part of an event's processing).
        ncal = 9

        def calibrate():
            cal_t = lib.ydet_timer_create(n_ops + 1)
            cal_ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            cal_m = [torch.cuda.Event(enable_timing=True) for _ in range(2 * ncal + 1)]
            with torch.cuda.stream(streams[0]):
                for i in range(ncal):
                    cal_m[2 * i].record()
                    step(0)
                    cal_m[2 * i + 1].record()
                    step(0, cal_t, 0, cal_ev)
                cal_m[2 * ncal].record()
            torch.cuda.synchronize()
            diffs = sorted(cal_m[2 * i + 1].elapsed_time(cal_m[2 * i + 2]) - cal_m[2 * i].elapsed_time(cal_m[2 * i + 1])
                           for i in range(ncal))
            lib.ydet_timer_destroy(cal_t)
            return max(diffs[ncal // 2] / (n_ops + 3), 0.0)      # median over the pairs

        # (a split plan carries no per-op events: its calibration runs with the unsplit plan after the timed region)
        event_ms = calibrate() if plan_holder["plan"].R == 1 else 0.0
        event_src = "a step alone with its events minus the same step without, per event (median of 9 pairs before the timed region)"
        if args.event_every > 0:
            sampled = [k for k in range(args.steps) if k % args.event_every == 0]   # steps that carry events
        else:
            sampled = [args.steps - 1]
        n_ev = len(sampled)
        timer = lib.ydet_timer_create(n_ev * (n_ops + 1))
        if not timer:
            raise RuntimeError("ydet_timer_create failed")
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_ev)]

        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        split = plan_holder["plan"].R > 1
        turn = 0
        for k in range(args.steps):
            if split or k not in sampled:
                step(turn % S)
                turn += 1
            else:
                j = sampled.index(k)
                step_alone(timer, j * (n_ops + 1), evs[j])
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
        repeat_ms = []
        for _ in range(args.repeat):      # diagnostic: the same K steps again (no events), to see the run-to-run spread
            torch.cuda.synchronize()
            r0 = time.perf_counter()
            for k in range(args.steps):
                step(k % S)
            torch.cuda.synchronize()
            repeat_ms.append(round((time.perf_counter() - r0) / args.steps * 1e3, 3))
        if split:   # kernel statistics: the same steps again as ONE launch list, events after every op
            engine.set_batch_split(1)
            model.invalidate_plans()
            step()
            torch.cuda.synchronize()
            event_ms = calibrate()
            for j in range(n_ev):
                step(0, timer, j * (n_ops + 1), evs[j])
            torch.cuda.synchronize()

    if world > 1:
        on_gpu = torch.distributed.get_backend() == "nccl"
        t = torch.tensor([elapsed], device=dev if on_gpu else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel device times from the events recorded during the timed steps ----------------------
    # An interval between two events holds one kernel plus the events' own cost, which is taken off (event_ms above).
    plan = plan_holder["plan"]
    per_label = {}
    buf = (ctypes.c_float * n_ops)()
    for k in range(n_ev):
        _ffi.check(lib.ydet_timer_elapsed(timer, k * (n_ops + 1), n_ops, buf), "ydet_timer_elapsed")
        for i, info in enumerate(plan.op_info):
            d = per_label.setdefault(info["label"], dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
            d["ms"] += max(buf[i] - event_ms, 1e-4); d["launches"] += 1; d["flops"] += info["flops"]; d["bytes"] += info["bytes"]
    if args.per_op and rank == 0:
        per_op = [0.0] * n_ops
        for k in range(n_ev):
            _ffi.check(lib.ydet_timer_elapsed(timer, k * (n_ops + 1), n_ops, buf), "ydet_timer_elapsed")
            for i in range(n_ops):
                per_op[i] += max(buf[i] - event_ms, 1e-4) / n_ev
        with open(args.per_op, "w") as fh:
            fh.write("op\tlabel\tshape(N,H,W,Cin,Cout,k,s)\tus\tTFLOP/s\tGB/s\n")
            for i, info in enumerate(plan.op_info):
                us = per_op[i] * 1e3
                fh.write(f"{i}\t{info['label']}\t{info.get('shape', '')}\t{us:.1f}\t{info['flops'] / (us * 1e-6) / 1e12:.1f}\t"
                         f"{info['bytes'] / (us * 1e-6) / 1e9:.0f}\n")
    lib.ydet_timer_destroy(timer)
    nms_ms = sum(e[1].elapsed_time(e[2]) for e in evs) / n_ev
    conv = {k: v for k, v in per_label.items() if k.startswith("conv")}
    conv_ms = sum(v["ms"] for v in conv.values())
    conv_flops = sum(v["flops"] for v in conv.values())
    dom_label, dom = max(conv.items(), key=lambda kv: kv[1]["ms"])
    peak = PEAK_TFLOPS[args.compute]
    dom_tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    stack_tf = conv_flops / (conv_ms * 1e-3) / 1e12
    kernels = {k: dict(avg_us=round(v["ms"] / v["launches"] * 1e3, 2), launches_per_step=v["launches"] // n_ev,
                       tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                       gbs=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1))
               for k, v in sorted(per_label.items(), key=lambda kv: -kv[1]["ms"])}

    n_boxes = 3 * (args.size // 8) ** 2 + 3 * (args.size // 16) ** 2 + 3 * (args.size // 32) ** 2
    nms_bytes = args.batch * n_boxes * 85 * 4 + args.batch * 300 * 6 * 4

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the figure comes
    # from the committed summary of the separate rocprofv3 --pmc passes over this same command (tools/profile_round.sh)
    traffic, traffic_src = None, None
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as fh:
            tj = json.load(fh)
        if tj.get("batch") == args.batch and tj.get("size") == args.size and dom_label in tj.get("kernels", {}):
            traffic = tj["kernels"][dom_label]["hbm_bytes_per_launch"]
            traffic_src = tj.get("source")

    # what the NMS kernels actually move when the decode hands them its row maxima: the maxima, the rows that own a
    # candidate above the threshold bin (85 floats each, at most max_nms of them and their neighbours in the bin), the
    # candidate lists and the padded output
    nms_moved = args.batch * (n_boxes * 4 + 2 * 4096 * 8 + 300 * 85 * 4 + 300 * (16 + 8) + 300 * 6 * 4)
    # MFMA busy cycles of the dominant kernel from the committed counter passes (tools/profile_mfma.sh), if present
    mfma_busy, mfma_src = None, None
    mpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_mfma.json")
    if os.path.exists(mpath):
        with open(mpath) as fh:
            mj = json.load(fh)
        if dom_label in mj.get("kernels", {}):
            mfma_busy = mj["kernels"][dom_label].get("mfma_busy_frac")
            mfma_src = mj.get("source")

    result = {
        "metric": "images/sec (fwd+decode+NMS) YOLOv5s 640 bs64",
        "value": round(world * args.batch * args.steps / elapsed, 2),
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        **({"repeat_ms": repeat_ms} if repeat_ms else {}),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.compute,
        "data": "synthetic",
        "config": {"workload": f"YOLOv5s (yolov5_official/model.py) {args.size}x{args.size}, batch {args.batch} per GPU, "
                               "forward + Detect decode + NMS (conf 0.001, iou 0.6, multi_label, max_det 300)",
                   "images_per_gpu": args.batch, "global_batch": world * args.batch, "parallelism": f"dp{world}", "batch_split": args.split,
                   "streams": S,
                   "weights": "seeded synthetic (yolo_amd.synthetic.init_deterministic)"},
        "roofline": {"bound": "mfma", "kernel": dom_label, "achieved": round(dom_tf, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(dom_tf / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "mfma_busy_frac_counters": mfma_busy, "mfma_counter_source": mfma_src,
                     "avg_launch_us": round(dom["ms"] / dom["launches"] * 1e3, 2),
                     "launches_per_step": dom["launches"] // n_ev,
                     "launches_measured": dom["launches"], "event_steps": n_ev,
                     "flops_per_launch": dom["flops"] / dom["launches"],
                     "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
                     "event_overhead_us": round(event_ms * 1e3, 2), "event_overhead_source": event_src},
        "conv_stack": {"tflops": round(stack_tf, 2), "mfma_util_frac": round(stack_tf / peak, 4),
                       "ms_per_step": round(conv_ms / n_ev, 3),
                       "gflop_per_image": round(conv_flops / n_ev / args.batch / 1e9, 3)},
        "decode": "in the epilogue of the three prediction convolutions (kernel conv_head_decode)",
        "nms": {"ms_per_step": round(nms_ms, 3), "gbs": round(nms_bytes / (nms_ms * 1e-3) / 1e9, 1),
                "frac_of_hbm_peak": round(nms_bytes / (nms_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "algorithmic_bytes": nms_bytes, "traffic_bytes": nms_moved,
                "note": "gbs prices the unfused operator (every prediction row read once); with the decode's row maxima "
                        "the kernels touch traffic_bytes and the time is latency (select, rank, greedy), not bandwidth"},
        "kernels": kernels,
    }

    if rank == 0 and world == 1 and not args.no_vendor:
        try:     # (a reference point only: never in the way of the result line)
            result["roofline"]["vendor_gemm"] = vendor_gemm(dev, [i for i in plan.op_info if i["label"] == dom_label],
                                                            dom["ms"] / n_ev)
        except Exception as exc:      # noqa: BLE001
            result["roofline"]["vendor_gemm"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(model, args)
    if rank == 0 and world == 1 and not args.no_configs:
        del model, x
        torch.cuda.empty_cache()
        result["configs"] = other_configs(dev, args)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


def vendor_gemm(dev, infos, ours_ms):
    """The launches of the dominant kernel as PLAIN bf16 GEMMs of the same M x N x K (M pixels, N output channels,
    K = taps x input channels; no gather, bias, activation or residual) through torch.matmul, i.e. hipBLASLt / rocBLAS on
    this box: a reference point for `achieved` besides the dense peak.  Outside the timed region."""
    import torch
    shapes = {}
    for i in infos:
        n, h, w, cin, cout, k, s = i["shape"]
        key = (n * (h // s) * (w // s), cout, k * k * cin)
        shapes[key] = shapes.get(key, 0) + 1
    tot_us, tot_flop = 0.0, 0.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for (M, N, K), cnt in shapes.items():
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        for _ in range(5):
            a @ b.t()
        ts = []
        for _ in range(5):
            e0.record()
            for _ in range(8):
                a @ b.t()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 8 * 1e3)
        us = sorted(ts)[2]
        tot_us += cnt * us
        tot_flop += cnt * 2.0 * M * N * K
        del a, b
    return {"tflops": round(tot_flop / tot_us / 1e6, 1), "us_per_step": round(tot_us, 1), "ours_us_per_step": round(ours_ms * 1e3, 1),
            "launches": sum(shapes.values()),
            "note": "the same launches as plain bf16 GEMMs (M pixels x N channels x K) through torch.matmul = the vendor "
                    "library on this box, bursts of eight, median; no gather / bias / activation / residual"}


def cpu_baseline(model, args):
    """The oracle (CPU restatement: oracle/nets.py + oracle/postproc.py) on the host cores, bounded sample."""
    import yaml
    from oracle import nets as ON
    from oracle import postproc as OP
    from yolo_amd.synthetic import images
    from yolo_amd.yolov5_official.model import DEFAULT_CFG
    with open(DEFAULT_CFG) as fh:
        cfg = yaml.safe_load(fh)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    det = model.model[-1]
    strides = [float(s) for s in det.stride]
    anchor_grid = det.anchor_grid.detach().float().cpu().view(det.nl, det.na, 2).numpy()
    cores = torch.get_num_threads()
    bs = 4
    x = images(bs, args.size, seed=100)

    def one():
        with torch.no_grad():
            heads = ON.yolov5_heads(sd, x, cfg)
        z, _ = OP.detect_decode([h.numpy() for h in heads], strides, anchor_grid, det.na, det.no)
        OP.non_max_suppression(z, 0.001, 0.6, multi_label=True)

    one()  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        one()
        n += bs
        if time.perf_counter() - t0 > args.cpu_seconds:
            break
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} images of {args.size}x{args.size} in batches of {bs}, fp32, same weights, "
                      f"forward+decode+NMS (conf 0.001, iou 0.6, multi_label), {dt:.1f} s"}


FA_MODELS = {   # tag: (module, class, config attribute of models/model_config.py)
    "v3tiny": ("yolo_amd.models.v3tiny", "YOLOv3", "yolov3_tiny_cfg"),
    "v2d19": ("yolo_amd.models.v2d19", "YOLOv2", "yolov2_d19_cfg"),
    "v3spp": ("yolo_amd.models.v3spp", "YOLOv3", "yolov3_d53_cfg"),
    "v4": ("yolo_amd.models.v4", "YOLOv4", "yolov4_cfg"),
}


def build_family_a(tag, size):
    import importlib
    from yolo_amd.models import model_config as mc
    from yolo_amd.synthetic import init_deterministic, scale_fa_heads
    modname, cls, cfg = FA_MODELS[tag]
    net = getattr(importlib.import_module(modname), cls)(size, 20, False, anchor_size=getattr(mc, cfg)["anchor_size_voc"])
    init_deterministic(net, seed=0)
    scale_fa_heads(net, tag)
    return net.eval()


def other_configs(dev, args):
    """The other benchmark configurations, each forward + decode + per-class NMS of every image on one GPU:
    C1 YOLOv3-tiny 416 batch 1 (with the CPU restatement of the same path on the host cores, all threads and one thread),
    C2 YOLOv2-D19 416 batch 32, C3 YOLOv3-SPP 608 batch 64, C4 YOLOv4 608 batch 64 (its single-GPU shard)."""
    from yolo_amd import engine
    from yolo_amd.synthetic import images
    out = {}
    for name, tag, size, batch, steps in (("C1", "v3tiny", 416, 1, 50), ("C2", "v2d19", 416, 32, 10), ("C3", "v3spp", 608, 64, 5),
                                          ("C4", "v4", 608, 64, 5)):
        net = build_family_a(tag, size).to(dev)
        x = images(batch, size, seed=1, kind="randn").to(dev)
        # a single image leaves the deepest layers a handful of tiles each: their K is split over workgroups (opt-in: the
        # order of the fp32 sums then depends on the launch shape)
        engine.set_split_k(batch == 1)
        step_fn = net.detect_batch
        with torch.no_grad():
            for _ in range(2):
                step_fn(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step_fn(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        engine.set_split_k(False)
        out[name] = {"workload": f"{tag} {size}x{size} batch {batch}, forward + decode + per-class NMS", "images_per_s": round(batch / dt, 1),
                     "ms_per_step": round(dt * 1e3, 3), "steps": steps, "dtype": args.compute, "split_k": batch == 1}
        if name == "C1" and not args.no_cpu:
            out[name]["cpu_baseline"] = cpu_baseline_c1(net, x.cpu())
        del net, x
        torch.cuda.empty_cache()
    return out


def cpu_baseline_c1(net, x):
    """BASELINE configuration C1 on the host: the oracle's YOLOv3-tiny (stock torch CPU ops) + numpy decode + postprocess
    (utils/utils.py:101-133) of the one image, timed with every core torch uses and with a single thread."""
    from oracle import nets as ON
    from oracle import postproc as OP
    sd = {k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    anchors = [np.asarray(a, np.float32).reshape(-1, 2) for a in net._level_anchors()]

    def one():
        with torch.no_grad():
            heads = ON.v3tiny_heads(sd, x)
        bb, sc = OP.fa_decode([h.numpy() for h in heads], anchors, net.num_anchors, net.num_classes, net.decode_v2, net.input_size)
        with np.errstate(all="ignore"):
            OP.postprocess(bb[0], sc[0], net.conf_thresh, net.num_classes, net.nms_thresh)

    res = {}
    all_threads = torch.get_num_threads()
    for label, nthreads in (("all_threads", all_threads), ("one_thread", 1)):
        torch.set_num_threads(nthreads)
        one()
        t0 = time.perf_counter()
        n = 0
        while n < 10 or time.perf_counter() - t0 < 4.0:
            one()
            n += 1
        dt = time.perf_counter() - t0
        res[label] = {"value": round(n / dt, 3), "unit": "images/s", "cores": nthreads, "iterations": n}
    torch.set_num_threads(all_threads)
    res["kind"] = "port"
    res["sample"] = "one 416x416 image per iteration, fp32, same weights, forward + decode + postprocess (conf 0.001, nms 0.5)"
    return res


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: images/sec of YOLOv5s 640x640 forward + Detect decode + NMS, batch 64 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--compute bf16|fp32] [--batch 64] [--no-cpu]

One step = one batch of synthetic images (already resident in HBM) through
    Model.forward (HIP launch list + fused decode)  ->  non_max_suppression (validation regime of the
    reference's val.run: conf 0.001, iou 0.6, multi_label)  ->  async D2H of the padded detections.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  Started by a launcher (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) the process is one rank; started plainly with --gpus N it launches the N
ranks itself (fresh child processes, before this process has touched a GPU), relays rank 0's JSON line and fails if
any rank fails.  Images are independent, so each rank owns its own batch (weak scaling) and the data path has no
collective; the ONE exchange of the path -- the all-gather of the per-image detections that a metric over the whole
set needs (yolov5_official/val.py:92-140, data/vocapi_evaluator.py:60-92) -- runs after the timed region on the last
step's real detections, and rank 0 computes mAP from the gathered set (`exchange` in the result line).
Timing: barrier + synchronize on both sides of exactly K plain steps, max over ranks; rank 0 prints ONE JSON line.

roofline: the conv kernels are MFMA-bound work; hipEvents recorded between the ops of the launch list on extra steps
AFTER the timed region (each alone on the GPU) give each kernel's device time, and achieved = algorithmic conv FLOPs /
that time (peak: 2.5 PFLOP/s dense bf16, 157.3 TFLOP/s fp32).  cpu_baseline: the oracle's CPU restatement of the same
path (stock torch CPU ops + numpy decode / NMS) on a bounded sample, rank 0 only.
"""
import argparse
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}
HBM_PEAK_GBS = 8000.0
GFLOP_PER_IMAGE = 16.96   # YOLOv5s nc=80 at 640x640, conv MACs x 2 (measured on the reference's nn.Conv2d calls)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (a step is about 2 ms: a hundred of them keep the ramp-up and the drain of the three streams a small share of the timed region)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeat", type=int, default=0, help="diagnostic: time the K steps this many more times (reported as repeat_ms)")
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--compute", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-vendor", action="store_true", help="skip the vendor-GEMM reference of the dominant kernel's launches")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the other configurations (the same headline workload in fp32, YOLOv3-tiny 416 bs1 with its CPU leg, "
                         "YOLOv2-D19 416 bs32, YOLOv3-SPP 608 bs64, YOLOv4 608 bs64), which are timed after the headline region "
                         "on rank 0 of a 1-GPU run")
    ap.add_argument("--per-op", default=None, help="write a per-op timing table (tsv) to this path")
    ap.add_argument("--split", type=int, default=1, choices=[1, 2, 4],
                    help="run the forward pass as this many side-by-side launch lists over slices of the batch (the per-kernel "
                         "events are taken with the unsplit plan)")
    ap.add_argument("--event-steps", type=int, default=3,
                    help="per-op events (kernel durations for the roofline object) are recorded on this many extra steps after "
                         "the timed region, each alone on the GPU (so that a kernel's interval holds that kernel only)")
    ap.add_argument("--mock-gpu", action="store_true",
                    help="control-flow rehearsal on a machine without a GPU (CPU tests): the launcher, the rendezvous, the "
                         "exchange and the result line run as ever, the step is a stand-in that makes up detections on the "
                         "host; the line says \"mock\": true and its value means nothing")
    ap.add_argument("--streams", type=int, default=3,
                    help="batches are issued round-robin on this many HIP streams (each with its own plan / activation "
                         "arena), so the tail of one batch's kernels and its small-grid NMS overlap the next batch")
    return ap.parse_args(argv)


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as fresh child processes of this one
    (which has not imported torch, let alone touched a GPU), rank 0's stdout is this process's stdout."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), YDET_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    codes = [p.wait() for p in procs]
    if any(codes):
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return 1
    return 0


def csrc_hash():
    """sha1 over the HIP sources the library is built from (the stamp a counter summary under profiles/ carries)."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "yolo_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    with open(os.path.join(ROOT, "include", "ydet.h"), "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


def committed_counters(fname, dom_label, args, field):
    """A per-launch counter figure of the dominant kernel from a committed rocprofv3 --pmc summary (PMC counters cannot be
    read from inside this process): (value, source) -- or (None, reason) when the summary is absent, lists another
    workload, or was taken on other kernel sources than the ones this run is built from."""
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        return None, f"profiles/{fname} is absent"
    with open(path) as fh:
        j = json.load(fh)
    if j.get("batch", args.batch) != args.batch or j.get("size", args.size) != args.size:
        return None, f"profiles/{fname} is for batch {j.get('batch')} size {j.get('size')}"
    if j.get("csrc_sha1") != csrc_hash():
        return None, f"profiles/{fname} was taken on kernel sources {j.get('csrc_sha1')}, this tree is {csrc_hash()}"
    k = j.get("kernels", {}).get(dom_label)
    if not k or field not in k:
        return None, f"profiles/{fname} has no {field} for {dom_label}"
    return k[field], j.get("source")


def measure_headline(args, compute, dev, rank, world, steps, warmup, streams_n, event_steps, repeat=0):
    """Warm-up, exactly `steps` timed plain steps (barrier + synchronize on both sides), then `event_steps` steps with
    per-op events, each alone on the GPU.  Returns a dict of raw measurements and the objects later legs need."""
    import torch
    from yolo_amd import _ffi, engine
    from yolo_amd.synthetic import images, init_deterministic
    from yolo_amd.yolov5_official.model import Model
    from yolo_amd.yolov5_official.utils import non_max_suppression_padded
    lib = _ffi.lib()
    engine.set_compute_dtype(compute)
    engine.set_batch_split(args.split)
    model = Model()
    init_deterministic(model, seed=0, head_convs=list(model.model[-1].m))
    model = model.to(dev).eval()
    x = images(args.batch, args.size, seed=100 + rank).to(dev)
    S = max(1, streams_n) if args.split == 1 else 1
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    pinned_det = [torch.empty(args.batch, 300, 6, pin_memory=True) for _ in range(S)]
    pinned_cnt = [torch.empty(args.batch, dtype=torch.int32, pin_memory=True) for _ in range(S)]
    holder = {}

    def _build(m, xx, cmp):
        return engine.build_plan(xx.device, cmp, xx.shape[0], lambda g, n: m._lower(g, n, *xx.shape[1:]))

    def step(si=0, timer=None, mark=0, ev=None):
        """One batch, start to finish, on stream si (the plan cache hands every stream its own plan)."""
        with torch.cuda.stream(streams[si]):
            cmp = engine.compute_dtype_for(model)
            key = (tuple(x.shape), cmp, x.device.index)
            plan = model._plans().get(key, lambda: _build(model, x, cmp))
            holder["plan"] = plan
            outs = plan.run(x, timer=timer, first_mark=mark)
            z, _ = model.model[-1].finish(outs)
            if ev is not None:
                ev[1].record()
            det, cnt = non_max_suppression_padded(z, 0.001, 0.6, multi_label=True)
            if ev is not None:
                ev[2].record()
            pinned_det[si].copy_(det, non_blocking=True)
            pinned_cnt[si].copy_(cnt, non_blocking=True)
            holder["det"], holder["cnt"] = det, cnt
        return cnt

    def step_alone(*a, **k):
        """A step on stream 0 with nothing else on the GPU: the other streams are drained before and held after."""
        for o in streams[1:]:
            streams[0].wait_stream(o)
        r = step(0, *a, **k)
        for o in streams[1:]:
            o.wait_stream(streams[0])
        return r

    with torch.no_grad():
        for k in range(max(warmup, S)):
            step(k % S)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k % S)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        elapsed = time.perf_counter() - t0
        repeat_ms = []
        for _ in range(repeat):      # diagnostic: the same K steps again, to see the run-to-run spread
            torch.cuda.synchronize()
            r0 = time.perf_counter()
            for k in range(steps):
                step(k % S)
            torch.cuda.synchronize()
            repeat_ms.append(round((time.perf_counter() - r0) / steps * 1e3, 3))

        # ---- per-kernel device times: extra steps, outside the timed region ------------------------------------------
        if holder["plan"].R > 1:      # (a split plan carries no per-op events: they are taken with the unsplit plan)
            engine.set_batch_split(1)
            model.invalidate_plans()
            step(0)
            torch.cuda.synchronize()
        plan = holder["plan"]
        n_ops = len(plan.ops)
        # What an event adds to the stream: the same step alone with and without its events (back-to-back events overstate
        # it: a kernel in between hides part of an event's processing); median over pairs.
        ncal = 9
        cal_t = lib.ydet_timer_create(n_ops + 1)
        cal_ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        cal_m = [torch.cuda.Event(enable_timing=True) for _ in range(2 * ncal + 1)]
        for o in streams[1:]:
            streams[0].wait_stream(o)
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
        event_ms = max(diffs[ncal // 2] / (n_ops + 3), 0.0)
        n_ev = max(1, event_steps)
        timer = lib.ydet_timer_create(n_ev * (n_ops + 1))
        if not timer:
            raise RuntimeError("ydet_timer_create failed")
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n_ev)]
        for j in range(n_ev):
            step_alone(timer, j * (n_ops + 1), evs[j])
        torch.cuda.synchronize()

    per_label, per_op = {}, [0.0] * n_ops
    buf = (ctypes.c_float * n_ops)()
    for k in range(n_ev):
        _ffi.check(lib.ydet_timer_elapsed(timer, k * (n_ops + 1), n_ops, buf), "ydet_timer_elapsed")
        for i, info in enumerate(plan.op_info):
            d = per_label.setdefault(info["label"], dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
            ms = max(buf[i] - event_ms, 1e-4)
            d["ms"] += ms; d["launches"] += 1; d["flops"] += info["flops"]; d["bytes"] += info["bytes"]
            per_op[i] += ms / n_ev
    lib.ydet_timer_destroy(timer)
    nms_ms = sum(e[1].elapsed_time(e[2]) for e in evs) / n_ev
    return dict(model=model, x=x, plan=plan, elapsed=elapsed, repeat_ms=repeat_ms, per_label=per_label, per_op=per_op,
                n_ev=n_ev, nms_ms=nms_ms, event_ms=event_ms, streams=S, det=holder["det"], cnt=holder["cnt"])


def mock_measurement(args, rank, steps):
    """Stand-in for measure_headline on a machine without a GPU (--mock-gpu): seeded made-up detections per rank."""
    import torch
    gen = torch.Generator().manual_seed(100 + rank)
    t0 = time.perf_counter()
    det = torch.zeros(args.batch, 300, 6)
    cnt = torch.randint(5, 40, (args.batch,), generator=gen, dtype=torch.int32)
    for i in range(args.batch):
        c = int(cnt[i])
        xy = torch.rand(c, 2, generator=gen) * (args.size - 64)
        wh = torch.rand(c, 2, generator=gen) * 60 + 4
        conf = torch.sort(torch.rand(c, generator=gen), descending=True).values
        det[i, :c] = torch.cat([xy, xy + wh, conf[:, None], torch.randint(0, 80, (c, 1), generator=gen).float()], 1)
    return dict(elapsed=max(time.perf_counter() - t0, 1e-6) * steps, det=det, cnt=cnt, streams=0, repeat_ms=[])


def kernel_summary(m, compute):
    """(dominant conv label, its record, roofline pieces, conv stack pieces, kernels table) of a measurement."""
    per_label, n_ev = m["per_label"], m["n_ev"]
    conv = {k: v for k, v in per_label.items() if k.startswith("conv")}
    conv_ms = sum(v["ms"] for v in conv.values())
    conv_flops = sum(v["flops"] for v in conv.values())
    dom_label, dom = max(conv.items(), key=lambda kv: kv[1]["ms"])
    peak = PEAK_TFLOPS[compute]
    dom_tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    stack_tf = conv_flops / (conv_ms * 1e-3) / 1e12
    kernels = {k: dict(avg_us=round(v["ms"] / v["launches"] * 1e3, 2), launches_per_step=v["launches"] // n_ev,
                       tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["flops"] else None,
                       gbs=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1))
               for k, v in sorted(per_label.items(), key=lambda kv: -kv[1]["ms"])}
    roof = {"bound": "mfma", "kernel": dom_label, "achieved": round(dom_tf, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(dom_tf / peak, 4), "avg_launch_us": round(dom["ms"] / dom["launches"] * 1e3, 2),
            "launches_per_step": dom["launches"] // n_ev, "launches_measured": dom["launches"], "event_steps": n_ev,
            "flops_per_launch": dom["flops"] / dom["launches"], "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
            "event_overhead_us": round(m["event_ms"] * 1e3, 2),
            "event_overhead_source": "a step alone with its events minus the same step without, per event (median of 9 pairs "
                                     "after the timed region)"}
    stack = {"tflops": round(stack_tf, 2), "mfma_util_frac": round(stack_tf / peak, 4), "ms_per_step": round(conv_ms / n_ev, 3)}
    return dom_label, dom, roof, stack, kernels, conv_flops


def exchange_leg(m, args, dev, rank, world):
    """The one exchange of the path, on the last step's real detections: every rank contributes its padded detections,
    counts and global image ids to ONE all_gather_into_tensor (contiguous shards of a known total: no size exchange, no
    host synchronisation before the collective); rank 0 then computes mAP over the gathered set against synthetic labels
    (each image's five most confident detections, jittered by a seeded 1.5-pixel noise -- a stand-in for a label file,
    there is no dataset on the box)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from yolo_amd import distributed as ydist, ops
    from yolo_amd.yolov5_official.utils import ap_per_class
    det, cnt = m["det"], m["cnt"]
    ids = torch.arange(rank * args.batch, (rank + 1) * args.batch, device=det.device)
    total = world * args.batch
    backend = dist.get_backend() if world > 1 else "none"
    if backend == "gloo":      # rehearsal on CPU tensors
        det, cnt, ids = det.cpu(), cnt.cpu(), ids.cpu()
    times = []
    for _ in range(3):
        if world > 1:
            dist.barrier()
        if det.is_cuda:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        g_det, g_cnt, g_ids = ydist.all_gather_detections(det, cnt, ids, total=total)
        if det.is_cuda:
            torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    out = {"backend": backend, "exchange_ms": round(min(times), 3), "exchange_ms_first": round(times[0], 3),
           "images": int(g_ids.numel()), "ranks_seen": int(torch.unique(g_ids // args.batch).numel()),
           "bytes_per_rank": int(args.batch * (300 * 6 + 2) * 4)}
    if rank != 0:
        return out
    if not args.mock_gpu:
        g_det, g_cnt = g_det.to(dev), g_cnt.to(dev)
    n = g_det.shape[0]
    gen = torch.Generator().manual_seed(7)
    k = torch.clamp(g_cnt.cpu(), max=5)
    lab_rows, start = [], [0]
    top5 = g_det[:, :5].cpu()                    # (rows are in descending confidence)
    for i in range(n):
        ki = int(k[i])
        box = top5[i, :ki, :4] + torch.randn(ki, 4, generator=gen) * 1.5
        lab_rows.append(torch.cat([top5[i, :ki, 5:6], box], 1))          # (class, x1, y1, x2, y2)
        start.append(start[-1] + ki)
    labels = torch.cat(lab_rows, 0).to(g_det.device)
    iouv = torch.linspace(0.5, 0.95, 10).to(g_det.device)
    if g_det.is_cuda:      # TP matrices of all images in one launch (ydet_process_batch)
        start_t = torch.tensor(start, device=g_det.device, dtype=torch.int32)
        correct = ops.process_batch(g_det, g_cnt, labels, start_t, iouv).cpu()
    else:                  # mock rehearsal without a GPU: the host form of the same matching
        from yolo_amd.yolov5_official.utils import process_batch
        correct = torch.zeros(n, g_det.shape[1], 10, dtype=torch.bool)
        for i in range(n):
            c = int(g_cnt[i])
            if c and start[i + 1] > start[i]:
                correct[i, :c] = process_batch(g_det[i, :c], labels[start[i]:start[i + 1]], iouv)
    conf, cls, cnt_h = g_det[..., 4].cpu(), g_det[..., 5].cpu(), g_cnt.cpu().tolist()
    stats = [(correct[i, :c], conf[i, :c], cls[i, :c], labels[start[i]:start[i + 1], 0].cpu().tolist()) for i, c in enumerate(cnt_h) if c]
    stats = [np.concatenate(s, 0) for s in zip(*stats)]
    p, r, ap, f1, _ = ap_per_class(*stats)
    out.update({"map50": round(float(ap[:, 0].mean()), 4), "map50_95": round(float(ap.mean()), 4), "labels": int(labels.shape[0]),
                "labels_source": "synthetic: each image's five most confident detections, jittered (seeded)"})
    return out


def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}")

    import torch
    import yolo_amd  # noqa: F401
    from yolo_amd import _ffi, distributed as ydist, engine

    rank, world, local = ydist.init_from_env("cuda")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s)")
    on_gloo = world > 1 and torch.distributed.get_backend() == "gloo"
    if args.mock_gpu:
        return mock_main(args, rank, world)
    if local >= torch.cuda.device_count() and not on_gloo:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local}, {torch.cuda.device_count()} visible")
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))   # (gloo rehearsal: ranks may share a GPU)
    torch.cuda.set_device(dev)
    _ffi.lib()

    m = measure_headline(args, args.compute, dev, rank, world, args.steps, args.warmup, args.streams, args.event_steps, args.repeat)
    elapsed = m["elapsed"]
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if on_gloo else dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    exchange = None
    if world > 1:
        try:      # (outside `value`: a failing collective must not cost the result line)
            exchange = exchange_leg(m, args, dev, rank, world)
        except Exception as exc:      # noqa: BLE001
            exchange = {"error": f"{type(exc).__name__}: {exc}"}

    plan, n_ev = m["plan"], m["n_ev"]
    dom_label, dom, roof, stack, kernels, conv_flops = kernel_summary(m, args.compute)
    if args.per_op and rank == 0:
        with open(args.per_op, "w") as fh:
            fh.write("op\tlabel\tshape(N,H,W,Cin,Cout,k,s)\tus\tTFLOP/s\tGB/s\n")
            for i, info in enumerate(plan.op_info):
                us = m["per_op"][i] * 1e3
                fh.write(f"{i}\t{info['label']}\t{info.get('shape', '')}\t{us:.1f}\t{info['flops'] / (us * 1e-6) / 1e12:.1f}\t"
                         f"{info['bytes'] / (us * 1e-6) / 1e9:.0f}\n")

    n_boxes = 3 * (args.size // 8) ** 2 + 3 * (args.size // 16) ** 2 + 3 * (args.size // 32) ** 2
    nms_bytes = args.batch * n_boxes * 85 * 4 + args.batch * 300 * 6 * 4
    # what the NMS kernels actually move when the decode hands them its row maxima: the maxima, the rows that own a
    # candidate above the threshold bin (85 floats each, at most max_nms of them and their neighbours in the bin), the
    # candidate lists and the padded output
    nms_moved = args.batch * (n_boxes * 4 + 2 * 4096 * 8 + 300 * 85 * 4 + 300 * (16 + 8) + 300 * 6 * 4)
    traffic, traffic_src = committed_counters("pmc_traffic.json", dom_label, args, "hbm_bytes_per_launch")
    mfma_busy, mfma_src = committed_counters("pmc_mfma.json", dom_label, args, "mfma_busy_frac")
    roof.update({"traffic": traffic, "traffic_source": traffic_src, "mfma_busy_frac_counters": mfma_busy,
                 "mfma_counter_source": mfma_src})
    stack["gflop_per_image"] = round(conv_flops / n_ev / args.batch / 1e9, 3)

    result = {
        "metric": "images/sec (fwd+decode+NMS) YOLOv5s 640 bs64",
        "value": round(world * args.batch * args.steps / elapsed, 2),
        "unit": "images/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        **({"repeat_ms": m["repeat_ms"]} if m["repeat_ms"] else {}),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": args.compute,
        "data": "synthetic",
        "config": {"workload": f"YOLOv5s (yolov5_official/model.py) {args.size}x{args.size}, batch {args.batch} per GPU, "
                               "forward + Detect decode + NMS (conf 0.001, iou 0.6, multi_label, max_det 300)",
                   "images_per_gpu": args.batch, "global_batch": world * args.batch, "parallelism": f"dp{world}", "batch_split": args.split,
                   "streams": m["streams"], "launcher": "self" if os.environ.get("YDET_BENCH_SELF_LAUNCHED") else ("external" if world > 1 else "none"),
                   "weights": "seeded synthetic (yolo_amd.synthetic.init_deterministic)"},
        "roofline": roof,
        "conv_stack": stack,
        "decode": "in the epilogue of the three prediction convolutions (kernel conv_head_decode)",
        "nms": {"ms_per_step": round(m["nms_ms"], 3), "algorithmic_bytes": nms_bytes, "traffic_bytes": nms_moved,
                "note": "latency-bound (select, rank, greedy over <= 300 candidates per image), not a bandwidth figure: with the "
                        "decode's row maxima the kernels touch traffic_bytes of the algorithmic_bytes an unfused scan would read"},
        "csrc_sha1": csrc_hash(),
        "kernels": kernels,
    }
    if exchange is not None:
        result["exchange"] = exchange

    if rank == 0 and world == 1 and not args.no_vendor:
        try:     # (a reference point only: never in the way of the result line)
            result["roofline"]["vendor_gemm"] = vendor_gemm(dev, [i for i in plan.op_info if i["label"] == dom_label],
                                                            dom["ms"] / n_ev)
        except Exception as exc:      # noqa: BLE001
            result["roofline"]["vendor_gemm"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(m["model"], args)
    if rank == 0 and world == 1 and not args.no_configs:
        m.clear()
        del plan
        torch.cuda.empty_cache()
        result["configs"] = other_configs(dev, args, rank, world)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def mock_main(args, rank, world):
    """--mock-gpu: everything around the GPU step (rank launch, rendezvous, max over ranks, the exchange, mAP over the
    gathered set, the result line) with a stand-in for the step itself."""
    import torch
    m = mock_measurement(args, rank, args.steps)
    elapsed = m["elapsed"]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    exchange = exchange_leg(m, args, torch.device("cpu"), rank, world) if world > 1 else None
    if rank == 0:
        print(json.dumps({"metric": "images/sec (fwd+decode+NMS) YOLOv5s 640 bs64", "mock": True,
                          "value": round(world * args.batch * args.steps / elapsed, 2), "unit": "images/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "scaling": "weak",
                          "config": {"launcher": "self" if os.environ.get("YDET_BENCH_SELF_LAUNCHED") else ("external" if world > 1 else "none")},
                          "exchange": exchange}), flush=True)
    if world > 1:
        torch.distributed.barrier()
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
    import torch
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
FA_GFLOP = {"v3tiny": 6.18, "v2d19": 29.36, "v3spp": 140.90, "v4": 98.69}   # conv MACs x 2 per image at the configuration's size


def build_family_a(tag, size):
    import importlib
    from yolo_amd.models import model_config as mc
    from yolo_amd.synthetic import init_deterministic, scale_fa_heads
    modname, cls, cfg = FA_MODELS[tag]
    net = getattr(importlib.import_module(modname), cls)(size, 20, False, anchor_size=getattr(mc, cfg)["anchor_size_voc"])
    init_deterministic(net, seed=0)
    scale_fa_heads(net, tag)
    return net.eval()


def other_configs(dev, args, rank, world):
    """The other benchmark configurations on one GPU: the headline workload with fp32 MFMA (YOLOv5s 640 bs64, images/s and
    the dominant kernel against the 157.3 TFLOP/s fp32 peak), and, each forward + decode + per-class NMS of every image:
    C1 YOLOv3-tiny 416 batch 1 (with the CPU restatement of the same path on the host cores, all threads and one thread),
    C2 YOLOv2-D19 416 batch 32, C3 YOLOv3-SPP 608 batch 64, C4 YOLOv4 608 batch 64 (its single-GPU shard)."""
    import torch
    from yolo_amd import engine
    from yolo_amd.synthetic import images
    out = {}
    if args.compute == "bf16":
        m = measure_headline(args, "fp32", dev, rank, world, steps=5, warmup=2, streams_n=args.streams, event_steps=1)
        dom_label, dom, roof, stack, kernels, _ = kernel_summary(m, "fp32")
        out["headline_fp32"] = {"workload": f"YOLOv5s {args.size}x{args.size} batch {args.batch}, forward + decode + NMS, fp32 MFMA "
                                            "(v_mfma_f32_16x16x4_f32), fp32 activations and weights",
                                "images_per_s": round(args.batch * 5 / m["elapsed"], 1), "ms_per_step": round(m["elapsed"] / 5 * 1e3, 3),
                                "steps": 5, "dtype": "fp32", "roofline": roof, "conv_stack": stack}
        m.clear()
        engine.set_compute_dtype(args.compute)
        torch.cuda.empty_cache()
    for name, tag, size, batch, steps in (("C1", "v3tiny", 416, 1, 50), ("C2", "v2d19", 416, 32, 12), ("C3", "v3spp", 608, 64, 6),
                                          ("C4", "v4", 608, 64, 6)):
        net = build_family_a(tag, size).to(dev)
        x = images(batch, size, seed=1, kind="randn").to(dev)
        # a single image leaves the deepest layers a handful of tiles each: their K is split over workgroups (opt-in: the
        # order of the fp32 sums then depends on the launch shape)
        engine.set_split_k(batch == 1)
        step_fn = net.detect_batch
        # Batches are issued round-robin on HIP streams like the headline's (each stream has its own plan and activation
        # arena): the per-class NMS of one batch (vector ALU work on a few workgroups) runs under the next batch's
        # convolutions.  A single image (C1) is a latency measurement: one stream.
        S = 1 if batch == 1 else max(1, args.streams)
        streams = [torch.cuda.Stream(dev) for _ in range(S)]
        with torch.no_grad():
            for k in range(2 * S):
                with torch.cuda.stream(streams[k % S]):
                    step_fn(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                with torch.cuda.stream(streams[k % S]):
                    step_fn(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        engine.set_split_k(False)
        tf = FA_GFLOP[tag] * batch / dt / 1e3
        out[name] = {"workload": f"{tag} {size}x{size} batch {batch}, forward + decode + per-class NMS", "images_per_s": round(batch / dt, 1),
                     "ms_per_step": round(dt * 1e3, 3), "steps": steps, "streams": S, "dtype": args.compute, "split_k": batch == 1,
                     "conv_tflops_whole_step": round(tf, 1), "mfma_util_frac_whole_step": round(tf / PEAK_TFLOPS[args.compute], 4)}
        if name == "C1" and not args.no_cpu:
            out[name]["cpu_baseline"] = cpu_baseline_c1(net, x.cpu())
        del net, x
        torch.cuda.empty_cache()
    return out


def cpu_baseline_c1(net, x):
    """BASELINE configuration C1 on the host: the oracle's YOLOv3-tiny (stock torch CPU ops) + numpy decode + postprocess
    (utils/utils.py:101-133) of the one image, timed with every core torch uses and with a single thread."""
    import numpy as np
    import torch
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

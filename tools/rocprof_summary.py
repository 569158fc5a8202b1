#!/usr/bin/env python3
"""Turn rocprofv3 CSV output into the summaries kept under profiles/.

    rocprof_summary.py stats   <kernel_trace.csv>                      -> per-kernel duration statistics (CSV on stdout)
    rocprof_summary.py traffic <fetch_counter.csv> <write_counter.csv> <batch> <size>  -> HBM traffic per kernel (JSON)
    rocprof_summary.py counters <counter_collection.csv> [more.csv ...]   -> per-kernel means of every counter collected,
                                                                             plus the MFMA figures derived from them (JSON)

Kernel symbols are also given the label bench.py uses for them (conv_dma<128x128,bf16->bf16>, ...), so that the two
sources can be compared line by line.  FETCH_SIZE / WRITE_SIZE are kilobytes; on gfx950 FETCH_SIZE tallies the
128-byte requests of a wide coalesced read at 64 bytes, so the read side is doubled.
"""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_hash  # noqa: E402  (the stamp bench.py compares with the tree it runs: stale summaries read as null)


def label_of(name):
    m = re.search(r"conv_pw_kernel(?:ILi|<)(\d+)", name)      # first template argument: K steps of 32 channels
    if m:
        return f"conv_pw<k{32 * int(m.group(1))}>"
    m = re.search(r"conv_patch_kernel(?:ILi|<)(\d+)", name)   # first template argument: input channels
    if m:
        return f"conv_patch<c{m.group(1)}>"
    m = re.search(r"conv_(dma|generic)_kernel", name)
    if m:
        kind = m.group(1)
        if "DF16bDF16b" in name:
            dts = "bf16->bf16"
        elif "DF16bf" in name or "_Accum" in name:   # the demangler prints <__bf16, float> as "bool _Accum"
            dts = "bf16->f32"
        else:
            dts = "f32->f32"
        nums = re.findall(r"Li(\d+)E", name) or re.findall(r"(?<![\w.])(\d+)(?=[,>])", name)
        bn, bm = nums[0], nums[1]
        if kind == "dma":
            # template arguments: <TIN, TOUT, BN, BM, WARPS_N, WARPS_M, HEAD, KCH, NST, PIPE>: HEAD is the first bool
            m5 = re.search(r"(?:Li\d+E){4}Lb([01])E", name) or re.search(r"(?:\d+, ){4}(true|false)", name)
            if m5 and m5.group(1) in ("1", "true"):      # the prediction conv with the decode epilogue
                return "conv_head_decode"
            return f"conv_dma<{bn}x{bm},{dts}>"
        one = "Lb1E" in name or re.search(r"true>", name)
        return f"conv_generic<{bn}x{bm},{'1x1' if one else 'kxk'},{dts}>"
    if "bottleneck_c32" in name and ("kernel<true" in name or "ILb1E" in name):
        return "conv_c3_tail_c32"
    for key, lab in (("focus_pair", "conv_focus_pair"), ("focus_conv3x3", "conv_focus3x3"), ("bottleneck_c32", "conv_bottleneck_c32"), ("bottleneck_c64", "conv_bottleneck_c64"), ("resblock_c64", "conv_resblock_c64"), ("decode_v5", "decode_v5"), ("v5_pick", "nms_v5_pick"),
                     ("v5_nms", "nms_v5_greedy"), ("v5_rank", "nms_v5_rank"), ("v5_rowmax", "nms_v5_rowmax"),
                     ("v5_exact", "nms_v5_exact_select"), ("spp_lds", "spp"), ("upsample2x", "upsample"),
                     ("nchw_to_nhwc", "nchw2nhwc")):
        if key in name:
            return lab
    return ""


def stats(path):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(list)
    for r in rows:
        agg[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in agg.values()) or 1
    print('"Label","Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f'"{label_of(name)}","{name}",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{100.0 * sum(v) / total:.2f},{min(v)},{max(v)}')


def traffic(fetch_csv, write_csv, batch, size):
    def per_kernel(path, counter):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        return agg
    fetch, write = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    out = {}
    for name in fetch:
        lab = label_of(name)
        if not lab:
            continue
        f, w = fetch[name], write.get(name, [0.0])
        d = out.setdefault(lab, dict(launches=0, fetch_kb_raw=0.0, write_kb=0.0))
        d["launches"] += len(f)
        d["fetch_kb_raw"] += sum(f)
        d["write_kb"] += sum(w) * (len(f) / max(len(w), 1))
    for d in out.values():
        n = d["launches"]
        d["fetch_kb_raw"] = round(d["fetch_kb_raw"] / n, 1)
        d["write_kb"] = round(d["write_kb"] / n, 1)
        d["hbm_bytes_per_launch"] = round((2.0 * d["fetch_kb_raw"] + d["write_kb"]) * 1024)
    print(json.dumps(dict(source="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `bench.py --no-cpu` (separate runs); "
                                 "kilobytes per launch, read side doubled (gfx950 tallies 128-B requests at 64 B)",
                          batch=int(batch), size=int(size), csrc_sha1=csrc_hash(), kernels=out), indent=1))


def counters(paths):
    """Per kernel label: launches and the mean of each counter per launch.  Derived, where their inputs were collected:
    mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)
                      (busy cycles summed over the SIMDs of the chip against the cycles the launch was active);
    mfma_flops      = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 (the counter ticks once per 512 FLOP of a wave's bf16 MFMA)."""
    agg = {}
    for path in paths:
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            lab = label_of(r["Kernel_Name"])
            if lab:
                per[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for lab, cs in per.items():
            d = agg.setdefault(lab, {})
            for c, v in cs.items():
                d[c] = round(sum(v) / len(v), 1)
                d["launches"] = max(d.get("launches", 0), len(v))
    for d in agg.values():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE"):
            d["mfma_busy_frac"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
        if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in d:
            d["mfma_flops"] = d["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512.0
        # SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* tick once per four shader cycles: as fractions of the SIMD-cycles
        # of the launch (waves_per_simd: average number of waves resident on a SIMD while the launch was active)
        if d.get("GRBM_GUI_ACTIVE"):
            simd_cycles = d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
            for c, key in (("SQ_ACTIVE_INST_VALU", "valu_active_frac"), ("SQ_ACTIVE_INST_LDS", "lds_inst_active_frac"),
                           ("SQ_ACTIVE_INST_VMEM", "vmem_inst_active_frac"), ("SQ_ACTIVE_INST_ANY", "any_inst_active_frac"),
                           ("SQ_WAVE_CYCLES", "waves_per_simd")):
                if c in d:
                    d[key] = round(d[c] * 4.0 / simd_cycles, 4)
    print(json.dumps(dict(source="rocprofv3 --pmc passes of `bench.py --no-cpu --streams 1` (counters only, no trace in the same "
                                 "run); means per launch; GRBM_GUI_ACTIVE is the sum over the 8 XCDs", csrc_sha1=csrc_hash(), kernels=agg),
                     indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "counters":
        counters(sys.argv[2:])
    else:
        traffic(*sys.argv[2:6])

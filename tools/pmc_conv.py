#!/usr/bin/env python3
"""Per-kernel means of the counters tools/pmc_conv.sh / tools/pmc_cmd.sh collected:
    python tools/pmc_conv.py gpurun_out/pmcconv_<tag> [kernel-name substring, default "conv"]"""
import collections
import csv
import glob
import os
import sys

FILT = sys.argv[2] if len(sys.argv) > 2 else "conv"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if FILT in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    for c, v in sorted(m.items()):
        print(f"   {c:34s} {v:16.1f}  (n={len(cs[c])})")
    g = m.get("GRBM_GUI_ACTIVE")
    if g:
        simd = g / 8.0 * 1024.0
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES",):
            if c in m:
                print(f"   -> mfma busy frac of SIMD-cycles      {m[c] / simd:.3f}")
        for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS"):
            if c in m:
                print(f"   -> {c:28s} x4 / SIMD-cycles = {m[c] * 4.0 / simd:.3f}")
        for c in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_DATA_FIFO_FULL"):
            if c in m:
                print(f"   -> {c:28s} / CU-cycles = {m[c] / (g / 8.0 * 256.0):.3f}")
    if "TCC_HIT_sum" in m:
        print(f"   -> L2 hit rate {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")

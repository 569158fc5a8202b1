#!/bin/bash
# Kernel traces (and, for the heaviest one, HBM traffic counters) of the other benchmark configurations on the GPU box
# (run through gpurun from the repository root):   tools/profile_configs.sh <tag>     e.g. r03
# Leaves gpurun_out/prof_<tag>_<model>/trace (rocprofv3 --kernel-trace --stats) for v2d19 / v3spp / v4 and
# gpurun_out/prof_<tag>_v3spp/{fetch,write} (--pmc passes of their own: counters never share a run with a trace).
set -e
tag=${1:-r03}
export TMPDIR=/tmp
for m in v2d19 v3spp v4; do
  out=gpurun_out/prof_${tag}_$m
  mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 tools/bench_configs.py --only $m --steps 5 --warmup 2 > $out/trace.log 2>&1
  tail -1 $out/trace.log | cut -c1-200
done
out=gpurun_out/prof_${tag}_v3spp
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 tools/bench_configs.py --only v3spp --steps 2 --warmup 1 > $out/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 tools/bench_configs.py --only v3spp --steps 2 --warmup 1 > $out/write.log 2>&1
ls $out/trace $out/fetch $out/write
# MFMA / issue counters of the heaviest configuration (two passes, 8 SQ counters each at most)
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/mfma_a -o a -- python3 tools/bench_configs.py --only v3spp --steps 2 --warmup 1 > $out/mfma_a.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/mfma_b -o b -- python3 tools/bench_configs.py --only v3spp --steps 2 --warmup 1 > $out/mfma_b.log 2>&1

#!/bin/bash
# Hardware counters of ONE convolution shape under a forced tile (run through gpurun from the repository root):
#   tools/pmc_conv.sh <tag> <lib.so> <tile or -> "<N,H,W,Cin,Cout,k,s,res>"
# Four rocprofv3 --pmc passes (counters never share a run with a trace; at most 8 SQ / 4 TCC / 4 TCP counters per pass),
# each a short tools/ab_conv.py run; tools/pmc_conv.py prints per-kernel means.
tag=$1; lib=$2; tile=$3; shape=$4
out=gpurun_out/pmcconv_$tag
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o c -- python3 tools/ab_conv.py --libs $lib --tiles $tile --shapes "$shape" --rounds 4 --burst 4 > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 1; }
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE && \
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE && \
run c TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE && \
run d TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE && \
python3 tools/pmc_conv.py $out

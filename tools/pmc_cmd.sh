#!/bin/bash
# Hardware counters of the kernels of ANY short GPU program (run through gpurun from the repository root):
#   tools/pmc_cmd.sh <tag> <kernel-name substring> python3 <script> <args...>
# Four rocprofv3 --pmc passes (counters never share a run with a trace; at most 8 SQ / 4 TCC / 4 TCP counters per pass);
# tools/pmc_conv.py prints per-kernel means of the kernels whose name contains the substring.
tag=$1; filt=$2; shift; shift
out=gpurun_out/pmccmd_$tag
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o c -- "${CMD[@]}" > $out/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/$name.log; exit 1; }
}
CMD=("$@")
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE && \
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE && \
run c TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE && \
run d TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE && \
python3 tools/pmc_conv.py $out "$filt"

#!/bin/bash
# MFMA / issue / LDS counters of the headline benchmark (run through gpurun from the repository root):
#   tools/profile_mfma.sh <tag>
# Two rocprofv3 --pmc passes (8 SQ counters each at most; counters never share a run with a trace), one stream so that
# kernels do not overlap.  tools/rocprof_summary.py counters turns the CSVs into profiles/<tag>_pmc_mfma.json.
set -e
tag=${1:-r02}
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $out/a -o a -- python3 bench.py --steps 4 --warmup 2 --no-cpu --no-configs --streams 1 > $out/a.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_LDS_DATA_FIFO_FULL SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/b -o b -- python3 bench.py --steps 4 --warmup 2 --no-cpu --no-configs --streams 1 > $out/b.log 2>&1
find $out -name "*counter_collection.csv" | xargs ls -la

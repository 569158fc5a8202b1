#!/bin/bash
# Summaries of the CSVs tools/profile_all.sh left under gpurun_out/ (run here, after gpurun merged them back):
#   tools/profile_collect.sh <tag>    -> profiles/<tag>_*.csv / .json, profiles/pmc_traffic.json, profiles/pmc_mfma.json
set -e
tag=${1:-r03}
p=gpurun_out/prof_$tag
python3 tools/rocprof_summary.py stats $p/trace/t_kernel_trace.csv > profiles/${tag}_kernel_stats.csv
python3 tools/rocprof_summary.py stats $p/trace1/t_kernel_trace.csv > profiles/${tag}_kernel_stats_one_stream.csv
cp $p/trace/t_kernel_stats.csv profiles/${tag}_rocprofv3_kernel_stats_raw.csv
python3 tools/rocprof_summary.py traffic $p/fetch/f_counter_collection.csv $p/write/w_counter_collection.csv 64 640 > profiles/pmc_traffic.json
cp profiles/pmc_traffic.json profiles/${tag}_pmc_traffic.json
python3 tools/rocprof_summary.py counters $(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | sort) > profiles/pmc_mfma.json
cp profiles/pmc_mfma.json profiles/${tag}_pmc_mfma.json
for m in v2d19 v3spp v4; do
  python3 tools/rocprof_summary.py stats gpurun_out/prof_${tag}_$m/trace/t_kernel_trace.csv > profiles/${tag}_${m}_kernel_stats.csv
done
python3 tools/rocprof_summary.py traffic gpurun_out/prof_${tag}_v3spp/fetch/f_counter_collection.csv gpurun_out/prof_${tag}_v3spp/write/w_counter_collection.csv 64 608 > profiles/${tag}_v3spp_pmc_traffic.json
python3 tools/rocprof_summary.py counters $(find gpurun_out/prof_${tag}_v3spp/mfma_a gpurun_out/prof_${tag}_v3spp/mfma_b -name "*counter_collection.csv" | sort) > profiles/${tag}_v3spp_pmc_mfma.json
ls -la profiles/ | grep $tag

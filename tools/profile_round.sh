#!/bin/bash
# Profiles of the headline benchmark on the GPU box (run through gpurun from the repository root):
#   tools/profile_round.sh <tag>     e.g. r01
# Leaves gpurun_out/prof_<tag>/{trace,fetch,write}/ with rocprofv3's CSVs; tools/rocprof_summary.py turns them
# into profiles/<tag>_kernel_stats.csv and profiles/pmc_traffic.json.  Three separate runs: counters never share a
# run with a trace.
set -e
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-configs > $out/trace.log 2>&1
# the same trace with one stream: no kernel overlaps another, so a kernel's average is its duration alone (what the
# per-op events of bench.py measure); with three streams the durations include the time a kernel shares the chip
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace1 -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-configs --streams 1 > $out/trace1.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-configs > $out/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-configs > $out/write.log 2>&1
tail -1 $out/trace.log | cut -c1-300
ls $out/trace $out/fetch $out/write

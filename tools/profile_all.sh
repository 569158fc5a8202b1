#!/bin/bash
# Every profile of the current tree in one go (run through gpurun from the repository root):   tools/profile_all.sh <tag>   e.g. r03
# Headline traces and HBM counters (tools/profile_round.sh), its MFMA / LDS counters (tools/profile_mfma.sh), traces of the
# other configurations and the HBM counters of the heaviest (tools/profile_configs.sh).  Raw CSVs stay under gpurun_out/;
# the summaries go to gpurun_out/profiles_<tag>/ -- copy them to profiles/ (tools/profile_collect.sh <tag> does).
set -e
tag=${1:-r03}
tools/profile_round.sh $tag > gpurun_out/profile_round_$tag.log 2>&1
tools/profile_mfma.sh $tag > gpurun_out/profile_mfma_$tag.log 2>&1
tools/profile_configs.sh $tag > gpurun_out/profile_configs_$tag.log 2>&1
echo done

#!/bin/bash
# Copies the summaries of tools/profile_round.sh runs from gpurun_out/ (scratch) into profiles/ (tracked):
#   tools/keep_profiles.sh <tag> [<lit tag>]
# traffic.json / valu.json (and traffic_lit.json / valu_lit.json) are what bench.py quotes its counter side fields from.
set -e
TAG=$1; LIT=$2
S=gpurun_out/prof_$TAG
cp $S/${TAG}_kernel_trace_summary.json $S/${TAG}_pmc_summary.json $S/${TAG}_kernel_stats.csv profiles/
cp $S/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
cp $S/traffic.json profiles/traffic.json; cp $S/valu.json profiles/valu.json
if [ -n "$LIT" ]; then
  S=gpurun_out/prof_$LIT
  cp $S/${LIT}_kernel_trace_summary.json $S/${LIT}_pmc_summary.json profiles/
  [ -f $S/${LIT}_kernel_stats.csv ] && cp $S/${LIT}_kernel_stats.csv profiles/
  cp $S/bench_under_rocprof.json profiles/${LIT}_bench_under_rocprof.json
  cp $S/traffic.json profiles/traffic_lit.json; cp $S/valu.json profiles/valu_lit.json
fi

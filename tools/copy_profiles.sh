#!/bin/bash
# tools/copy_profiles.sh rNN: copy the evidence tools/collect_profiles.sh left in gpurun_out/prof into profiles/ under the round's prefix
set -u
R=${1:?round prefix, e.g. r06}; P=gpurun_out/prof; D=profiles
cpi() { [ -f "$1" ] && cp "$1" "$2"; }
cpi $P/bench.json $D/${R}_bench.json
for c in 3 4 5; do cpi $P/bench_cfg$c.json $D/${R}_bench_cfg$c.json; done
for k in inflight isolated cfg3 cfg4 cfg5 train train_cfg3 train_cfg4 train_cfg5; do cpi $P/trace_$k/t_kernel_stats.csv $D/${R}_rocprofv3_kernel_stats_$k.csv; done
for k in traffic mfma; do
  cpi $P/pmc_$k.json $D/${R}_pmc_$k.json; cpi $P/pmc_$k.txt $D/${R}_pmc_$k.txt
  for c in 3 4 5; do cpi $P/pmc_${k}_cfg$c.json $D/${R}_pmc_${k}_cfg$c.json; done
done
ls $D | grep "^${R}_" | wc -l

#!/bin/bash
# Run on the GPU box from the repo root (gpurun -- 'bash tools/collect_profiles.sh'): writes the round's evidence into gpurun_out/prof/.
# rocprofv3 gets the python program directly after "--" (no env / bash -c hops); PMC passes are separate from the trace pass.
set -u
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/prof; mkdir -p $O
python3 bench.py --steps 30 --warmup 3 > $O/bench.json 2> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --inflight 1 > $O/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --inflight 1 --no-graph > $O/pmc_$c.log 2>&1
done
cd $R
python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv 5 $O/pmc_traffic.json > $O/pmc_traffic.txt
rm -f $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv $O/trace/t_kernel_trace.csv
ls -la $O $O/trace

#!/bin/bash
# Run on the GPU box from the repo root (gpurun -- 'bash tools/collect_profiles.sh'): writes the round's evidence into gpurun_out/prof/.
# rocprofv3 gets the python program directly after "--" (no env / bash -c hops); PMC passes are separate from the trace passes.
set -u
export TMPDIR=/tmp
# a plain shell export: under rocprofv3 the profiler's library initialises the GPU before Python starts, so bench.py's own
# os.environ.setdefault comes too late and the runs would see the runtime's default of 4 hardware queues
export GPU_MAX_HW_QUEUES=16
echo "GPU_MAX_HW_QUEUES=$GPU_MAX_HW_QUEUES"
R=$PWD; O=$R/gpurun_out/prof; mkdir -p $O
COMMIT=${1:-unknown}
PART=${2:-all}          # "main" = the cfg-2 evidence, "others" = cfg 3/4/5, "train" = the training step, "all" = main + others (separate gpurun calls fit the 20-minute limit)
if [ "$PART" = train ]; then
# the training step: kernel trace of tools/train_bench.py (2 warm-up + 2 timed + 1 event-profiled step = 5 steps)
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/trace_train -o t --output-format csv -- python3 $R/tools/train_bench.py 2 > $O/trace_train.log 2>&1
for c in 3 4 5; do   # the other BASELINE configurations' training steps
  rocprofv3 --kernel-trace --stats -d $O/trace_train_cfg$c -o t --output-format csv -- python3 $R/tools/train_bench.py 2 $c > $O/trace_train_cfg$c.log 2>&1
done
cd $R
find $O -name "*_kernel_trace.csv" -delete
exit 0
fi
if [ "$PART" != others ]; then
python3 bench.py --steps 30 --warmup 3 > $O/bench.json 2> $O/bench.err
cd /tmp
# kernel traces of the default command in BOTH modes: the timed one (3 graphs in flight) and one slice in flight
rocprofv3 --kernel-trace --stats -d $O/trace_inflight -o t --output-format csv -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --repeats 0 --headline-only > $O/trace_inflight.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/trace_isolated -o t --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --repeats 0 --inflight 1 --headline-only > $O/trace_isolated.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --repeats 0 --inflight 1 --no-graph --headline-only > $O/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE \
  -d $O/pmc_mfma -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --repeats 0 --inflight 1 --no-graph --headline-only > $O/pmc_mfma.log 2>&1
cd $R
python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE/p_counter_collection.csv $O/pmc_WRITE_SIZE/p_counter_collection.csv $O/pmc_FETCH_SIZE.log $O/pmc_traffic.json "$COMMIT" > $O/pmc_traffic.txt
python3 tools/pmc_mfma.py $O/pmc_mfma/p_counter_collection.csv $O/pmc_mfma.log $O/pmc_mfma.json "$COMMIT" > $O/pmc_mfma.txt
# (the PMC tools divide by the number of forwards the run executed: bench.py prints it as forwards_run on its line, which the .log holds)
fi
if [ "$PART" != main ]; then
# the other BASELINE configurations: one bench line (CPU baseline bounded to 16 threads, one forward) and one trace each
for c in 3 4 5; do
  python3 bench.py --config $c --steps 12 --warmup 3 --cpu-forwards 1 --cpu-threads 16 > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $O/trace_cfg$c -o t --output-format csv -- python3 $R/bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline --repeats 0 --inflight 1 --headline-only > $O/trace_cfg$c.log 2>&1
  for k in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $k -d $O/pmc_${k}_cfg$c -o p --output-format csv -- python3 $R/bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline --repeats 0 --inflight 1 --no-graph --headline-only > $O/pmc_${k}_cfg$c.log 2>&1
  done
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE \
    -d $O/pmc_mfma_cfg$c -o p --output-format csv -- python3 $R/bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline --repeats 0 --inflight 1 --no-graph --headline-only > $O/pmc_mfma_cfg$c.log 2>&1
  cd $R
  python3 tools/pmc_traffic.py $O/pmc_FETCH_SIZE_cfg$c/p_counter_collection.csv $O/pmc_WRITE_SIZE_cfg$c/p_counter_collection.csv $O/pmc_FETCH_SIZE_cfg$c.log $O/pmc_traffic_cfg$c.json "$COMMIT" $c > $O/pmc_traffic_cfg$c.txt
  python3 tools/pmc_mfma.py $O/pmc_mfma_cfg$c/p_counter_collection.csv $O/pmc_mfma_cfg$c.log $O/pmc_mfma_cfg$c.json "$COMMIT" > $O/pmc_mfma_cfg$c.txt
done
fi
find $O -name "*_counter_collection.csv" -delete; find $O -name "*_kernel_trace.csv" -delete
ls -la $O $O/trace_inflight

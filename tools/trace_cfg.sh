#!/bin/bash
# tools/trace_cfg.sh CFG [OUTDIR]: rocprofv3 kernel trace of one slice in flight for one BASELINE configuration, summarised per
# (kernel, grid) by tools/trace_by_grid.py.  Run on the GPU box from the repo root.
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
C=${1:-4}; R=$PWD; O=${2:-gpurun_out/trace_cfg$C}; case $O in /*) ;; *) O=$R/$O;; esac; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O -o t --output-format csv -- python3 $R/bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline --repeats 0 --inflight 1 --headline-only > $O/trace.log 2>&1
cd $R
T=$(find $O -name "t_kernel_trace.csv" | head -1)
python3 tools/trace_by_grid.py $T > $O/by_grid.txt
find $O -name "t_kernel_trace.csv" -delete
head -40 $O/by_grid.txt

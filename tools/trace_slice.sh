#!/bin/bash
# tools/trace_slice.sh CFG BRANCHES [OUT]: rocprofv3 kernel trace of single-slice eager forwards, summarised by tools/trace_slice.py (run on the GPU box from the repo root)
set -u
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
C=${1:-2}; B=${2:-1}; R=$PWD; O=${3:-gpurun_out/slice_cfg${C}_b$B}; case $O in /*) ;; *) O=$R/$O;; esac; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python3 $R/tools/trace_slice.py run $C $B 3 > $O/run.log 2>&1
cd $R
T=$(find $O -name "t_kernel_trace.csv" | head -1)
python3 tools/trace_slice.py show $T > $O/timeline.txt
find $O -name "t_kernel_trace.csv" -delete
head -8 $O/timeline.txt

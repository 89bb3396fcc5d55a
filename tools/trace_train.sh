#!/bin/bash
# tools/trace_train.sh [config 2|3|4|5]: kernel trace of one configuration's training step (tools/train_bench.py) -> gpurun_out/tr4/:
#   gaps.txt  wall / union of busy intervals / idle gaps of two steps (tools/trace_gaps.py)
#   perq.txt  per hardware queue (main stream, weight-gradient side stream): launches, kernel time, the top kernels of one step
#   stats.csv rocprofv3's kernel statistics
R=$PWD; O=$R/gpurun_out/tr4; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O -o t --output-format csv -- python3 $R/tools/train_bench.py 3 ${1:-4} > $O/log.txt 2>&1
cd $R
T=$(find $O -name "t_kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py $T 2 2 > $O/gaps.txt 2>&1
python3 - "$T" > $O/perq.txt <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows))
ends = [i for i, e in enumerate(ev) if "multi_tensor_apply" in e[2]]
cuts = [ends[i] for i in range(len(ends)) if i + 1 == len(ends) or ends[i + 1] - ends[i] > 50]
seg = ev[cuts[1] + 1: cuts[2] + 1]
import re
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = n.replace("cine::", "").replace("void ", "")
    return n[:70]
perq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
for a, b, n, q in seg:
    perq[q][short(n)][0] += 1; perq[q][short(n)][1] += b - a
for q, d in perq.items():
    tot = sum(v[1] for v in d.values())
    print(f"queue {q}: {sum(v[0] for v in d.values())} launches, {tot/1e6:.2f} ms")
    for n, (c, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"   {t/1e3:9.1f} us {c:5d} x {t/c/1e3:7.1f}  {n}")
P
find $O -name "*_kernel_trace.csv" -delete
cp $(find $O -name "t_kernel_stats.csv" | head -1) $O/stats.csv

"""Where a training step's wall time goes: from a rocprofv3 kernel trace (t_kernel_trace.csv) take the dispatches between two
markers (the Adam kernels that end each step), and print per step: wall, union of busy intervals, idle gaps, per-queue busy time.
  python tools/trace_gaps.py <t_kernel_trace.csv> [first step] [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows))
# a step ends with the optimizer's fused kernels: find the wgrad-free stretches -> split at multi_tensor_apply / adam kernels
ends = [i for i, e in enumerate(ev) if "multi_tensor_apply" in e[2] or "adam" in e[2].lower()]
cuts = [ends[i] for i in range(len(ends)) if i + 1 == len(ends) or ends[i + 1] - ends[i] > 50]
print("steps found:", len(cuts))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for s in range(first, min(first + nst, len(cuts))):
    seg = ev[cuts[s - 1] + 1: cuts[s] + 1]
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    busy, cur_s, cur_e, gaps = 0, None, None, []
    for a, b, _, _ in seg:
        if cur_e is None or a > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s; gaps.append((a - cur_e, a))
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    busy += cur_e - cur_s
    perq = collections.defaultdict(int)
    for a, b, _, q in seg: perq[q] += b - a
    gaps.sort(reverse=True)
    print(f"step {s}: {len(seg)} dispatches  wall {(t1 - t0) / 1e6:.2f} ms  busy(union) {busy / 1e6:.2f} ms  idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps"
          f"  (> 20 us: {sum(1 for g in gaps if g[0] > 20000)}, sum {sum(g[0] for g in gaps if g[0] > 20000) / 1e6:.2f} ms)")
    print("   kernel time per queue (ms):", {q: round(v / 1e6, 2) for q, v in perq.items()})
    # where the large gaps are: kernel that follows the gap
    for g, a in gaps[:12]:
        nxt = next(e for e in seg if e[0] == a)
        print(f"   gap {g / 1e3:8.1f} us before {nxt[2][:90]}")

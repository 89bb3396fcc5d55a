"""(rocprofv3 --kernel-trace serialises the dispatches of different streams on this stack: the concurrency it reports is 1.0 and the
durations are the isolated ones -- useful only as a per-kernel table of an in-flight COMMAND, not as evidence of overlap.)
Per-kernel time of an in-flight run from a rocprofv3 kernel trace: calls, summed duration, average, and the run's concurrency
(summed kernel time / union of busy intervals) over the last `frac` of the trace (the timed region; the first part is warm-up/capture).
  python tools/trace_inflight_summary.py <t_kernel_trace.csv> [frac=0.5]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * (1 - frac)
ev = [e for e in ev if e[0] >= t_lo]
per = collections.defaultdict(lambda: [0, 0])
busy, cs, ce = 0, None, None
for a, b, k in ev:
    k = re.sub(r"\(.*", "", k.replace("(anonymous namespace)::", "")).replace("void ", "")
    per[k][0] += 1; per[k][1] += b - a
    if ce is None or a > ce:
        if ce is not None: busy += ce - cs
        cs, ce = a, b
    else:
        ce = max(ce, b)
busy += ce - cs
wall = ev[-1][1] - ev[0][0]
tot = sum(v[1] for v in per.values())
print(f"window {wall / 1e6:.2f} ms, busy union {busy / 1e6:.2f} ms, summed kernel time {tot / 1e6:.2f} ms, concurrency {tot / busy:.2f}")
for k, (n, d) in sorted(per.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{k[:84]:84s} n={n:6d} sum={d / 1e6:9.2f} ms ({100 * d / tot:5.1f} %) avg={d / n / 1e3:8.1f} us")

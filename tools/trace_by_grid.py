"""Per (kernel, grid, workgroup) launch statistics from a rocprofv3 kernel trace: count, mean / min / max duration in us.
  python tools/trace_by_grid.py <t_kernel_trace.csv> [name filter] [min total us]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(list)
for r in rows:
    if flt and flt not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"][:100], r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"), r.get("Workgroup_Size_X"), r.get("LDS_Block_Size", ""))
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v):10.1f} us {100 * sum(v) / tot:5.1f}%  n={len(v):5d}  mean {sum(v) / len(v):8.2f}  min {min(v):8.2f}  max {max(v):8.2f}  grid {k[1]}x{k[2]}x{k[3]} wg {k[4]} lds {k[5]}  {k[0]}")

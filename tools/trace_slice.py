"""One slice alone, as a timeline.  Two halves:
  python tools/trace_slice.py run CFG BRANCHES [N]    -- under `rocprofv3 --kernel-trace`: N eager forwards of BASELINE config CFG with the 2-D U-Net
                                                          passes as BRANCHES concurrent runs, a host sync and a pause between them
  python tools/trace_slice.py show t_kernel_trace.csv  -- per forward (dispatches separated by > 1 ms of idle): wall, union of busy intervals, idle,
                                                          summed kernel time, kernel time per queue, the largest gaps, and the dispatch list of the LAST forward"""
import csv, sys, os, time, collections


def run(cfg_id, nb, n):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "deep-cine-cardiac-mri_amd")]
    import torch
    import bench
    from cine_hip import ops, synth
    torch.set_grad_enabled(False)
    dev = torch.device("cuda:0")
    cfg = bench.CONFIGS[cfg_id]()
    ex = synth.make_cine_slice(bench.FRAMES, bench.COILS, bench.H, bench.W, accel=cfg["accel"], seed=0, noise_std=cfg["noise"])
    net = cfg["hip"]().eval(); synth.fill_parameters_(net, cfg["wseed"], keep=cfg["keep"]); net = net.to(dev)
    mk, mask = ex["masked_kspace"].to(dev), ex["mask"].to(dev)
    extra = (ex["sens_maps"].to(dev),) if cfg["needs_sens"] else ()
    with ops.branches(nb):
        for _ in range(n + 2):
            net(mk, mask, *extra)
            torch.cuda.synchronize(); time.sleep(0.005)


def show(path, dump=True):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0),
                 int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)) for r in rows)
    segs, cur = [], [ev[0]]
    for e in ev[1:]:
        if e[0] - max(x[1] for x in cur[-8:]) > 1_000_000:
            segs.append(cur); cur = []
        cur.append(e)
    segs.append(cur)
    segs = [s for s in segs if len(s) > 50]
    print("forwards found:", len(segs))
    for k, seg in enumerate(segs):
        t0, t1 = seg[0][0], max(e[1] for e in seg)
        busy, cs, ce, gaps = 0, None, None, []
        for a, b, *_ in seg:
            if ce is None or a > ce:
                if ce is not None:
                    busy += ce - cs; gaps.append((a - ce, a))
                cs, ce = a, b
            else:
                ce = max(ce, b)
        busy += ce - cs
        perq = collections.defaultdict(int)
        for a, b, _, q, *_ in seg: perq[q] += b - a
        print(f"forward {k}: {len(seg)} dispatches  wall {(t1 - t0) / 1e6:.3f} ms  busy(union) {busy / 1e6:.3f}  idle {(t1 - t0 - busy) / 1e6:.3f} in {len(gaps)} gaps  "
              f"sum of kernel time {sum(e[1] - e[0] for e in seg) / 1e6:.3f}  per queue {dict((q, round(v / 1e6, 2)) for q, v in perq.items())}")
    if dump:
        seg = segs[-1]
        t0 = seg[0][0]
        qs = {q: i for i, q in enumerate(sorted(set(e[3] for e in seg)))}
        prev_end = t0
        for a, b, name, q, wg, grid in seg:
            short = name.replace("void cine::(anonymous namespace)::", "").replace("cine::", "").split("(")[0][:60]
            print(f"{(a - t0) / 1e3:9.1f} us  +{(b - a) / 1e3:7.1f}  q{qs[q]}  gap {(a - prev_end) / 1e3:6.1f}  wgs {grid // max(wg, 1):6d}  {short}")
            prev_end = max(prev_end, b)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 3)
    else:
        show(sys.argv[2])

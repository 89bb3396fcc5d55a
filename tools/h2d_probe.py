import sys, os, time, json
sys.argv = ["bench.py", "--no-cpu-baseline", "--headline-only"]
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")]
import bench, torch
args = bench.parse()
torch.cuda.set_device(0); dev = torch.device("cuda", 0); torch.set_grad_enabled(False)
wl = bench.Workload(2, args, 1, 0, 0, dev, 40)
wl.run(10, False); torch.cuda.synchronize()
for mode in (False, True, "inline", False, True, True):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    wl.run(40, True, mode)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"h2d={mode!s:7}  enqueue {1e3 * (t1 - t0):7.2f} ms  total {1e3 * (t2 - t0):7.2f} ms  -> {40 / (t2 - t0):6.1f} slices/s")
# copies alone, while nothing computes / while the graphs run WITHOUT waiting for them (no dependency): does the copy engine slow the kernels?
cs = torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(cs):
    for k in range(40):
        wl.mks[k % wl.S].copy_(wl.host_mk[k % wl.S], non_blocking=True)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"40 copies alone: enqueue {1e3 * (t1 - t0):.2f} ms total {1e3 * (t2 - t0):.2f} ms")
extra = [torch.empty_like(wl.mks[0]) for _ in range(2)]
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(cs):
    for k in range(40):
        extra[k % 2].copy_(wl.host_mk[k % wl.S], non_blocking=True)       # into buffers nobody reads: pure interference
wl.run(40, True, False)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"independent copies beside the replays: enqueue {1e3 * (t1 - t0):.2f} ms total {1e3 * (t2 - t0):.2f} ms -> {40 / (t2 - t0):.1f} slices/s")

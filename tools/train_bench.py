"""Diagnostic: time of one training step (forward + SSIMLoss + backward + Adam) on the HIP path, per kernel family.
usage: train_bench.py [steps] [config: 2 = XF-VarNet (default), 3 = XT-XPDNet, 4 = 3D-CineNet, 5 = CRNN-VarNet]
Under torch.distributed.run (one rank per GPU, RCCL) every rank trains on its own slice and the gradients are averaged with ONE flat
all-reduce per step (cine_hip.shard.GradientAllReduce); CINE_FORCE_COLLECTIVE=1 runs that all-reduce in a world of one rank too."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")]
import ctypes
import torch
import reconstruction.models as M
from reconstruction.utils import SSIMLoss
from cine_hip import synth
from cine_hip._lib import lib

import torch.distributed as dist
from cine_hip import shard
world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
dev = torch.device(f"cuda:{local}")
torch.cuda.set_device(dev)
if "WORLD_SIZE" in os.environ:
    dist.init_process_group("nccl", device_id=dev)
    shard.FORCE_COLLECTIVE = os.environ.get("CINE_FORCE_COLLECTIVE", "0") == "1"
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ex = synth.make_cine_slice(15, 15, 200, 200, accel={2: 4, 3: 8, 4: 6, 5: 8}[cfg], seed=rank)
net = {2: lambda: M.VarNet(6, 8, 3, 16, 3, "XF"), 3: lambda: M.XPDNet(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT"),
       4: lambda: M.CineNet(6, 6, 16, 3, "3D"), 5: lambda: M.VarNet_RNN(5, 8, 3, 16)}[cfg]()
synth.fill_parameters_(net, 1); net = net.to(dev).train()
mk, mask, target = ex["masked_kspace"].to(dev), ex["mask"].to(dev), ex["target"].to(dev)
extra = (ex["sens_maps"].to(dev),) if cfg == 4 else ()
lossf = SSIMLoss().to(dev)
opt = torch.optim.Adam(net.parameters(), lr=3e-4)
sync = shard.GradientAllReduce(net)

def step():
    opt.zero_grad(set_to_none=True)
    out = net(mk, mask, *extra)
    loss = lossf(out.unsqueeze(1), target.unsqueeze(1), target.max())
    loss.backward()
    sync()
    opt.step()
    return loss

if os.environ.get("CINE_EXTRA_STREAMS"):      # diagnostics: idle streams created BEFORE the step's side streams exist (what a long bench process leaves behind)
    _idle = [torch.cuda.Stream() for _ in range(int(os.environ["CINE_EXTRA_STREAMS"]))]
    for s_ in _idle:
        with torch.cuda.stream(s_):
            torch.zeros(1, device=dev)
    torch.cuda.synchronize()
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.time()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n): l = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(f"training step: {dt * 1e3:.1f} ms   loss {float(l):.5f}   peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB" +
      (f"   [{world} rank(s), RCCL gradient all-reduce of {sum(p.numel() for p in sync.params) * 4 / 2**20:.1f} MiB per step: {world / dt:.1f} slices/s]" if dist.is_initialized() else ""))
if rank != 0:
    sys.exit(0)
if os.environ.get("CINE_TRAIN_REGIONS"):      # more regions of the same length: the spread between them, and how long the HOST needs to enqueue a step
    regs, enq = [], []
    for _ in range(int(os.environ["CINE_TRAIN_REGIONS"])):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        regs.append((time.perf_counter() - t0) / n * 1e3); enq.append((t1 - t0) / n * 1e3)
    print("regions (ms per step):", " ".join(f"{r:.2f}" for r in regs), "  host enqueue per step:", " ".join(f"{r:.2f}" for r in enq))
with torch.no_grad():
    for _ in range(2): net(mk, mask, *extra)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): net(mk, mask, *extra)
    torch.cuda.synchronize()
print(f"inference forward (eager): {(time.time() - t0) / n * 1e3:.1f} ms")
L = lib()
nf = L.cine_profile_families()
L.cine_profile_begin()
step()
ms = (ctypes.c_double * nf)(); cnt = (ctypes.c_long * nf)()
L.cine_profile_end(ms, cnt, nf)
for i in range(nf):
    print(f"  {L.cine_profile_family_name(i).decode():12s} {ms[i]:8.2f} ms  {cnt[i]:5d} launches")
print(f"  sum {sum(ms):.2f} ms")

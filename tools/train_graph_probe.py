"""Probe: can one whole training step (forward + SSIMLoss + backward through the HIP gradient kernels + Adam) of a BASELINE
configuration be captured into ONE hipGraph, and what does replaying it buy?
usage: train_graph_probe.py [config 2|3|4|5] [steps]
Prints the eager and the graph-replayed step time, and the loss sequences of both (they must agree: same kernels, same order)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")]
import torch
import reconstruction.models as M
from reconstruction.models.varnet import SensitivityModel
from reconstruction.utils import SSIMLoss
from cine_hip import synth, train

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
make = {2: lambda: M.VarNet(6, 8, 3, 16, 3, "XF"), 3: lambda: M.XPDNet(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT"),
        4: lambda: M.CineNet(6, 6, 16, 3, "3D"), 5: lambda: M.VarNet_RNN(5, 8, 3, 16)}[cfg]
ex = synth.make_cine_slice(15, 15, 200, 200, accel={2: 4, 3: 8, 4: 6, 5: 8}[cfg], seed=0)
mk, mask, target = ex["masked_kspace"].to(dev), ex["mask"].to(dev), ex["target"].to(dev)
extra = (ex["sens_maps"].to(dev),) if cfg == 4 else ()
kw = {} if cfg == 4 else {"acs": SensitivityModel.acs_window(mask)}


def build():
    net = make()
    synth.fill_parameters_(net, 1)
    net = net.to(dev).train()
    return net, SSIMLoss().to(dev)


def run_eager():
    net, lossf = build()
    opt = torch.optim.Adam(net.parameters(), lr=3e-4)
    losses = []

    def step():
        opt.zero_grad(set_to_none=True)
        out = net(mk, mask, *extra, **kw)
        loss = lossf(out.unsqueeze(1), target.unsqueeze(1), target.max())
        loss.backward()
        opt.step()
        return loss.detach()
    for _ in range(3):
        losses.append(step())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        losses.append(step())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, [float(l) for l in losses]


def run_graphed():
    net, lossf = build()
    opt = torch.optim.Adam(net.parameters(), lr=3e-4, capturable=True)
    gs = train.GraphedTrainingStep(net, lambda out, tgt: lossf(out.unsqueeze(1), tgt.unsqueeze(1), tgt.max()), opt,
                                   (mk, mask) + extra, target, forward_kwargs=kw, warmup=3)
    losses = list(gs.warmup_losses)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        losses.append(gs.step(mk, mask, *extra, target=target).clone())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, [float(l) for l in losses]


te, le = run_eager()
print(f"cfg {cfg}: eager step {te * 1e3:.2f} ms   losses {[round(x, 6) for x in le]}")
tg, lg = run_graphed()
print(f"cfg {cfg}: one hipGraph per step {tg * 1e3:.2f} ms   losses {[round(x, 6) for x in lg]}")
print(f"cfg {cfg}: max |loss difference| {max(abs(a - b) for a, b in zip(le, lg)):.2e}   speed-up {te / tg:.3f}x")

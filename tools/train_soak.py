"""Diagnostic: N training steps of a BASELINE configuration on the HIP path; device memory in use after every 10 steps (it must not grow)
and the loss curve.  usage: train_soak.py [steps] [config]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")]
import torch
import reconstruction.models as M
from reconstruction.utils import SSIMLoss
from cine_hip import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
ex = synth.make_cine_slice(15, 15, 200, 200, accel={2: 4, 3: 8, 4: 6, 5: 8}[cfg], seed=0)
net = {2: lambda: M.VarNet(6, 8, 3, 16, 3, "XF"), 3: lambda: M.XPDNet(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT"),
       4: lambda: M.CineNet(6, 6, 16, 3, "3D"), 5: lambda: M.VarNet_RNN(5, 8, 3, 16)}[cfg]()
synth.fill_parameters_(net, 1); net = net.to(dev).train()
mk, mask, target = ex["masked_kspace"].to(dev), ex["mask"].to(dev), ex["target"].to(dev)
extra = (ex["sens_maps"].to(dev),) if cfg == 4 else ()
lossf = SSIMLoss().to(dev)
opt = torch.optim.Adam(net.parameters(), lr=3e-4)
for k in range(steps):
    opt.zero_grad(set_to_none=True)
    loss = lossf(net(mk, mask, *extra).unsqueeze(1), target.unsqueeze(1), target.max())
    loss.backward()
    opt.step()
    if k % 10 == 9 or k == 0:
        torch.cuda.synchronize()
        print(f"step {k + 1:3d}  loss {float(loss.detach()):.5f}  allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB  reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB", flush=True)

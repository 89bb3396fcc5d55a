"""tools/host_profile.py [config]: cProfile of the HOST side of ten training steps (where the Python thread spends its time while it enqueues a step:
blocking reads of the device show up as {method 'cpu'}; ctypes launches are charged to their caller's own time)."""
import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")]
import torch
import reconstruction.models as M
from reconstruction.utils import SSIMLoss
from cine_hip import synth
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
ex = synth.make_cine_slice(15, 15, 200, 200, accel={2: 4, 3: 8, 4: 6, 5: 8}[cfg], seed=0)
net = {2: lambda: M.VarNet(6, 8, 3, 16, 3, "XF"), 3: lambda: M.XPDNet(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT"),
       4: lambda: M.CineNet(6, 6, 16, 3, "3D"), 5: lambda: M.VarNet_RNN(5, 8, 3, 16)}[cfg]()
synth.fill_parameters_(net, 1); net = net.to(dev).train()
mk, mask, target = ex["masked_kspace"].to(dev), ex["mask"].to(dev), ex["target"].to(dev)
extra = (ex["sens_maps"].to(dev),) if cfg == 4 else ()
lossf = SSIMLoss().to(dev)
opt = torch.optim.Adam(net.parameters(), lr=3e-4)
def step():
    opt.zero_grad(set_to_none=True)
    out = net(mk, mask, *extra)
    loss = lossf(out.unsqueeze(1), target.unsqueeze(1), target.max())
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])

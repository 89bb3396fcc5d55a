"""Diagnostic: cfg-2-size sensitivity network (225 planes of 200 x 200), HIP gradients vs the float64 / float32 CPU oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd"), os.path.join(ROOT, "tests")]
import torch
import reconstruction.models as M
from cine_hip import synth
from oracle import varnet_ref as V
from conftest import rnd
dev = torch.device("cuda:0")
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)), float((a - b).norm() / b.norm())
torch.set_num_threads(32)
ex = synth.make_cine_slice(15, 15, 200, 200, accel=4, seed=0)
hip = M.VarNet(1, 8, 3, 16, 3, "XF"); synth.fill_parameters_(hip, 1)
g = rnd(1, 1, 1, 15, 200, 200, 2)
res = {}
for dt in (torch.float64, torch.float32):
    ref = V.VarNet(1, 8, 3, 16, 3, "XF"); ref.load_state_dict(hip.state_dict()); ref = ref.to(dt)
    t0 = time.time()
    S = ref.sens_net(ex["masked_kspace"].to(dt), ex["mask"]); (S * g.to(dt)).sum().backward()
    res[dt] = {k: p.grad.clone() for k, p in ref.sens_net.named_parameters()}
    print(dt, f"{time.time() - t0:.1f}s", flush=True)
hip = hip.to(dev)
S = hip.sens_net(ex["masked_kspace"].to(dev), ex["mask"].to(dev)); (S * g.to(dev)).sum().backward()
for k, p in hip.sens_net.named_parameters():
    print(f"{k:55s} hip-vs-f64 max {rel(p.grad, res[torch.float64][k])[0]:.2e} l2 {rel(p.grad, res[torch.float64][k])[1]:.2e}   cpu32-vs-f64 max {rel(res[torch.float32][k], res[torch.float64][k])[0]:.2e} l2 {rel(res[torch.float32][k], res[torch.float64][k])[1]:.2e}")

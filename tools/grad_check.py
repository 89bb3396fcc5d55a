"""Diagnostic: gradients of the HIP training path against the CPU oracle's autograd (float64 = truth, float32 = the
reference arithmetic's own noise floor).  python tools/grad_check.py [unet|normunet|xfyf|dc|varnet ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")]
import numpy as np
import torch

import reconstruction.models as M
from reconstruction.models.denoisers.unet import Unet
from reconstruction.models.denoisers.norm_unet import NormUnet
from cine_hip import synth, ops, autograd as ag
from oracle import varnet_ref as V, regularisers as R

dev = torch.device("cuda:0")


def rnd(seed, *shape):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def report(name, hip_named, ref64_named, ref32_named):
    worst = 0.0
    for k in ref64_named:
        e = rel(hip_named[k], ref64_named[k])
        e32 = rel(ref32_named[k], ref64_named[k])
        worst = max(worst, e)
        flag = "  <<<" if e > max(1e-4, 20 * e32) else ""
        print(f"  {name} {k:60s} hip-vs-f64 {e:.2e}   f32-vs-f64 {e32:.2e}{flag}")
    print(f"  {name}: worst {worst:.2e}")


def grads_of(module, loss):
    module.zero_grad(set_to_none=True)
    loss.backward()
    return {k: p.grad.detach().clone() for k, p in module.named_parameters() if p.grad is not None}


def check_unet(n=6, cin=2, cout=2, chans=4, pools=2, h=32, w=16, seed=0):
    hip = Unet(chans, pools, cin, cout).to(dev)
    synth.fill_parameters_(hip, seed + 1)
    x = rnd(seed, n, cin, h, w)
    gy = rnd(seed + 7, n, cout, h, w)
    res = {}
    for dt in (torch.float64, torch.float32):
        ref = R.Unet(chans, pools, cin, cout).to(dt)
        ref.load_state_dict({k: v.to(dt) for k, v in hip.state_dict().items()})
        xr = x.to(dt).clone().requires_grad_(True)
        y = ref(xr)
        g = grads_of(ref, (y * gy.to(dt)).sum())
        g["__x__"] = xr.grad
        g["__y__"] = y.detach()
        res[dt] = g
    xh = x.detach().to(dev).requires_grad_(True)
    with torch.enable_grad():
        yh = hip(xh)
        gh = grads_of(hip, (yh * gy.to(dev)).sum())
    gh["__x__"] = xh.grad
    gh["__y__"] = yh.detach()
    report(f"unet n{n} c{chans} p{pools} {h}x{w}", gh, res[torch.float64], res[torch.float32])


def check_normunet(n=5, chans=4, pools=2, h=24, w=20, seed=3):
    hip = NormUnet(chans, pools).to(dev)
    synth.fill_parameters_(hip, seed + 1)
    x = rnd(seed, n, 1, h, w, 2)
    gy = rnd(seed + 7, n, 1, h, w, 2)
    res = {}
    for dt in (torch.float64, torch.float32):
        ref = R.NormUnet(chans, pools).to(dt)
        ref.load_state_dict({k: v.to(dt) for k, v in hip.state_dict().items()})
        xr = x.to(dt).clone().requires_grad_(True)
        y = ref(xr)
        g = grads_of(ref, (y * gy.to(dt)).sum())
        g["__x__"] = xr.grad; g["__y__"] = y.detach()
        res[dt] = g
    xh = x.detach().to(dev).requires_grad_(True)
    with torch.enable_grad():
        yh = hip(xh)
        gh = grads_of(hip, (yh * gy.to(dev)).sum())
    gh["__x__"] = xh.grad; gh["__y__"] = yh.detach()
    report(f"normunet {h}x{w}", gh, res[torch.float64], res[torch.float32])


def check_varnet(dyn="XF", casc=2, t=5, c=3, h=24, w=20, share=False, seed=0, chans=4, pools=2):
    ex = synth.make_cine_slice(t, c, h, w, accel=4, center_lines=4, seed=seed)
    hip = M.VarNet(casc, 4, 2, chans, pools, dyn, share).to(dev)
    synth.fill_parameters_(hip, seed + 1)
    with torch.no_grad():
        for i, cs in enumerate(hip.cascades):
            cs.lambda_reg.fill_(0.3 + 0.2 * i)
    target = rnd(seed + 5, 1, t, h, w).abs()
    res = {}
    for dt in (torch.float64, torch.float32):
        ref = V.VarNet(casc, 4, 2, chans, pools, dyn, share).to(dt)
        ref.load_state_dict({k: v.to(dt) for k, v in hip.state_dict().items()})
        out = ref(ex["masked_kspace"].to(dt), ex["mask"])
        g = grads_of(ref, ((out - target.to(dt)) ** 2).sum())
        g["__y__"] = out.detach()
        res[dt] = g
    with torch.enable_grad():
        out = hip(ex["masked_kspace"].to(dev), ex["mask"].to(dev))
        gh = grads_of(hip, ((out - target.to(dev)) ** 2).sum())
    gh["__y__"] = out.detach()
    report(f"varnet {dyn} share={share}", gh, res[torch.float64], res[torch.float32])


if __name__ == "__main__":
    what = sys.argv[1:] or ["unet", "normunet", "varnet"]
    torch.manual_seed(0)
    if "unet" in what:
        check_unet()
        check_unet(n=4, chans=8, pools=3, h=48, w=16, seed=2)
        check_unet(n=3, chans=4, pools=2, h=26, w=22, seed=4)        # odd sizes after pooling: zero-pad crop
    if "normunet" in what:
        check_normunet()
    if "varnet" in what:
        check_varnet("XF")
        check_varnet("XT", share=True)
        check_varnet("2D")


def check_golden(tag="XT", dyn="XT", share=False):
    """the varnet_grad.npz case: HIP vs the stored reference gradients, per parameter"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden, state_dict_from
    from reconstruction.data import transforms
    from reconstruction.utils import SSIMLoss
    g = load_golden("varnet_grad")
    net = M.VarNet(2, 4, 2, 4, 2, dyn, share)
    net.load_state_dict(state_dict_from(g, f"{tag}::sd::"), strict=True)
    net = net.to(dev)
    mk, mask, target = (torch.from_numpy(g[k]).to(dev) for k in ("masked_kspace", "mask", "target"))
    with torch.enable_grad():
        out = net(mk, mask)
        tgt, o = transforms.center_crop_to_smallest(target, out)
        loss = SSIMLoss().to(dev)(o.unsqueeze(1), tgt.unsqueeze(1), data_range=tgt.max())
        gh = grads_of(net, loss)
    print("loss", float(loss), float(g[f"{tag}_loss"]), "out err", rel(out.detach(), torch.from_numpy(g[f"{tag}_out"])))
    for k, v in gh.items():
        print(f"  {k:60s} {rel(v, torch.from_numpy(g[f'{tag}::grad::{k}'])):.2e}  floor {float(g[f'{tag}::floor::{k}']):.1e}  gmax {float(v.abs().max()):.2e}")


if "golden" in sys.argv[1:]:
    check_golden()

"""Diagnostic: per-Function gradient errors (HIP vs the float64 oracle) on the varnet_grad.npz XT case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import reconstruction.models as M
from cine_hip import ops, autograd as ag
from oracle import varnet_ref as V, centered_fft as cf, complex_ops as co
from conftest import load_golden, state_dict_from, rnd

dev = torch.device("cuda:0")
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))

tag, dyn = (sys.argv[1], sys.argv[1][:2]) if len(sys.argv) > 1 else ("XT", "XT")
g = load_golden("varnet_grad")
hip = M.VarNet(2, 4, 2, 4, 2, dyn, tag.endswith("ws")); hip.load_state_dict(state_dict_from(g, f"{tag}::sd::")); hip = hip.to(dev)
ref = V.VarNet(2, 4, 2, 4, 2, dyn, tag.endswith("ws")); ref.load_state_dict(state_dict_from(g, f"{tag}::sd::")); ref = ref.double()
mk, mask = torch.from_numpy(g["masked_kspace"]), torch.from_numpy(g["mask"])
with torch.enable_grad():
    # reference chain in float64, keeping the intermediates
    sens_r = ref.sens_net(mk.double(), mask)
    x0_r = ref.cascades[0].sens_reduce(mk.double(), sens_r)
    # ---- xfyf on x0
    img = x0_r.detach().squeeze(2).float()
    gout = rnd(5, *x0_r.shape)
    ir = img.double().requires_grad_(True)
    yr = ref.cascades[0].xfyf_transform(ir); (yr * gout.double()).sum().backward()
    ih = img.to(dev).requires_grad_(True)
    yh = hip.cascades[0].xfyf_transform(ih); (yh * gout.to(dev)).sum().backward()
    print("xfyf fwd", rel(yh.detach(), yr.detach()), "gimg", rel(ih.grad, ir.grad))
    # ---- image dc
    m = yr.detach().float()
    sens = sens_r.detach().float()
    lam = ref.cascades[0].lambda_reg.detach().float()
    mr, sr, lr = m.double().requires_grad_(True), sens.double().requires_grad_(True), lam.double().requires_grad_(True)
    v = torch.nn.functional.softplus(lr)
    kth = cf.fft2c(co.complex_mul(mr, sr)); mkk = mask.double()
    k = (1 - mkk) * kth + mkk * (kth + v * mk.double()) / (1 + v)
    out_r = co.complex_mul(cf.ifft2c(k), co.complex_conj(sr)).sum(dim=2, keepdim=True)
    (out_r * gout.double()).sum().backward()
    md, sd, ld = m.to(dev).requires_grad_(True), sens.to(dev).requires_grad_(True), lam.to(dev).requires_grad_(True)
    zf = ag.CoilReduceFn.apply(mk.to(dev), sd, mask.to(dev))
    out = ag.ImageDcFn.apply(md, sd, zf, mask.to(dev), ld)
    (out * gout.to(dev)).sum().backward()
    print("dc fwd", rel(out.detach(), out_r.detach()), "gm", rel(md.grad, mr.grad), "gs", rel(sd.grad, sr.grad), "glam", rel(ld.grad, lr.grad))
    # ---- sens net
    hip.zero_grad(); ref.zero_grad()
    gs = rnd(6, *sens_r.shape)
    (sens_r * gs.double()).sum().backward()
    sh = hip.sens_net(mk.to(dev), mask.to(dev)); (sh * gs.to(dev)).sum().backward()
    e = {k: rel(p.grad, dict(ref.named_parameters())[k].grad) for k, p in hip.named_parameters() if p.grad is not None}
    print("sens fwd", rel(sh.detach(), sens_r.detach()), "worst param grad", max(e.values()))
    # ---- abs
    xa = out_r.detach().float().squeeze(2)
    ga = rnd(7, *xa.shape[:-1])
    ar = xa.double().requires_grad_(True); (co.complex_abs(ar) * ga.double()).sum().backward()
    ah = xa.to(dev).requires_grad_(True); (ag.AbsFn.apply(ah) * ga.to(dev)).sum().backward()
    print("abs", rel(ah.grad, ar.grad))
    # ---- SSIM loss gradient on the device vs the CPU (same torch code)
    from reconstruction.utils import SSIMLoss
    from reconstruction.data import transforms
    target = torch.from_numpy(g["target"])
    o = torch.from_numpy(g[f"{tag}_out"])
    oc = o.clone().requires_grad_(True)
    t1, o1 = transforms.center_crop_to_smallest(target, oc); l1 = SSIMLoss()(o1.unsqueeze(1), t1.unsqueeze(1), t1.max()); l1.backward()
    o64 = o.double().requires_grad_(True)
    t3, o3 = transforms.center_crop_to_smallest(target.double(), o64); l3 = SSIMLoss().double()(o3.unsqueeze(1), t3.unsqueeze(1), t3.max()); l3.backward()
    od = o.to(dev).requires_grad_(True)
    t2, o2 = transforms.center_crop_to_smallest(target.to(dev), od); l2 = SSIMLoss().to(dev)(o2.unsqueeze(1), t2.unsqueeze(1), t2.max()); l2.backward()
    print("ssim loss cpu/gpu/f64", float(l1), float(l2), float(l3), "grad gpu-vs-f64", rel(od.grad, o64.grad), "cpu32-vs-f64", rel(oc.grad, o64.grad))
    # ---- whole model, plain sum loss, vs the float64 oracle
    hip.zero_grad(); ref.zero_grad()
    w = rnd(8, 1, 5, 24, 20)
    (ref(mk.double(), mask) * w.double()).sum().backward()
    (hip(mk.to(dev), mask.to(dev)) * w.to(dev)).sum().backward()
    rp = dict(ref.named_parameters())
    e = {k: rel(p.grad, rp[k].grad) for k, p in hip.named_parameters()}
    print("whole model, linear loss: worst", max(e.values()), max(e, key=e.get))
    # ---- the chain with every intermediate's gradient kept, HIP vs float64 oracle
    hip.zero_grad(); ref.zero_grad()
    md_, mkd = mask.to(dev), mk.to(dev)
    S = hip.sens_net(mkd, md_); S.retain_grad()
    x0 = ag.CoilReduceFn.apply(mkd, S, None); x0.retain_grad()
    zf = ag.CoilReduceFn.apply(mkd, S, md_); zf.retain_grad()
    hs = {"S": S, "x0": x0, "zf": zf}
    x = x0
    for i, cas in enumerate(hip.cascades):
        m_ = cas.regularise(x); m_.retain_grad(); hs[f"m{i}"] = m_
        x = ag.ImageDcFn.apply(m_, S, zf, md_, cas.lambda_reg); x.retain_grad(); hs[f"x{i + 1}"] = x
    out = ag.AbsFn.apply(x.squeeze(2))
    (out * w.to(dev)).sum().backward()
    Sr = ref.sens_net(mk.double(), mask); Sr.retain_grad()
    c0 = ref.cascades[0]
    x0r = c0.sens_reduce(mk.double(), Sr); x0r.retain_grad()
    rs = {"S": Sr, "x0": x0r}
    xr = x0r
    for i, cas in enumerate(ref.cascades):
        mr_ = cas.xfyf_transform(xr.squeeze(2)); mr_.retain_grad(); rs[f"m{i}"] = mr_
        v = torch.nn.functional.softplus(cas.lambda_reg)
        kth = cas.sens_expand(mr_, Sr)
        kk = (1 - mkk) * kth + mkk * (kth + v * mk.double()) / (1 + v)
        xr = cas.sens_reduce(kk, Sr); xr.retain_grad(); rs[f"x{i + 1}"] = xr
    outr = co.complex_abs(xr.squeeze(2))
    (outr * w.double()).sum().backward()
    for k in rs:
        print(f"  {k}: value {rel(hs[k].detach(), rs[k].detach()):.2e}  grad {rel(hs[k].grad, rs[k].grad):.2e}")
    d = (hs["x1"].grad.double().cpu() - rs["x1"].grad).abs().squeeze() / rs["x1"].grad.abs().max()      # (t, h, w, 2)
    print("x1 grad error: elements > 1e-4:", int((d > 1e-4).sum()), "of", d.numel(), "rows (h) hit:", sorted(set((d > 1e-4).nonzero()[:, 1].tolist())),
          "cols (w) hit:", sorted(set((d > 1e-4).nonzero()[:, 2].tolist())))
    # ---- xfyf of cascade 1 on x1 with the oracle's incoming gradient; tiny input perturbations toggle a kink artefact
    x1 = rs["x1"].detach().squeeze(2); g1 = rs["m1"].grad.detach()
    c1r, c1h = ref.cascades[1], hip.cascades[1]
    for trial in range(6):
        pert = x1 * (1 + (1e-6 * rnd(100 + trial, *x1.shape).double() if trial else 0))
        ir = pert.clone().requires_grad_(True); (c1r.xfyf_transform(ir) * g1).sum().backward()
        i32 = pert.float().clone().requires_grad_(True)
        r32 = V.VarNet(2, 4, 2, 4, 2, dyn, tag.endswith("ws")); r32.load_state_dict(state_dict_from(g, f"{tag}::sd::"))
        (r32.cascades[1].xfyf_transform(i32) * g1.float()).sum().backward()
        ih = pert.float().to(dev).requires_grad_(True); (c1h.xfyf_transform(ih) * g1.float().to(dev)).sum().backward()
        dh = (ih.grad.double().cpu() - ir.grad).abs() / ir.grad.abs().max()
        print(f"  trial {trial}: hip-vs-f64 {float(dh.max()):.2e} (cols hit {sorted(set((dh > 1e-4).nonzero()[:, 3].tolist()))})   cpu-f32-vs-f64 {rel(i32.grad, ir.grad):.2e}")

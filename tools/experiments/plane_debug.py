"""Plane-persistent U-Net kernel vs layer-by-layer launches on cfg-2 shaped planes (run twice: the plain run saves the
reference, the run with CINE_PLANE_KERNEL=1 compares)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")]
import torch
from cine_hip import ops, synth
from reconstruction.models.denoisers.unet import Unet
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
pools = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
u0, u1 = Unet(16, pools, 2, 2).eval(), Unet(16, pools, 2, 2).eval()
synth.fill_parameters_(u0, 3); synth.fill_parameters_(u1, 4)
u0.to(dev); u1.to(dev)
w = ops.UnetWeights([u0, u1])
x = torch.randn(n, 2, 208, 16, device=dev)
y = ops.unet2d_forward(x, w)
torch.cuda.synchronize()
path = os.path.join(sys.argv[3] if len(sys.argv) > 3 else "/tmp", "plane_ref_%d_%d.pt" % (n, pools))
if not os.environ.get("CINE_PLANE_KERNEL"):
    torch.save(y.cpu(), path); print("saved reference", float(y.abs().max()))
else:
    ref = torch.load(path)
    d = (y.cpu() - ref).abs()
    per = d.flatten(1).max(dim=1).values
    bad = (per > 1e-4 * float(ref.abs().max())).nonzero().flatten()
    print("max abs diff", float(d.max()), "ref max", float(ref.abs().max()), "bad planes", bad.numel(), bad[:20].tolist())

"""Time cine_image_dc and the hybrid-space FFT+DC chain it replaces at cfg-2 size (eager launches, cuda events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")):
    sys.path.insert(0, p)
import torch
from cine_hip import ops
dev = torch.device("cuda:0")
t, c, h, w = 15, 15, 200, 200
img = torch.randn(1, t, 1, h, w, 2, device=dev); sens = torch.randn(1, 1, c, h, w, 2, device=dev)
zf = torch.randn(1, t, 1, h, w, 2, device=dev); kref = torch.randn(1, t, c, h, w, 2, device=dev)
mask = (torch.rand(1, t, 1, h, 1, 1, device=dev) < 0.25).byte(); lam = torch.tensor([0.54], device=dev)
hyb = torch.empty_like(kref)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("image_dc            %.1f us" % timeit(lambda: ops.image_dc(img, sens, zf, mask, lam)))
print("expand_dc_hybrid    %.1f us" % timeit(lambda: ops.expand_dc_hybrid(img, sens, kref, mask, lam, out=hyb)))
print("hybrid_reduce       %.1f us" % timeit(lambda: ops.hybrid_reduce(hyb, sens)))
print("kspace_to_hybrid    %.1f us" % timeit(lambda: ops.kspace_to_hybrid(kref, out=hyb)))
print("masked k->hybrid    %.1f us" % timeit(lambda: ops.kspace_to_hybrid(kref, out=hyb, mask=mask)))

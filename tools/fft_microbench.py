#!/usr/bin/env python3
"""Time the FFT+DC passes of one cfg-2 cascade in isolation (hipEvents, 20 reps each)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")]
import torch
from cine_hip import ops

dev = torch.device("cuda:0")
t, c, h, w = 15, 15, 200, 200
k = torch.randn(1, t, c, h, w, 2, device=dev)
sens = torch.randn(1, 1, c, h, w, 2, device=dev)
mask = (torch.rand(1, t, 1, h, 1, 1, device=dev) > 0.75).byte()
lam = torch.tensor([0.54], device=dev)
hyb = ops.kspace_to_hybrid(k)
img = ops.hybrid_reduce(hyb, sens)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20

def timeit(name, fn, nbytes):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:28s} {us:8.1f} us   {nbytes / us / 1e6:6.2f} TB/s (algorithmic {nbytes / 1e6:.0f} MB)")

K = k.numel() * 4
I = img.numel() * 4
timeit("kspace_to_hybrid (C1)", lambda: ops.kspace_to_hybrid(k, out=hyb), 2 * K)
timeit("hybrid_reduce (R2)", lambda: ops.hybrid_reduce(hyb, sens), K + 2 * I)
timeit("expand_dc_hybrid (E1+E2C1)", lambda: ops.expand_dc_hybrid(img, sens, k, mask, lam, out=hyb), 2 * I + K + K)
timeit("sens_expand_dc (E1+E2)", lambda: ops.sens_expand_dc(img, sens, k, mask, lam, out=hyb), 2 * I + K + K)
timeit("fft2c 225 imgs", lambda: ops.fft2c(k), 2 * K)

print("-- DC sensitivity to mask density (expand_dc_hybrid = E1 + E2C1)")
for dens in (0.0, 0.25, 1.0):
    m2 = (torch.rand(1, t, 1, h, 1, 1, device=dev) < dens).byte()
    timeit(f"expand_dc_hybrid dens={dens}", lambda: ops.expand_dc_hybrid(img, sens, k, m2, lam, out=hyb), 2 * I + K + K)
timeit("sens_expand (no DC: E1+col fwd)", lambda: ops.sens_expand_dc(img, sens, out=hyb), 2 * I + K)

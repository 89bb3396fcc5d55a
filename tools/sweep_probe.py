"""Time cine_bcrnn_sweep for c = 8 / 16 / 24 hidden channels (one / two / three 8-channel chunks per step): what a chunk costs in the step chain."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "deep-cine-cardiac-mri_amd"))
import torch
from cine_hip import ops
dev = torch.device("cuda:0")
for c in (8, 16, 24):
    T, h, w = 15, 200, 200
    P = torch.randn(T, c, h, w, device=dev) * 0.1
    W = torch.randn(c, c, 3, 3, device=dev) * 0.05
    wp = ops.pack_conv3x3(W)
    zero = torch.zeros(1, c, h, w, device=dev)
    for _ in range(3): ops.bcrnn_sweep(P, wp, zero)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 20
    for _ in range(R): ops.bcrnn_sweep(P, wp, zero)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    print(f"c={c}: sweep {dt*1e6:.1f} us = {dt*1e6/16:.2f} us per step launch", flush=True)

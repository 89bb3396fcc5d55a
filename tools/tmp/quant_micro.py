import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")):
    sys.path.insert(0, p)
import torch
from cine_hip import ops
dev = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (c, h, w, cout) in ((16, 208, 16, 16), (32, 104, 8, 32), (64, 52, 4, 64)):
    wt = ops.pack_conv3x3(torch.randn(cout, c, 3, 3, device=dev) / 10)
    for n in (256, 320, 384, 400, 448, 512, 768):
        x = torch.randn(n, c, h, w, device=dev); px = ops.instnorm_partials(x)
        t = timeit(lambda: ops.conv3x3_in([(x, px, 1)], wt, cout, h, w))
        print(f"{c}->{cout} @{h}x{w} n={n}: {t:.1f} us  ({t / n * 400:.1f} us per 400 planes)")

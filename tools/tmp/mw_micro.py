import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")):
    sys.path.insert(0, p)
import torch
from cine_hip import ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def conv_ex(x0, p0, c0, m0, h0, w0, x1, p1, c1, m1, h1, w1, add, cout, h, w, cin):
    wt = ops.pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) / 10)
    n = x0.shape[0]
    y = torch.empty(n, cout, h, w, device=dev); py = torch.empty(n, cout, L.cine_conv_stat_partials(cout, h, w, 0), 3, device=dev)
    def run():
        _lib.check(L.cine_conv3x3_ex(x0.data_ptr(), None if p0 is None else p0.data_ptr(), 0 if p0 is None else p0.shape[2], c0, m0, h0, w0,
                                     None if x1 is None else x1.data_ptr(), None if p1 is None else p1.data_ptr(), 0 if p1 is None else p1.shape[2], c1, m1, h1, w1, add,
                                     wt.data_ptr(), None, None, 0, y.data_ptr(), py.data_ptr(), n, cout, h, w, 1e-5, 0.2, torch.cuda.current_stream().cuda_stream))
    return run
n = 200
first = torch.randn(n, 16, 200, 16, device=dev); pf = ops.instnorm_partials(first)
s0 = torch.randn(n, 16, 100, 8, device=dev); ps0 = ops.instnorm_partials(s0)
s0w = torch.randn(n, 64, 100, 8, device=dev); ps0w = ops.instnorm_partials(s0w)
s1 = torch.randn(n, 32, 50, 4, device=dev); ps1 = ops.instnorm_partials(s1)
s1w = torch.randn(n, 64, 50, 4, device=dev); ps1w = ops.instnorm_partials(s1w)
print("scale0 ACT 16->16 @100x8 (fast)        %.1f us" % timeit(conv_ex(s0, ps0, 16, 1, 100, 8, None, None, 0, 0, 0, 0, 0, 16, 100, 8, 16)))
print("scale0 ACT 64->16 @100x8 (fast, ref)   %.1f us" % timeit(conv_ex(s0w, ps0w, 64, 1, 100, 8, None, None, 0, 0, 0, 0, 0, 16, 100, 8, 64)))
print("scale0 DWT(first 16ch) 64->16 @100x8   %.1f us" % timeit(conv_ex(first, pf, 16, 3 | 8, 200, 16, None, None, 0, 0, 0, 0, 0, 16, 100, 8, 64)))
print("scale0 IWT(64ch@50x4)+skip 16->16      %.1f us" % timeit(conv_ex(s1w, ps1w, 64, 4 | 8, 50, 4, s0, ps0, 16, 1, 100, 8, 1, 16, 100, 8, 16)))
print("scale1 ACT 32->32 @50x4 (fast)         %.1f us" % timeit(conv_ex(s1, ps1, 32, 1, 50, 4, None, None, 0, 0, 0, 0, 0, 32, 50, 4, 32)))
print("scale1 DWT(16ch@100x8) 64->32 @50x4    %.1f us" % timeit(conv_ex(s0, ps0, 16, 3 | 8, 100, 8, None, None, 0, 0, 0, 0, 0, 32, 50, 4, 64)))
print("final IWT(64ch@100x8)+first 16->10     %.1f us" % timeit(conv_ex(s0w, ps0w, 64, 4 | 8, 100, 8, first, pf, 16, 1, 200, 16, 1, 10, 200, 16, 16)))
print("first 12->16 @200x16 (fast)            %.1f us" % timeit(conv_ex(torch.randn(n, 12, 200, 16, device=dev), None, 12, 0, 200, 16, None, None, 0, 0, 0, 0, 0, 16, 200, 16, 12)))

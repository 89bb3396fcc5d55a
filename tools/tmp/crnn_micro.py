import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")):
    sys.path.insert(0, p)
import torch
from cine_hip import ops
dev = torch.device("cuda:0")
hid = torch.randn(1, 16, 200, 200, device=dev); add = torch.randn(1, 16, 200, 200, device=dev)
w = ops.pack_conv3x3(torch.randn(16, 16, 3, 3, device=dev) / 12)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        h = hid
        for i in range(15): h = ops.conv3x3_sum([h], w, None, 16, addend=add, relu=True)
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    h = hid
    for i in range(15): h = ops.conv3x3_sum([h], w, None, 16, addend=add, relu=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): g.replay()
e1.record(); torch.cuda.synchronize()
print("h2h step (graph replay of a 15-step chain): %.1f us per step" % (e0.elapsed_time(e1) / 20 / 15 * 1e3))

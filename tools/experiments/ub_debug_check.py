"""CPU recomputation for tools/ub_debug.bin: python tools/ub_debug_check.py STOP (reads gpurun_out/ub/*.bin)."""
import sys, numpy as np, torch, torch.nn.functional as F
stop = int(sys.argv[1])
d = "gpurun_out/ub/"
ld = lambda n, *s: torch.from_numpy(np.fromfile(d + n, dtype=np.float32).reshape(*s))
x1 = ld("x1.bin", 1, 32, 104, 8)
cin = [32, 64, 64, 128, 128, 128, 64]; cout = [64, 64, 128, 128, 64, 64, 64]
w = [ld(f"w{l}.bin", *((128, 64, 2, 2) if l == 4 else (cout[l], cin[l], 3, 3))) for l in range(7)]
act = lambda t: F.leaky_relu(F.instance_norm(t, eps=1e-5), 0.2)
p1 = F.avg_pool2d(act(x1), 2)
a0 = act(F.conv2d(p1, w[0], padding=1)); a1 = act(F.conv2d(a0, w[1], padding=1))
p2 = F.avg_pool2d(a1, 2)
a2 = act(F.conv2d(p2, w[2], padding=1)); a3 = act(F.conv2d(a2, w[3], padding=1))
a4 = act(F.conv_transpose2d(a3, w[4], stride=2))
a5 = act(F.conv2d(torch.cat([a4, a1], 1), w[5], padding=1))
y = F.conv2d(a5, w[6], padding=1)
PS2, PS3, BUF = 368, 176, 23552
dump = ld("dump.bin", BUF)
def lay2(t):  # (64, 52, 4) -> buffer layout
    b = torch.zeros(BUF); c = t.shape[0]
    v = b[:c * PS2].view(c, PS2)[:, :54 * 6].view(c, 54, 6); v[:, 1:53, 0:4] = t; return b
def lay3(t):
    b = torch.zeros(BUF); c = t.shape[0]
    v = b[:c * PS3].view(c, PS3)[:, :34 * 4].view(c, 34, 4); v[:, 1:27, 0:2] = t; return b
want = {0: lay2(a0[0]), 1: lay3(p2[0]), 2: lay3(a2[0]), 3: lay3(a3[0]), 4: lay2(a4[0]), 5: lay2(a5[0])}
if stop in want:
    wv = want[stop]
    err = (dump - wv).abs()
    print("stop", stop, "finite", bool(torch.isfinite(dump).all()), "max abs err", float(err.max()), "ref absmax", float(wv.abs().max()),
          "nonzero where ref is zero:", int(((wv == 0) & (dump != 0)).sum()))
    if float(err.max()) > 1e-3:
        idx = int(err.argmax()); print(" worst at", idx, "got", float(dump[idx]), "want", float(wv[idx]))
else:
    got = ld("y.bin", 64, 52, 4); sk = ld("skip.bin", 64, 52, 4); py = ld("py.bin", 64, 3)
    print("final y max abs err", float((got - y[0]).abs().max()), "ref absmax", float(y.abs().max()), "skip err", float((sk - a1[0]).abs().max()))
    print("stats mean err", float((py[:, 1] - y[0].mean(dim=(1, 2))).abs().max()), "M2 rel err", float(((py[:, 2] - y[0].var(dim=(1, 2), unbiased=False) * 208).abs() / (y[0].var(dim=(1, 2), unbiased=False) * 208)).max()))

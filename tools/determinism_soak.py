"""Race shake-out: N eager forwards of each BASELINE configuration on the same input must be bit-identical (the kernels have no atomics and
fixed reduction orders, so any difference is a missing barrier / a read of unwritten LDS), also with other work in flight on a second stream.
  python tools/determinism_soak.py [forwards per configuration, default 60]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")]
import torch
import reconstruction.models as M
from cine_hip import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
nets = {2: lambda: M.VarNet(6, 8, 3, 16, 3, "XF"), 3: lambda: M.XPDNet(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT"),
        4: lambda: M.CineNet(6, 6, 16, 3, "3D"), 5: lambda: M.VarNet_RNN(5, 8, 3, 16)}
bad = 0
for cfg, make in nets.items():
    ex = synth.make_cine_slice(15, 15, 200, 200, accel={2: 4, 3: 8, 4: 6, 5: 8}[cfg], seed=cfg)
    net = make(); synth.fill_parameters_(net, cfg); net = net.to(dev).eval()
    mk, mask = ex["masked_kspace"].to(dev), ex["mask"].to(dev)
    extra = (ex["sens_maps"].to(dev),) if cfg == 4 else ()
    side = torch.cuda.Stream()
    noise = torch.randn(64, 16, 208, 16, device=dev)
    with torch.no_grad():
        ref = net(mk, mask, *extra).clone()
        diff = 0
        for i in range(n):
            if i % 2:                                   # unrelated traffic on another stream: different co-residency for the kernels under test
                with torch.cuda.stream(side):
                    for _ in range(8): noise = torch.nn.functional.leaky_relu(noise * 1.0001, 0.2)
            out = net(mk, mask, *extra)
            if not torch.equal(out, ref):
                diff += 1
        torch.cuda.synchronize()
    print(f"cfg {cfg}: {n} forwards, {diff} differ from the first", flush=True)
    bad += diff
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Probe: throughput with S independent slices in flight on S streams (hipGraph per stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")]
import torch
import reconstruction.models as M
from cine_hip import synth

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
use_graph = (sys.argv[3] != "eager") if len(sys.argv) > 3 else True
net = M.VarNet(6, 8, 3, 16, 3, "XF").eval(); synth.fill_parameters_(net, 1); net.to(dev)
exs = [synth.make_cine_slice(15, 15, 200, 200, accel=4, seed=s) for s in range(S)]
mks = [e["masked_kspace"].to(dev) for e in exs]; masks = [e["mask"].to(dev) for e in exs]
acs = net.sens_net.acs_window(masks[0])
net(mks[0], masks[0], acs=acs); torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(S)]
graphs, outs = [], []
for s in range(S):
    with torch.cuda.stream(streams[s]):
        net(mks[s], masks[s], acs=acs)
    torch.cuda.synchronize()
    if use_graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[s]):
            o = net(mks[s], masks[s], acs=acs)
        graphs.append(g); outs.append(o)
torch.cuda.synchronize()
def step():
    for s in range(S):
        with torch.cuda.stream(streams[s]):
            if use_graph: graphs[s].replay()
            else: net(mks[s], masks[s], acs=acs)
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"streams={S} graph={use_graph}: {S * steps / dt:.1f} slices/s  ({dt / steps * 1e3:.2f} ms per round of {S})")

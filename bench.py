#!/usr/bin/env python3
"""bench.py -- cine slices/sec of the XF-VarNet hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full forward of BASELINE.json configs[1] on one synthetic cine slice
already resident in HBM: XF-VarNet, 6 cascades, 15 coils x 15 frames x 200x200, R=4
(sens-map network + 6 x [sens_reduce, x-f/y-f U-Nets, sens_expand + soft DC] + final
magnitude), fp32 end to end through the hand-written HIP kernels.  Slices are independent, so a GPU
keeps `--inflight` (default 3) of them in flight, each replaying its own hipGraph on its own stream:
their memory-bound and MFMA-bound phases interleave.  K steps = K slices in total.

N > 1: launched by torch.distributed.run, one process per GPU (RCCL = backend "nccl").
Slices are independent, so ranks shard them with no data-path collective (weak scaling:
K slices per GPU); the only exchange is one all-gather of the (K, 15, 200, 200) outputs for
volume assembly, inside the timed region.  Time = max over ranks, value = N*K / time.

Rank 0 prints ONE JSON line with `roofline` (the dominant kernel family, measured live with
hipEvents on the launch stream via cine_profile_*), `roofline_fft_dc` (the HBM-bound FFT+DC
family) and `cpu_baseline` (the CPU oracle timed on this host's cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# ---- workload (BASELINE.json configs[1]) and its algorithmic work
CFG = dict(cascades=6, sens_chans=8, sens_pools=3, chans=16, pools=3, dyn="XF",
           frames=15, coils=15, h=200, w=200, accel=4)
MB = 1e6
K_MB, I_MB, S_MB = 72.0, 4.8, 4.8                      # k-space, image, sens maps (fp32 complex)
FFT_DC_BYTES_PER_SLICE = (6 * ((K_MB + S_MB + I_MB) + (I_MB + S_MB + K_MB + K_MB)) + (K_MB + S_MB + 2.4)) * MB


def unet_conv3_macs(chans, pools, in_ch, h, w):
    """MACs of the 3x3 convolutions of one U-Net pass on one (in_ch, h, w) plane (unet.py:51-71)."""
    macs, ch, hh, ww, cin = 0, chans, h, w, in_ch
    dims = []
    for d in range(pools):
        macs += hh * ww * 9 * (cin * ch + ch * ch)
        dims.append((hh, ww, ch))
        cin, ch, hh, ww = ch, ch * 2, hh // 2, ww // 2
    macs += hh * ww * 9 * (cin * ch + ch * ch)                 # bottleneck
    for (hh, ww, c) in reversed(dims):
        macs += hh * ww * 9 * (2 * c * c + c * c)              # conv on cat([up, skip]) + second conv
    return macs


def pad16(n):
    return ((n - 1) | 15) + 1


# FLOPs the conv3x3 MFMA kernel executes per slice: 6 cascades x (200 x-f + 200 y-f planes of
# 2 x 208 x 16) through U-Net(16, 3), plus the sens-map U-Net(8, 3) on 15 coil images of 208 x 208.
CONV3_FLOP_PER_SLICE = 2.0 * (
    CFG["cascades"] * (CFG["h"] + CFG["w"]) * unet_conv3_macs(CFG["chans"], CFG["pools"], 2, pad16(CFG["w"]), pad16(CFG["frames"]))
    + CFG["coils"] * unet_conv3_macs(CFG["sens_chans"], CFG["sens_pools"], 2, pad16(CFG["h"]), pad16(CFG["w"])))
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--inflight", type=int, default=3,
                    help="independent slices in flight per GPU, each on its own HIP stream (its own hipGraph)")
    ap.add_argument("--batch", type=int, default=1,
                    help="slices per step: one forward over a (batch, t, coil, h, w, 2) k-space batch, the reference's batch axis")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-forwards", type=int, default=1)
    return ap.parse_args()


def build_model(dev):
    import reconstruction.models as M
    from cine_hip import synth
    net = M.VarNet(CFG["cascades"], CFG["sens_chans"], CFG["sens_pools"], CFG["chans"], CFG["pools"], CFG["dyn"]).eval()
    synth.fill_parameters_(net, 1)
    return net.to(dev)


def profile_families(fn, iters=3):
    """Per-kernel-family device time of `fn`, via hipEvent pairs on the launch stream."""
    from cine_hip._lib import lib
    L = lib()
    nf = L.cine_profile_families()
    ms = (ctypes.c_double * nf)()
    cnt = (ctypes.c_long * nf)()
    torch.cuda.synchronize()
    L.cine_profile_begin()
    for _ in range(iters):
        fn()
    L.cine_profile_end(ms, cnt, nf)
    return {L.cine_profile_family_name(i).decode(): (ms[i] / iters, cnt[i] // iters) for i in range(nf)}


def cpu_baseline(ex, forwards):
    """The CPU oracle (oracle/, a restatement of the reference's PyTorch CPU path, pinned to the
    reference's outputs by tests/test_oracle_golden.py) on this host's cores: bounded sample of
    1 warm-up + `forwards` timed forwards of the same cfg-2 slice."""
    from oracle import varnet_ref as V
    from cine_hip import synth
    net = V.VarNet(CFG["cascades"], CFG["sens_chans"], CFG["sens_pools"], CFG["chans"], CFG["pools"], CFG["dyn"]).eval()
    synth.fill_parameters_(net, 1)
    cores = torch.get_num_threads()
    with torch.no_grad():
        net(ex["masked_kspace"], ex["mask"])
        t0 = time.perf_counter()
        for _ in range(forwards):
            out = net(ex["masked_kspace"], ex["mask"])
        dt = (time.perf_counter() - t0) / forwards
    return out, {"value": 1.0 / dt, "unit": "cine slices/sec", "cores": cores, "kind": "port",
                 "sample": f"1 warm-up + {forwards} timed forwards of the same cfg-2 slice "
                           f"(torch CPU fp32, {cores} threads), {dt:.2f} s/slice"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from cine_hip import synth, shard
    S = max(1, min(args.inflight, args.steps))
    B = max(1, args.batch)
    exs = [synth.make_cine_slice(CFG["frames"], CFG["coils"], CFG["h"], CFG["w"], accel=CFG["accel"], seed=rank * B + i)
           for i in range(B)]
    ex = exs[0]
    mk = torch.cat([e["masked_kspace"] for e in exs]).to(dev)
    mask = torch.cat([e["mask"] for e in exs]).to(dev)
    net = build_model(dev)
    acs = net.sens_net.acs_window(mask)          # host read-back of the 1-D mask, outside capture
    # every in-flight slice has its own input copy, stream and (when captured) graph; weights are shared
    mks = [mk] + [mk.clone() for _ in range(S - 1)]

    def forward(i=0):
        return net(mks[i], mask, acs=acs)

    out = forward()                               # also packs the weights
    torch.cuda.synchronize()

    streams = [torch.cuda.Stream() for _ in range(S)]
    graphs, gouts = [], []
    use_graph = not args.no_graph
    if use_graph:
        try:
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    forward(i)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[i]):
                    o = forward(i)
                graphs.append(g); gouts.append(o)
        except Exception as e:                                        # pragma: no cover
            if rank == 0:
                print(f"# hipGraph capture failed ({e}); running eagerly", file=sys.stderr)
            use_graph = False
    torch.cuda.synchronize()

    outs = torch.empty((args.steps * B,) + tuple(out.shape[1:]), device=dev)   # this rank's slices

    def run(nsteps, keep):
        """nsteps slices, round-robin over the S streams; each stream is an in-order queue."""
        for k in range(nsteps):
            i = k % S
            with torch.cuda.stream(streams[i]):
                o = gouts[i] if use_graph else None
                if use_graph:
                    graphs[i].replay()
                else:
                    o = forward(i)
                if keep:
                    outs[k * B:(k + 1) * B].copy_(o)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)

    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    run(args.warmup, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, True)
    volume = shard.assemble_volume(outs, world * args.steps * B)      # one all-gather over xGMI (no-op at N=1)
    assert volume.shape[0] == world * args.steps * B
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(device_ids=[local])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- rank 0: per-family device time (eager launches, hipEvents on the launch stream)
    fam = profile_families(forward)
    fam = {k: (v[0] / B, v[1]) for k, v in fam.items()}            # per slice
    conv_ms, conv_n = fam["conv3x3_mfma"]
    fft_ms = fam["fft_col_pass"][0] + fam["fft_row_pass"][0]
    roofline = {"bound": "mfma", "kernel": "cine::conv_mfma_kernel<8, CT, WM, WN, MT, TW, 9> (the 3x3 instantiations; 14 x 6 + 14 launches of one slice)",
                "achieved": CONV3_FLOP_PER_SLICE / (conv_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "traffic": None, "launches_per_slice": conv_n, "ms_per_slice": conv_ms,
                "avg_launch_us": conv_ms * 1e3 / max(conv_n, 1)}
    roofline["frac"] = roofline["achieved"] / roofline["peak"]
    # HBM-side bytes per launch of that kernel family: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    # command, summarised by tools/pmc_traffic.py into profiles/ (PMC collection cannot run inside the bench itself)
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tr = json.load(f)["families"]
        roofline["traffic"] = tr["conv3x3_mfma"]["hbm_MB_per_launch"] * 1e6
        roofline["traffic_unit"] = "HBM bytes per launch (mean over the 98 launches of a slice), PMC, profiles/r01_pmc_traffic.json"
    roof_fft = {"bound": "hbm", "kernel": "cine::col200_kernel + row200_reduce_kernel + row200_expand_kernel (sens_reduce x7, sens_expand+DC x6)",
                "achieved": FFT_DC_BYTES_PER_SLICE / (fft_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "traffic": None, "ms_per_slice": fft_ms}
    roof_fft["frac"] = roof_fft["achieved"] / roof_fft["peak"]
    if os.path.exists(tpath):
        roof_fft["traffic"] = (tr["fft_col_pass"]["hbm_MB_per_slice"] + tr["fft_row_pass"]["hbm_MB_per_slice"]) * 1e6
        roof_fft["traffic_unit"] = "HBM bytes per slice over all FFT passes, PMC"

    line = {
        "metric": "cine slices/sec, XF-VarNet R=4 15-coil 200x200x15t", "value": world * args.steps * B / dt,
        "unit": "cine slices/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: XF-VarNet, 6 cascades, 15 coils x 15 frames x 200x200, "
                               "R=4 Gaussian-density Cartesian mask, sens net 8ch/3 pools, U-Net 16ch/3 pools; "
                               f"{B} slice(s) per step (k-space batch axis), seeded random-init weights",
                   "slices_per_step": B,
                   "launch": ("hipGraph replay" if use_graph else "eager") + f", {S} independent steps in flight on {S} HIP streams",
                   "parallelism": f"slice-sharded x{world}, one all-gather for volume assembly"},
        "roofline": roofline, "roofline_fft_dc": roof_fft,
        "kernel_ms_per_slice": {k: round(v[0], 4) for k, v in fam.items()},
    }
    if not args.no_cpu_baseline and world == 1:        # the host-core baseline is timed on rank 0 of the 1-GPU run only
        ref_out, cb = cpu_baseline(ex, args.cpu_forwards)
        line["cpu_baseline"] = cb
        err = float((out[:1].cpu() - ref_out).abs().max() / ref_out.abs().max())
        line["parity_max_rel_err_vs_cpu_oracle"] = err
        from reconstruction.utils import evaluate
        tgt = ex["target"][0].numpy()
        line["parity_d_ssim_vs_cpu_oracle"] = abs(float(evaluate.ssim(tgt, out[0].cpu().numpy())) - float(evaluate.ssim(tgt, ref_out[0].numpy())))
        line["parity_nmse_vs_cpu_oracle"] = float(evaluate.nmse(ref_out[0].numpy(), out[0].cpu().numpy()))
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- cine slices/sec of the reconstruction hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {2,3,4,5}]

One "step" = one full forward of a BASELINE.json configuration on one synthetic cine slice already resident in HBM.
Default (and the driver's line) is configs[1]: XF-VarNet, 6 cascades, 15 coils x 15 frames x 200x200, R=4 (sens-map network
+ 6 x [x-f / y-f U-Nets, image-space data consistency] + magnitude), fp32 end to end through the hand-written HIP kernels.
Slices are independent, so a GPU keeps `--inflight` (default: 6..12, a divisor of K) DIFFERENT slices in flight, each replaying its own hipGraph
on its own stream: their memory-bound and MFMA-bound phases interleave.  K steps = K slices in total.

N > 1: one process per GPU over RCCL (backend "nccl").  Either the caller starts the ranks (torch.distributed.run: RANK /
LOCAL_RANK / WORLD_SIZE in the environment) or, when WORLD_SIZE is unset, this script starts `--gpus` ranks itself as child
processes BEFORE touching the GPU and relays rank 0's line.  Slices shard over ranks with no data-path collective (weak
scaling: K slices per GPU); the only exchange is one all-gather of the (K, t, h, w) outputs for volume assembly, inside the
timed region.  Time = max over ranks, value = N*K / time.

Rank 0 prints ONE JSON line.  `roofline` = the dominant kernel family (3x3 conv on fp32 MFMA): `frac` from the kernels'
ISOLATED duration (eager launches, one slice in flight, hipEvent pairs on the launch stream -- agrees with
profiles/r02_rocprofv3_kernel_stats_isolated.csv); `in_flight` = the same FLOPs over the driver-timed wall time of the
in-flight graph replay (a lower bound on the family's rate in the timed mode, see profiles/r02_..._inflight.csv).
`roofline_fft_dc` = the HBM-bound FFT + data-consistency family against SURVEY 8(d)'s algorithmic bytes.  `cpu_baseline` =
the CPU oracle on this host's cores (thread sweep, then >= 3 forwards at the best setting).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "deep-cine-cardiac-mri_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# Each slice in flight replays its graph on its own HIP stream; the runtime maps streams onto 4 hardware queues unless told
# otherwise, and with more slices than queues the graphs serialise (measured at cfg 2: 4 queues 143-147 slices/s, 16 queues
# 150-152).  Must be set before the first HIP call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MB = 1e6
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3
FRAMES, COILS, H, W = 15, 15, 200, 200


def pad16(n):
    return ((n - 1) | 15) + 1


def unet_conv3_macs(chans, pools, in_ch, h, w):
    """MACs of the 3x3 convolutions of one U-Net pass on one (in_ch, h, w) plane (reference denoisers/unet.py:51-71)."""
    macs, ch, hh, ww, cin = 0, chans, h, w, in_ch
    dims = []
    for _ in range(pools):
        macs += hh * ww * 9 * (cin * ch + ch * ch)
        dims.append((hh, ww, ch))
        cin, ch, hh, ww = ch, ch * 2, hh // 2, ww // 2
    macs += hh * ww * 9 * (cin * ch + ch * ch)                 # bottleneck
    for (hh, ww, c) in reversed(dims):
        macs += hh * ww * 9 * (2 * c * c + c * c)              # conv on cat([up, skip]) + second conv
    return macs


# ---- workloads: BASELINE.json configs[1..4] (SURVEY.md section 8d gives the widths and the conv GFLOP per forward)
def _cfg2():
    import reconstruction.models as M
    from oracle import varnet_ref as V
    flop = 2.0 * (6 * (H + W) * unet_conv3_macs(16, 3, 2, pad16(W), pad16(FRAMES)) + COILS * unet_conv3_macs(8, 3, 2, pad16(H), pad16(W)))
    k_mb, i_mb, s_mb = 72.0, 4.8, 4.8
    fft_bytes = (6 * ((k_mb + s_mb + i_mb) + (i_mb + s_mb + k_mb + k_mb)) + (k_mb + s_mb + 2.4)) * MB
    return dict(name="BASELINE.json configs[1]: XF-VarNet, 6 cascades, 15 coils x 15 frames x 200x200, R=4 Gaussian-density "
                     "Cartesian mask, sens net 8ch/3 pools, U-Net 16ch/3 pools",
                metric="cine slices/sec, XF-VarNet R=4 15-coil 200x200x15t", accel=4, noise=0.0, wseed=1, keep=("lambda",),
                hip=lambda: M.VarNet(6, 8, 3, 16, 3, "XF"), ref=lambda: V.VarNet(6, 8, 3, 16, 3, "XF"), needs_sens=False,
                conv_flop=flop, conv_kernel="cine::conv_mfma_kernel<8, CT, WM, WN, MT, TW, 9, 0> (the 3x3 instantiations)",
                fft_bytes=fft_bytes)


def _cfg3():
    import reconstruction.models as M
    from oracle import xpdnet_ref as X
    kw = dict(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT")
    return dict(name="BASELINE.json configs[2]: XT-XPDNet, MWCNN regulariser (script defaults), 10 cascades, n_primal 5, "
                     "15 coils x 15 frames x 200x200, R=8", metric="cine slices/sec, XT-XPDNet R=8 15-coil 200x200x15t",
                accel=8, noise=0.01, wseed=6, keep=(), hip=lambda: M.XPDNet(**kw), ref=lambda: X.XPDNet(**kw), needs_sens=False,
                conv_flop=416.2e9, conv_kernel="cine::conv_mfma_kernel<8, ..., 9> (MWCNN 3x3 convs with Haar DWT / IWT on load)",
                fft_bytes=None)


def _cfg4():
    import reconstruction.models as M
    from oracle import cinenet_ref as C
    return dict(name="BASELINE.json configs[3]: 3D CineNet, 6 cascades, CG 6, U-Net3D 16ch/3 pools, 15 coils x 15 frames x 200x200, R=6",
                metric="cine slices/sec, 3D CineNet R=6 15-coil 200x200x15t", accel=6, noise=0.0, wseed=7, keep=("lambda",),
                hip=lambda: M.CineNet(6, 6, 16, 3, "3D"), ref=lambda: C.CineNet(6, 6, 16, 3, "3D"), needs_sens=True,
                conv_flop=365.2e9, conv_kernel="cine::conv_mfma_kernel<8, ..., 9, 1> (3x3x3 convs as three 3x3 passes) + <4, ..., 27, 0> (small / narrow levels)", fft_bytes=None)


def _cfg5():
    import reconstruction.models as M
    from oracle import recurrent_ref as R
    return dict(name="BASELINE.json configs[4]: CRNN-VarNet, 5 cascades, sens net 8ch/3 pools, 16 hidden channels, "
                     "15 coils x 15 frames x 200x200, R=8", metric="cine slices/sec, CRNN-VarNet R=8 15-coil 200x200x15t",
                accel=8, noise=0.0, wseed=9, keep=("lambda",), hip=lambda: M.VarNet_RNN(5, 8, 3, 16),
                ref=lambda: R.VarNet_RNN(5, 8, 3, 16), needs_sens=False, conv_flop=155.0e9,
                conv_kernel="cine::conv_mfma_kernel<8, ..., 9> (CRNN cells: summed-input 3x3 convs with bias/addend/ReLU epilogue)",
                fft_bytes=None)


CONFIGS = {2: _cfg2, 3: _cfg3, 4: _cfg4, 5: _cfg5}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configs[N-1] (default 2 = the metric's config)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--inflight", type=int, default=0,
                    help="independent slices in flight per GPU, each on its own HIP stream (its own hipGraph); 0 = auto: the "
                         "largest count in 6..12 that divides --steps (whole rounds of streams: no half-empty last round), else 8")
    ap.add_argument("--batch", type=int, default=1,
                    help="slices per step: one forward over a (batch, t, coil, h, w, 2) k-space batch, the reference's batch axis")
    ap.add_argument("--repeats", type=int, default=4, help="extra timed K-step regions after the contract one (median reported beside value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-forwards", type=int, default=3)
    ap.add_argument("--cpu-threads", type=str, default="sweep", help="'sweep' (8,16,32,64,physical) or a number")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="TEST ONLY (tests/test_distributed_cpu.py): run the launcher, sharding, timed region and all-gather on "
                         "gloo/CPU with a stand-in for the forward; the line is marked invalid and measures nothing")
    return ap.parse_args()


# ------------------------------------------------------------------ multi-GPU launcher
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """WORLD_SIZE unset and --gpus N > 1: start N ranks as child processes (this process has not touched the GPU and never
    does), relay their output, exit with their code.  No exec: a re-exec of a GPU process takes the box down on this pool."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------ measurement helpers
def profile_families(fn, iters=3):
    """Per-kernel-family device time of `fn`, via hipEvent pairs on the launch stream (cine_profile_*)."""
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


def host_cpu():
    model, phys = "unknown", None
    try:
        cores = set()
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                pid = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":", 1)[1].strip()
                cores.add((pid, cid))
        phys = len(cores) or None
    except OSError:
        pass
    return model, phys or os.cpu_count() or 1


def cpu_baseline(cfg, ex, forwards, threads_arg):
    """The CPU oracle (oracle/: restatement of the reference's PyTorch CPU path, pinned to the reference's outputs by
    tests/test_oracle_golden.py / test_synth_golden.py) on this host's cores.  Bounded sample: one warm-up, one timed
    forward per thread setting of the sweep, then `forwards` timed forwards at the best setting."""
    from cine_hip import synth
    net = cfg["ref"]().eval()
    synth.fill_parameters_(net, cfg["wseed"], keep=cfg["keep"])
    model, phys = host_cpu()
    logical = os.cpu_count() or phys
    args = (ex["masked_kspace"], ex["mask"]) + ((ex["sens_maps"],) if cfg["needs_sens"] else ())
    if threads_arg == "sweep":
        cand = sorted({n for n in (8, 16, 32, 64, phys) if n <= logical})
    else:
        cand = [int(threads_arg)]
    sweep = {}
    with torch.no_grad():
        torch.set_num_threads(cand[0])
        out = net(*args)                                          # warm-up (allocator, MKL-DNN primitives)
        for n in cand:
            torch.set_num_threads(n)
            t0 = time.perf_counter()
            out = net(*args)
            sweep[n] = time.perf_counter() - t0
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        t0 = time.perf_counter()
        for _ in range(forwards):
            out = net(*args)
        dt = (time.perf_counter() - t0) / forwards
    return out, {"value": 1.0 / dt, "unit": "cine slices/sec", "cores": best, "kind": "port", "cpu_model": model,
                 "physical_cores": phys, "logical_cpus": logical,
                 "thread_sweep_s_per_slice": {str(k): round(v, 3) for k, v in sweep.items()},
                 "sample": f"1 warm-up, 1 forward per thread setting {cand}, then {forwards} timed forwards of the same slice at "
                           f"{best} threads (torch CPU fp32): {dt:.2f} s/slice"}


def timed_steps(run, nsteps, batch, outs, world, sync, barrier, device):
    """The timed region of the contract: barrier + sync, `nsteps` steps, volume assembly, sync + barrier; max over ranks."""
    from cine_hip import shard
    sync()
    if world > 1:
        barrier()
    sync()
    t0 = time.perf_counter()
    run(nsteps)
    volume = shard.assemble_volume(outs, world * nsteps * batch)      # one all-gather (RCCL over xGMI; no-op at N=1)
    assert volume.shape[0] == world * nsteps * batch
    sync()
    if world > 1:
        barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    return dt, volume


def selftest_cpu(args, world, rank):
    """tests/test_distributed_cpu.py: the launcher, rank bookkeeping, slice sharding, timed region and volume assembly on
    gloo/CPU.  The forward is a stand-in (a fixed function of the slice id); nothing is measured."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    K = args.steps
    outs = torch.empty(K, 3, 8, 6)

    def standin(slice_id):
        return torch.rand(3, 8, 6, generator=torch.Generator().manual_seed(1000 + slice_id))

    def run(nsteps):
        for k in range(nsteps):
            outs[k] = standin(rank + k * world)                       # rank r owns slices r, r + N, ...

    run(args.warmup)
    dt, volume = timed_steps(run, K, 1, outs, world, lambda: None, dist.barrier, torch.device("cpu"))
    ok = bool(torch.equal(volume, torch.stack([standin(i) for i in range(world * K)])))
    if rank == 0:
        print(json.dumps({"metric": "selftest (no measurement)", "value": None, "valid": False, "n_gpus": world, "rccl_ranks": world,
                          "steps": K, "warmup": args.warmup, "volume_ok": ok, "volume_slices": int(volume.shape[0]),
                          "data": "cpu stand-in for the forward (test only)", "timed_region_s": dt}))
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs")
    if args.selftest_cpu:
        return selftest_cpu(args, world, rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from cine_hip import synth
    cfg = CONFIGS[args.config]()
    if args.inflight > 0:
        S = max(1, min(args.inflight, args.steps))
    else:
        S = next((c for c in range(12, 5, -1) if args.steps % c == 0), 8) if args.steps >= 6 else max(1, args.steps)
    B = max(1, args.batch)
    # S DIFFERENT slices per rank (x B on the batch axis): seeds rank * S * B + ...
    exs = [[synth.make_cine_slice(FRAMES, COILS, H, W, accel=cfg["accel"], seed=(rank * S + i) * B + j, noise_std=cfg["noise"])
            for j in range(B)] for i in range(S)]
    host_mk = [torch.cat([e["masked_kspace"] for e in row]).pin_memory() for row in exs]
    mks = [h.to(dev, non_blocking=True) for h in host_mk]
    masks = [torch.cat([e["mask"] for e in row]).to(dev) for row in exs]
    senss = [torch.cat([e["sens_maps"] for e in row]).to(dev) for row in exs] if cfg["needs_sens"] else None
    net = cfg["hip"]().eval()
    synth.fill_parameters_(net, cfg["wseed"], keep=cfg["keep"])
    net = net.to(dev)
    from reconstruction.models.varnet import SensitivityModel
    acss = [SensitivityModel.acs_window(m) for m in masks]                       # host read-back of the 1-D mask, outside capture

    def forward(i=0):
        if cfg["needs_sens"]:
            return net(mks[i], masks[i], senss[i])
        return net(mks[i], masks[i], acs=acss[i])

    out = forward()                               # also packs the weights
    torch.cuda.synchronize()

    streams = [torch.cuda.Stream() for _ in range(S)]
    graphs, gouts = [], []
    use_graph = not args.no_graph
    if use_graph:
        try:
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    forward(i)                     # warm this stream's caches outside capture
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[i]):
                    o = forward(i)
                graphs.append(g); gouts.append(o)
        except Exception as e:                                        # pragma: no cover
            if rank == 0:
                print(f"# hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            use_graph = False
            graphs, gouts = [], []
    torch.cuda.synchronize()

    outs = torch.empty((args.steps * B,) + tuple(out.shape[1:]), device=dev)   # this rank's slices

    def run(nsteps, keep, h2d=False):
        """nsteps slices, round-robin over the S streams; each stream is an in-order queue."""
        for k in range(nsteps):
            i = k % S
            with torch.cuda.stream(streams[i]):
                if h2d:
                    mks[i].copy_(host_mk[i], non_blocking=True)      # pinned host buffer -> HBM, ahead of this slice's replay
                if use_graph:
                    graphs[i].replay()
                    o = gouts[i]
                else:
                    o = forward(i)
                if keep:
                    outs[k * B:(k + 1) * B].copy_(o)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)

    def timed(nsteps, h2d=False):
        return timed_steps(lambda n: run(n, True, h2d), nsteps, B, outs, world, torch.cuda.synchronize,
                           lambda: dist.barrier(device_ids=[local]), dev)[0]

    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    run(args.warmup, False)
    dt = timed(args.steps)                                            # THE timed region: exactly K steps
    extra = sorted(timed(args.steps) for _ in range(max(0, args.repeats)))
    dt_h2d = timed(args.steps, h2d=True)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- rank 0: per-family device time, ISOLATED (eager launches, one slice in flight, hipEvents on the launch stream)
    fam = profile_families(forward)
    fam = {k: (v[0] / B, v[1]) for k, v in fam.items()}            # per slice
    conv_ms, conv_n = fam["conv3x3_mfma"]
    fft_ms = fam["fft_col_pass"][0] + fam["fft_row_pass"][0]
    slices = world * args.steps * B
    roofline = {"bound": "mfma", "kernel": cfg["conv_kernel"], "mode": "isolated: eager launches, one slice in flight, hipEvent pairs per launch",
                "achieved": cfg["conv_flop"] / (conv_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "traffic": None, "launches_per_slice": conv_n, "ms_per_slice": conv_ms, "avg_launch_us": conv_ms * 1e3 / max(conv_n, 1),
                "flop_per_slice": cfg["conv_flop"]}
    roofline["frac"] = roofline["achieved"] / roofline["peak"]
    # the same FLOPs over the driver-timed wall time of the in-flight graph replay: whole-chip fp32-MFMA utilisation of the
    # timed mode, a lower bound on the conv family's own rate there (its kernels overlap other families' on other streams)
    roofline["in_flight"] = {"tflops_lower_bound": cfg["conv_flop"] * slices / world / dt / 1e12,
                             "frac_lower_bound": cfg["conv_flop"] * slices / world / dt / 1e12 / MFMA_F32_PEAK_TFLOPS,
                             "mode": ("hipGraph replay" if use_graph else "eager") + f", {S} slices in flight (the timed region)"}
    # HBM-side bytes per launch of that kernel family: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command,
    # summarised by tools/pmc_traffic.py into profiles/ (PMC collection cannot run inside the bench itself)
    tr, tr_src = None, None
    for name in ("r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if args.config == 2 and os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            tr, tr_src = tj["families"], f"profiles/{name}" + (f" @ {tj['commit']}" if "commit" in tj else "")
            break
    if tr and "conv3x3_mfma" in tr:
        roofline["traffic"] = tr["conv3x3_mfma"]["hbm_MB_per_launch"] * 1e6
        roofline["traffic_unit"] = f"HBM bytes per launch (mean over the launches of a slice), PMC FETCH_SIZE x2 + WRITE_SIZE, {tr_src}"
    line = {
        "metric": cfg["metric"], "value": slices / dt, "unit": "cine slices/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": cfg["name"] + f"; {B} slice(s) per step (k-space batch axis), seeded random-init weights, "
                               f"{S} different slices per rank",
                   "slices_per_step": B,
                   "launch": ("hipGraph replay" if use_graph else "eager") + f", {S} independent steps in flight on {S} HIP streams, "
                             f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}",
                   "parallelism": f"slice-sharded x{world}, one all-gather for volume assembly"},
        "rccl_ranks": world,
        "timed_region_s": dt,
        "repeat_values": [slices / d for d in extra], "repeat_median_value": (slices / extra[len(extra) // 2]) if extra else None,
        "value_with_h2d": slices / dt_h2d,
        "value_with_h2d_note": "same K steps with each slice's 72 MB k-space copied pinned-host -> HBM on its stream ahead of the "
                               "replay (what run_inference.py:53-61 times); overlaps the other streams' compute",
        "roofline": roofline,
        "kernel_ms_per_slice": {k: round(v[0], 4) for k, v in fam.items()},
    }
    if cfg["fft_bytes"]:
        roof_fft = {"bound": "hbm", "kernel": "cine::imgdc200_kernel + imgdc_sum_kernel (x6: sens_expand + FFT2 + DC + IFFT2 + sens_reduce on the "
                                              "coil-combined image), col200_kernel + row200_reduce_kernel (first reduce, zero-filled term, sens prologue)",
                    "mode": roofline["mode"], "achieved": cfg["fft_bytes"] / (fft_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "algorithmic_bytes": cfg["fft_bytes"], "traffic": None, "ms_per_slice": fft_ms,
                    "note": "achieved = SURVEY 8(d) bytes at the reference's module boundaries / kernel time; the image-space chain "
                            "moves far fewer bytes than that (traffic), so this is the step's speed in the survey's units, not a bandwidth"}
        roof_fft["frac"] = roof_fft["achieved"] / roof_fft["peak"]
        if tr and "fft_col_pass" in tr:
            roof_fft["traffic"] = (tr.get("fft_col_pass", {}).get("hbm_MB_per_slice", 0.0) + tr.get("fft_row_pass", {}).get("hbm_MB_per_slice", 0.0)) * 1e6 or None
            roof_fft["traffic_unit"] = f"HBM bytes per slice over all FFT / DC passes, PMC, {tr_src}"
        line["roofline_fft_dc"] = roof_fft
    # ---- parity of a GRAPH-REPLAYED output (stream 0's slice) and the CPU baseline
    if use_graph:
        graphs[0].replay()
        torch.cuda.synchronize()
        chk = gouts[0][:1].clone()
    else:
        chk = forward(0)[:1].clone()
    ex0 = exs[0][0]
    tgt_d = ex0["target"].to(dev)
    try:
        from cine_hip import ops
        m = ops.image_metrics(tgt_d[0], chk[0])
        line["ssim"] = {"value": float(m["ssim"]), "nmse": float(m["nmse"]), "psnr": float(m["psnr"]),
                        "note": "device kernel (cine_image_metrics), graph-replayed output vs the synthetic target; untrained weights"}
    except (ImportError, AttributeError):
        pass
    if not args.no_cpu_baseline and world == 1:        # the host-core baseline is timed on rank 0 of the 1-GPU run only
        ref_out, cb = cpu_baseline(cfg, ex0, args.cpu_forwards, args.cpu_threads)
        line["cpu_baseline"] = cb
        from reconstruction.utils import evaluate
        got = chk.cpu()
        line["parity_output"] = "hipGraph replay" if use_graph else "eager"
        line["parity_max_rel_err_vs_cpu_oracle"] = float((got - ref_out).abs().max() / ref_out.abs().max())
        tgt = ex0["target"][0].numpy()
        line["parity_d_ssim_vs_cpu_oracle"] = abs(float(evaluate.ssim(tgt, got[0].numpy())) - float(evaluate.ssim(tgt, ref_out[0].numpy())))
        line["parity_nmse_vs_cpu_oracle"] = float(evaluate.nmse(ref_out[0].numpy(), got[0].numpy()))
        if args.config == 3:
            line["parity_note"] = ("a 10-cascade XPDNet with random weights amplifies rounding: the reference's own fp32 output is 7.0e-4 of the "
                                   "peak (NMSE 2.9e-7) from its fp64 evaluation (tests/golden/xpdnet_cfg3.npz); every cascade alone agrees with "
                                   "the oracle to 1.6e-6 (tests/test_hip_parity.py::test_xpdnet_cfg3_every_cascade_vs_oracle)")
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

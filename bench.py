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

Rank 0 prints ONE JSON line.  `roofline` = the dominant kernel family (3x3 conv on fp32 MFMA).  `roofline.frac` =
`frac_timed_mode`: the family's algorithmic FLOPs of the K timed slices over the TIMED REGION's wall time (the mode `ms_per_step`
comes from: graphs of several slices overlap, so no per-launch duration exists there -- it is the WHOLE-CHIP fp32-MFMA utilisation
over the timed region, the figure the driver's clock anchors; rounds 1-3 printed the isolated per-launch figure under this key);
`frac_isolated` = the same FLOPs over the kernels' own duration with ONE slice in flight (eager launches, hipEvent
pairs on the launch stream -- agrees with profiles/rNN_rocprofv3_kernel_stats_isolated.csv).  `roofline_fft_dc` = the FFT +
data-consistency family, both in SURVEY 8(d)'s algorithmic bytes and in real (PMC) bytes.  `latency_ms_one_slice` = one graph
replay on one stream (what run_inference.py:53-61 times); `sustained_value` = one >= 60 s region of the same steps with the
slices finished in every 1-s window and the GPU's shader clock / power sampled beside them.  `cpu_baseline` = the CPU oracle on this host's cores (thread sweep, then >= 3 forwards at the
best setting); `oracle/` is imported by that leg (and the parity check it feeds) only.
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

AFFINITY = {"numa_node": None, "cpus": None, "applied": False}      # filled by main() before the first GPU call
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
def oracle_model(cfg_id):
    """The CPU oracle's model of a configuration.  The ONLY place bench.py touches `oracle/`: called by the cpu_baseline leg and
    by the parity checks that compare a replayed HIP output with it -- never by the workload set-up or the timed region."""
    if cfg_id == 2:
        from oracle import varnet_ref as V
        return V.VarNet(6, 8, 3, 16, 3, "XF")
    if cfg_id == 3:
        from oracle import xpdnet_ref as X
        return X.XPDNet(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT")
    if cfg_id == 4:
        from oracle import cinenet_ref as C
        return C.CineNet(6, 6, 16, 3, "3D")
    from oracle import recurrent_ref as R
    return R.VarNet_RNN(5, 8, 3, 16)


def _cfg2():
    import reconstruction.models as M
    flop = 2.0 * (6 * (H + W) * unet_conv3_macs(16, 3, 2, pad16(W), pad16(FRAMES)) + COILS * unet_conv3_macs(8, 3, 2, pad16(H), pad16(W)))
    k_mb, i_mb, s_mb = 72.0, 4.8, 4.8
    fft_bytes = (6 * ((k_mb + s_mb + i_mb) + (i_mb + s_mb + k_mb + k_mb)) + (k_mb + s_mb + 2.4)) * MB
    return dict(name="BASELINE.json configs[1]: XF-VarNet, 6 cascades, 15 coils x 15 frames x 200x200, R=4 Gaussian-density "
                     "Cartesian mask, sens net 8ch/3 pools, U-Net 16ch/3 pools",
                metric="cine slices/sec, XF-VarNet R=4 15-coil 200x200x15t", accel=4, noise=0.0, wseed=1, keep=("lambda",),
                hip=lambda: M.VarNet(6, 8, 3, 16, 3, "XF"), ref=lambda: oracle_model(2), needs_sens=False,
                conv_flop=flop, conv_kernel="the 3x3 family: cine::conv_plane_kernel<8, CT, WM, WN, MT, TW, MODE> (U-Net planes, 84 of 98 launches), conv_wide_kernel / conv_mfma_kernel<..., 9, 0> (sens-net)",
                fft_bytes=fft_bytes)


def _cfg3():
    import reconstruction.models as M
    kw = dict(num_cascades=10, sens_chans=8, sens_pools=3, n_primal=5, dynamic_type="XT")
    return dict(name="BASELINE.json configs[2]: XT-XPDNet, MWCNN regulariser (script defaults), 10 cascades, n_primal 5, "
                     "15 coils x 15 frames x 200x200, R=8", metric="cine slices/sec, XT-XPDNet R=8 15-coil 200x200x15t",
                accel=8, noise=0.01, wseed=6, keep=(), hip=lambda: M.XPDNet(**kw), ref=lambda: oracle_model(3), needs_sens=False,
                conv_flop=416.2e9, conv_kernel="cine::conv_mfma_kernel<8, ..., 9, 0> (MWCNN 3x3 convs with Haar DWT / IWT on load) + conv_plane_kernel (its InstanceNorm inner convs)",
                fft_bytes=None)


def _cfg4():
    import reconstruction.models as M
    return dict(name="BASELINE.json configs[3]: 3D CineNet, 6 cascades, CG 6, U-Net3D 16ch/3 pools, 15 coils x 15 frames x 200x200, R=6",
                metric="cine slices/sec, 3D CineNet R=6 15-coil 200x200x15t", accel=6, noise=0.0, wseed=7, keep=("lambda",),
                hip=lambda: M.CineNet(6, 6, 16, 3, "3D"), ref=lambda: oracle_model(4), needs_sens=True,
                conv_flop=365.2e9, conv_kernel="cine::conv_wide_kernel<..., V3 = 1> (3x3x3 convs as three 3x3 passes, levels 0 / 1) + conv_coarse_kernel (coarse levels: flattened positions, K split over the waves)", fft_bytes=None)


def _cfg5():
    import reconstruction.models as M
    return dict(name="BASELINE.json configs[4]: CRNN-VarNet, 5 cascades, sens net 8ch/3 pools, 16 hidden channels, "
                     "15 coils x 15 frames x 200x200, R=8", metric="cine slices/sec, CRNN-VarNet R=8 15-coil 200x200x15t",
                accel=8, noise=0.0, wseed=9, keep=("lambda",), hip=lambda: M.VarNet_RNN(5, 8, 3, 16),
                ref=lambda: oracle_model(5), needs_sens=False, conv_flop=155.0e9,
                conv_kernel="cine::conv_wide_kernel (CRNN cells: summed-input 3x3 convs with bias/addend/ReLU epilogue, paired time-sweep steps) + conv_plane_kernel (sens-net)",
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
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL (backend nccl), the device barrier and the all-gather of the volume assembly even at --gpus 1 "
                         "(start under `python -m torch.distributed.run --nproc-per-node=1`, or alone: rank 0 of a world of 1)")
    ap.add_argument("--branches", type=int, default=0, help="concurrent branches of every 2-D U-Net pass in the timed (throughput) mode: 1 (default), 2 or 4")
    ap.add_argument("--no-conv-plane", action="store_true", help="A/B: route the plane-wide 3x3 / transpose convs through the general kernel (cine_set_conv_plane(0))")
    ap.add_argument("--conv-plane-mask", type=int, default=-1, help="A/B: cine_set_conv_plane(mask) of the bench's thread: bit 0 the plane-wide 3x3 convs, bit 1 the transpose convs, bit 2 the wide-plane / volume kernel "
                         "(conv_wide_kernel: cfg 4, cfg 5, the sensitivity net), bit 4 SET = the general weight-gradient kernel; 7 = all lean kernels (the default)")
    ap.add_argument("--pin-numa", action="store_true", help="pin this process to its GPU's NUMA node at --gpus 1 too (always done for N > 1)")
    ap.add_argument("--sustained-seconds", type=float, default=60.0, help="length of the extra sustained region (0 = skip)")
    ap.add_argument("--latency-only", action="store_true", help="diagnostics: print latency_modes_ms of --config (one slice alone, every launch form) and exit")
    ap.add_argument("--headline-only", action="store_true", help="skip the latency / sustained regions and the other_configs / train_step extras of the default 1-GPU line")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="TEST ONLY (tests/test_distributed_cpu.py): run the launcher, sharding, timed region and all-gather on "
                         "gloo/CPU with a stand-in for the forward; the line is marked invalid and measures nothing")
    args = ap.parse_args()
    args.sustained_explicit = any(a.startswith("--sustained-seconds") for a in sys.argv[1:])
    return args


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


# ------------------------------------------------------------------ CPU affinity of a rank
def _parse_cpulist(txt):
    cpus = set()
    for part in txt.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_cpus(local_rank, sysfs="/sys"):
    """CPUs of the NUMA node the `local_rank`-th GPU hangs off, read from sysfs alone (no HIP call: the affinity must be set
    before the runtime starts its threads and pins its host buffers).  KFD lists the GPUs in the order HIP enumerates them
    (topology nodes with simd_count > 0); `location_id` (bus << 8 | devfn) + `domain` name the PCI function whose `numa_node`
    and `local_cpulist` the kernel exports.  Returns (node, cpus) or (None, None) when anything is missing."""
    try:
        base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
        gpus = []
        for name in sorted(os.listdir(base), key=lambda x: int(x) if x.isdigit() else 1 << 30):
            props = {}
            with open(os.path.join(base, name, "properties")) as f:
                for ln in f:
                    k, _, v = ln.strip().partition(" ")
                    props[k] = v
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if vis and all(x.strip().isdigit() for x in vis.split(",")):
            gpus = [gpus[int(x)] for x in vis.split(",") if int(x) < len(gpus)]
        pr = gpus[local_rank]
        loc, dom = int(pr["location_id"]), int(pr.get("domain", "0"))
        bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
        dev = os.path.join(sysfs, "bus/pci/devices", bdf)
        with open(os.path.join(dev, "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            # a box whose firmware does not name the node (numa_node = -1, seen on the driver's GPU boxes): the PCI function still
            # exports the CPUs local to it (local_cpulist = the root complex's cpumask); node id -1 marks this path in the result
            with open(os.path.join(dev, "local_cpulist")) as f:
                cpus = _parse_cpulist(f.read())
            return (-1, cpus) if cpus else (None, None)
        with open(os.path.join(sysfs, f"devices/system/node/node{node}/cpulist")) as f:
            cpus = _parse_cpulist(f.read())
        return node, (cpus or None)
    except (OSError, KeyError, IndexError, ValueError):
        return None, None


def pin_rank_to_gpu_numa(local_rank, world, sysfs="/sys"):
    """os.sched_setaffinity to the GPU's NUMA node, intersected with what this process may use (cgroup / taskset).  Never a
    re-exec.  With N ranks on one node that would otherwise all inherit every CPU, the host side of a rank (graph launches,
    pinned buffers, the phantom generator) stays next to its GPU."""
    info = {"numa_node": None, "cpus": None, "applied": False, "why": None}
    node, cpus = gpu_numa_cpus(local_rank, sysfs)
    if node is None:
        info["why"] = "sysfs names neither a NUMA node nor local CPUs for this GPU"
        return info
    try:
        allowed = os.sched_getaffinity(0)
        want = cpus & allowed
        info["numa_node"] = node
        info["source"] = "numa_node + node cpulist" if node >= 0 else "local_cpulist of the PCI function (numa_node = -1)"
        if want and want != allowed:
            os.sched_setaffinity(0, want)
            info["applied"] = True
        else:
            info["why"] = ("the GPU's local CPUs are every CPU this process may use (one node, or a cpuset that is already local)" if want
                           else "none of the GPU's local CPUs is in this process's cpuset")
        info["cpus"] = len(want or allowed)
    except (OSError, AttributeError) as e:
        info["why"] = f"sched_setaffinity: {e}"
    return info


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
    coll = world > 1 or (shard.FORCE_COLLECTIVE and dist.is_initialized())
    sync()
    if coll:
        barrier()
    sync()
    t0 = time.perf_counter()
    run(nsteps)
    volume = shard.assemble_volume(outs, world * nsteps * batch)      # one all-gather (RCCL over xGMI; no-op at N=1)
    assert volume.shape[0] == world * nsteps * batch
    sync()
    if coll:
        barrier()
    sync()
    dt = time.perf_counter() - t0
    per_rank = [dt]
    if coll:
        mine = torch.tensor([dt], device=device, dtype=torch.float64)
        allr = torch.empty(dist.get_world_size(), device=device, dtype=torch.float64)
        dist.all_gather_into_tensor(allr, mine)               # every rank's own clock over the same region: a straggler shows
        per_rank = [float(v) for v in allr.cpu()]
        dt = max(per_rank)                                     # the contract's time: MAX over ranks
    return dt, volume, per_rank


def leave_together(use_dist, barrier):
    """The end of every rank's main(): rank 0 still has host-side work after the timed region (kernel families, the line itself); the other
    ranks wait for it at ONE last barrier, so that no rank tears its communicator down while another could still enter a collective."""
    if use_dist:
        barrier()
        dist.destroy_process_group()


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
    dt, volume, per_rank = timed_steps(run, K, 1, outs, world, lambda: None, dist.barrier, torch.device("cpu"))
    ok = bool(torch.equal(volume, torch.stack([standin(i) for i in range(world * K)])))
    # main()'s ordering behind the timed region: the single-GPU extras are skipped at N > 1, ranks != 0 go straight to the last barrier, rank 0
    # does its host-side work first (a stand-in pause) and prints -- nobody leaves before rank 0 arrives
    extras = world == 1 or args.sustained_explicit
    t_leave = time.perf_counter()
    if rank == 0:
        time.sleep(0.5)
        print(json.dumps({"metric": "selftest (no measurement)", "value": None, "valid": False, "n_gpus": world, "rccl_ranks": world,
                          "steps": K, "warmup": args.warmup, "volume_ok": ok, "volume_slices": int(volume.shape[0]),
                          "data": "cpu stand-in for the forward (test only)", "timed_region_s": dt, "single_gpu_extras": extras,
                          "per_rank_timed_region_s": per_rank, "cpu_affinity": AFFINITY}))
    leave_together(world > 1, dist.barrier)
    waited = time.perf_counter() - t_leave
    if world > 1 and rank != 0 and waited < 0.4:
        raise SystemExit(4)                       # a rank left before rank 0 had finished its host-side work
    if not ok:
        raise SystemExit(3)


_STREAM_POOL = []


def bench_streams(n):
    """The bench's compute streams, created ONCE per process and shared by every Workload: torch hands out streams from a fixed pool of 32 per
    device and the runtime maps them onto GPU_MAX_HW_QUEUES = 16 hardware queues as they are first used -- a process that keeps asking for fresh
    streams (one set per configuration) soon has two ACTIVE streams on one queue (measured: cfg 4 165 -> 157, cfg 5 370 -> 330 slices/s, and a
    training step whose weight-gradient side stream shares the main stream's queue: 33 -> 46 ms).  Twelve streams + the copy stream + a side
    stream or two + the null stream stay within the 16 queues."""
    while len(_STREAM_POOL) < max(n, 13):          # all of them at the FIRST call: before any side stream / probe candidate has claimed a queue
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            torch.zeros(1, device="cuda")            # first use = the moment the runtime gives the stream its hardware queue
        _STREAM_POOL.append(st)
    return _STREAM_POOL[:n]


class Workload:
    """One BASELINE configuration set up on this rank's GPU: S different slices resident in HBM, the drop-in model, one hipGraph
    per slice (its own stream), and the timed region of the contract over them."""

    def __init__(self, cfg_id, args, world, rank, local, dev, steps, inflight=0):
        from concurrent.futures import ThreadPoolExecutor
        from cine_hip import synth
        self.cfg = cfg = CONFIGS[cfg_id]()
        self.args, self.world, self.rank, self.local, self.dev, self.steps = args, world, rank, local, dev, steps
        if inflight > 0:
            S = max(1, min(inflight, steps))
        else:
            S = next((c for c in range(12, 5, -1) if steps % c == 0), 8) if steps >= 6 else max(1, steps)
        self.S, self.B = S, max(1, args.batch)
        B = self.B
        # S DIFFERENT slices per rank (x B on the batch axis): seeds rank * S * B + ...; the phantoms are host work (0.8 s each):
        # a small thread pool per rank, with the intra-op threads divided among the ranks of the node
        nthr = torch.get_num_threads()
        torch.set_num_threads(max(1, nthr // max(world, 1) // 4))
        with ThreadPoolExecutor(max_workers=4) as pool:
            flat = list(pool.map(lambda sd: synth.make_cine_slice(FRAMES, COILS, H, W, accel=cfg["accel"], seed=sd, noise_std=cfg["noise"]),
                                 [(rank * S + i) * B + j for i in range(S) for j in range(B)]))
        torch.set_num_threads(nthr)
        self.exs = exs = [flat[i * B:(i + 1) * B] for i in range(S)]
        self.host_mk = [torch.cat([e["masked_kspace"] for e in row]).pin_memory() for row in exs]
        self.mks = [h.to(dev, non_blocking=True) for h in self.host_mk]
        self.masks = [torch.cat([e["mask"] for e in row]).to(dev) for row in exs]
        self.senss = [torch.cat([e["sens_maps"] for e in row]).to(dev) for row in exs] if cfg["needs_sens"] else None
        net = cfg["hip"]().eval()
        synth.fill_parameters_(net, cfg["wseed"], keep=cfg["keep"])
        self.net = net.to(dev)
        from reconstruction.models.varnet import SensitivityModel
        self.acss = [SensitivityModel.acs_window(m) for m in self.masks]             # host read-back of the 1-D mask, outside capture
        out = self.forward()                               # also packs the weights
        torch.cuda.synchronize()
        self.streams = bench_streams(S)
        self.graphs, self.gouts = [], []
        self.use_graph = not args.no_graph
        if self.use_graph:
            try:
                for i in range(S):
                    with torch.cuda.stream(self.streams[i]):
                        self.forward(i)                     # warm this stream's caches outside capture
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self.streams[i]):
                        o = self.forward(i)
                    self.graphs.append(g); self.gouts.append(o)
            except Exception as e:                                        # pragma: no cover
                if rank == 0:
                    print(f"# hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
                self.use_graph = False
                self.graphs, self.gouts = [], []
        torch.cuda.synchronize()
        for i, g in enumerate(self.graphs):                 # part of the set-up: the first replay of an instantiated graph uploads it
            with torch.cuda.stream(self.streams[i]):
                g.replay()
        torch.cuda.synchronize()
        self.outs = torch.empty((steps * B,) + tuple(out.shape[1:]), device=dev)   # this rank's slices
        for st in self.streams:
            st.wait_stream(torch.cuda.current_stream())

    def forward(self, i=0, branches=None, buf=0):
        """One forward of slice i.  The throughput mode (S slices in flight, one stream each) runs every U-Net pass on ONE stream (--branches, default 1):
        the other slices fill the chip, and side streams would only compete for hardware queues; the binding's own default (two concurrent branches,
        the better form for ONE slice at a time) is what `latency_modes` measures beside it."""
        from cine_hip import ops
        mk = (self.mks2 if buf else self.mks)[i]
        self.nforward = getattr(self, "nforward", 0) + 1            # (eager forwards + captures: what a --no-graph PMC pass divides its counters by)
        with ops.branches(branches or self.args.branches or 1):
            if self.cfg["needs_sens"]:
                return self.net(mk, self.masks[i], self.senss[i])
            return self.net(mk, self.masks[i], acs=self.acss[i])

    def run(self, nsteps, keep, h2d=False):
        """nsteps slices, round-robin over the S streams; each stream is an in-order queue.  h2d: every slice's k-space comes from its pinned host
        buffer, DOUBLE-BUFFERED: a stream alternates between two device buffers (two captured graphs); while step k replays from one, the copy of
        step k + S goes into the other on the copy stream -- it only has to wait for step k - S, which finished a whole round ago, so neither the
        copy stream nor the host ever waits for a replay (a copy that depends on a replay still in flight blocks the HOST inside hipMemcpyAsync on
        this runtime: measured 148 - 155 slices/s for that form against 172 with resident data).  h2d == "inline": the copy on the slice's own stream
        in front of its replay (round 5's form, kept for the A/B)."""
        S, B = self.S, self.B
        ahead = h2d and h2d != "inline"
        if ahead:
            if not getattr(self, "_prefetched", False):                     # outside a prefetching caller: the first round's copies are issued here
                self.prefetch()
            self._prefetched = False
            self.copy_stream.wait_stream(torch.cuda.current_stream())
        for k in range(nsteps):
            i = k % S
            s_ = (k // S) & 1 if ahead else 0                               # which of the stream's two buffers / graphs this step reads
            with torch.cuda.stream(self.streams[i]):
                if ahead:
                    self.streams[i].wait_event(self.ev_ready[s_][i])        # this slice's k-space has arrived (copied while the previous round replayed)
                elif h2d:
                    self.mks[i].copy_(self.host_mk[i], non_blocking=True)      # pinned host buffer -> HBM, ahead of this slice's replay
                if self.use_graph:
                    (self.graphs2 if s_ else self.graphs)[i].replay()
                    o = (self.gouts2 if s_ else self.gouts)[i]
                else:
                    o = self.forward(i, buf=s_)
                if keep:
                    self.outs[k * B:(k + 1) * B].copy_(o)
                if ahead:
                    self.ev_done[s_][i].record()
            if ahead:      # step k + S of this stream reads the OTHER buffer: copy it now (its last reader, step k - S, is long done)
                o_ = s_ ^ 1
                if k >= S:
                    self.copy_stream.wait_event(self.ev_done[o_][i])
                with torch.cuda.stream(self.copy_stream):
                    (self.mks2 if o_ else self.mks)[i].copy_(self.host_mk[i], non_blocking=True)
                    self.ev_ready[o_][i].record()
        for st in self.streams:
            torch.cuda.current_stream().wait_stream(st)

    def prefetch(self):
        """Set-up of the double-buffered copies (second device buffers, second graphs) and the copies of the FIRST round (one slice per stream): in a
        running loop they happened while the previous slices were reconstructed; a timed region issues them before its clock starts (every later
        copy -- nsteps of them, one per replay -- is inside the region)."""
        S = self.S
        if getattr(self, "copy_stream", None) is None:
            self.copy_stream = bench_streams(13)[12]
            self.ev_ready = [[torch.cuda.Event() for _ in range(S)] for _ in range(2)]
            self.ev_done = [[torch.cuda.Event() for _ in range(S)] for _ in range(2)]
            self.mks2 = [torch.empty_like(m) for m in self.mks]
            self.graphs2, self.gouts2 = [], []
            if self.use_graph:
                torch.cuda.synchronize()
                for i in range(S):
                    self.mks2[i].copy_(self.mks[i])
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self.streams[i]):
                        o = self.forward(i, buf=1)
                    self.graphs2.append(g); self.gouts2.append(o)
                torch.cuda.synchronize()
                for i, g in enumerate(self.graphs2):
                    with torch.cuda.stream(self.streams[i]):
                        g.replay()
                torch.cuda.synchronize()
        for st in self.streams:
            self.copy_stream.wait_stream(st)
        with torch.cuda.stream(self.copy_stream):
            for i in range(S):
                self.mks[i].copy_(self.host_mk[i], non_blocking=True)
                self.ev_ready[0][i].record()
        self.copy_stream.synchronize()
        self._prefetched = True

    def timed(self, nsteps, h2d=False):
        if h2d and h2d != "inline":
            self.prefetch()
        dt, _, self.per_rank_s = timed_steps(lambda n: self.run(n, True, h2d), nsteps, self.B, self.outs, self.world, torch.cuda.synchronize,
                                             lambda: dist.barrier(device_ids=[self.local]), self.dev)
        return dt

    def latency_one_slice(self, reps=15):
        """One slice, one stream, nothing else on the GPU: what the reference's inference loop times per slice
        (traintest_scripts/run_inference.py:53-61: model(masked_kspace, mask) between two clocks, batch 1).  Median of `reps`."""
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            with torch.cuda.stream(self.streams[0]):
                if self.use_graph:
                    self.graphs[0].replay()
                else:
                    self.forward(0)
            self.streams[0].synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2] * 1e3, ts[0] * 1e3

    def latency_modes(self, reps=15, branch_counts=(1, 2)):
        """`latency_one_slice` for every launch form of ONE slice: hipGraph replay and eager launches, with the 2-D U-Net passes on one stream or as
        2 / 4 concurrent branches (cine_unet2d_forward_branches: x-f / y-f networks, coil halves of the sens-net; bit-identical outputs).  The
        graphs of the timed region are left alone; the extra ones are captured for slice 0 on stream 0 and dropped afterwards."""
        from cine_hip import ops
        out = {}
        st = self.streams[0]
        ref = None
        for nb in branch_counts:
            with torch.cuda.stream(st):
                o = self.forward(0, nb).clone()              # warms this stream's side streams outside capture
            torch.cuda.synchronize()
            ref = o if ref is None else ref
            same = bool(torch.equal(o, ref))
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                with torch.cuda.stream(st):
                    self.forward(0, nb)
                st.synchronize()
                ts.append(time.perf_counter() - t0)
            ts.sort()
            rec = {"eager_ms": ts[len(ts) // 2] * 1e3, "eager_min_ms": ts[0] * 1e3, "bit_identical_to_one_stream": same}
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    go = self.forward(0, nb)
                with torch.cuda.stream(st):
                    g.replay()
                torch.cuda.synchronize()
                rec["bit_identical_to_one_stream"] = same and bool(torch.equal(go, ref))
                ts = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    with torch.cuda.stream(st):
                        g.replay()
                    st.synchronize()
                    ts.append(time.perf_counter() - t0)
                ts.sort()
                rec["graph_ms"], rec["graph_min_ms"] = ts[len(ts) // 2] * 1e3, ts[0] * 1e3
                del g, go
            except Exception as e:                                    # pragma: no cover
                rec["graph_error"] = f"{type(e).__name__}: {e}"
            out[str(nb)] = rec
        return out

    def sustained(self, seconds, est_step_s):
        """One long region of the timed mode (no volume assembly, no host sync inside): an event per finished slice, then the
        slices finished in every 1-s window.  Shows what clocks / power settle at, which a 0.13 s region cannot."""
        S, B = self.S, self.B
        n = max(S, int(seconds / est_step_s / S + 1) * S)
        ev0 = torch.cuda.Event(enable_timing=True)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        torch.cuda.synchronize()
        for st in self.streams:
            st.wait_stream(torch.cuda.current_stream())
        ev0.record(self.streams[0])
        t0 = time.perf_counter()
        with GpuTelemetry(self.dev.index or 0) as tele:
            for k in range(n):
                i = k % S
                with torch.cuda.stream(self.streams[i]):
                    if self.use_graph:
                        self.graphs[i].replay()
                    else:
                        self.forward(i)
                    evs[k].record()
                if k % (8 * S) == 8 * S - 1 and k >= 64 * S:      # keep the host at most ~64 rounds ahead: the sampler thread needs the GIL now and then
                    evs[k - 64 * S].synchronize()
            torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        done = sorted(ev0.elapsed_time(e) * 1e-3 for e in evs)              # completion times, seconds after the start
        total = done[-1]
        nwin = int(total)                                                    # whole 1-s windows
        per = [sum(1 for d in done if w <= d < w + 1) * B for w in range(nwin)]
        return {"value": n * B / total, "unit": "cine slices/sec", "steps": n, "seconds": total, "host_wall_s": wall,
                "window_s": 1.0, "window_min": min(per) if per else None, "window_max": max(per) if per else None,
                "windows": per, "gpu_telemetry": tele.summary(),
                "note": "same graphs / streams as the timed region, one event per finished slice on its stream; no volume assembly inside; "
                        "gpu_telemetry = shader clock / power / busy sampled from sysfs once per second beside the windows"}

    def replayed_output(self):
        """Output of stream 0's slice through the timed mode (a graph replay when graphs are in use)."""
        if self.use_graph:
            self.graphs[0].replay()
            torch.cuda.synchronize()
            return self.gouts[0][:1].clone()
        return self.forward(0)[:1].clone()

    def release(self):
        """Drop the graphs, tensors AND streams of this workload.  The streams matter: the runtime maps streams onto GPU_MAX_HW_QUEUES hardware
        queues, and once more streams are alive than queues a NEW stream (the next workload's, a training step's side stream) shares a queue
        with an old one -- measured: 26 idle streams left behind turn the cfg-2 training step from 33 ms into 46.6 ms (its weight-gradient side
        stream lands on the main stream's queue)."""
        import gc
        from cine_hip import ops
        torch.cuda.synchronize()
        self.graphs, self.gouts = [], []
        self.graphs2, self.gouts2 = [], []
        for name in ("mks", "mks2", "masks", "senss", "outs", "net", "host_mk", "exs", "copy_stream", "ev_ready", "ev_done"):
            if hasattr(self, name):
                setattr(self, name, None)
        self.streams = []                                      # (the stream objects themselves live in bench_streams' pool and are reused)
        gc.collect()
        torch.cuda.empty_cache()


class GpuTelemetry:
    """Shader clock / power / busy samples of the GPU under test while a region runs, read from the amdgpu sysfs files (no child
    process, no HIP call): pp_dpm_sclk (the level marked '*'), hwmon power1_average (uW), gpu_busy_percent.  The card is the one
    whose PCI address torch reports for the device; failing that, the busiest card at the first sample.  Every field is optional:
    a box that hides sysfs gives an empty series and the bench line says so."""

    def __init__(self, dev_index=0, period_s=1.0):
        import glob
        import threading
        self.period, self.samples, self._stop, self._thr = period_s, [], threading.Event(), None
        self.cards = [d for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")) if os.path.exists(os.path.join(d, "pp_dpm_sclk"))]
        self.card, self.how = None, None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
            for d in self.cards:
                if os.path.basename(os.path.realpath(d)).startswith(bdf):
                    self.card, self.how = d, "pci address of the torch device"
        except Exception:
            pass

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return f.read()
        except OSError:
            return None

    def _sample(self, d):
        import glob
        out = {}
        txt = self._read(os.path.join(d, "pp_dpm_sclk"))
        if txt:
            cur = [ln for ln in txt.splitlines() if ln.rstrip().endswith("*")]
            if cur:
                try:
                    out["sclk_mhz"] = int("".join(ch for ch in cur[0].split(":")[1] if ch.isdigit()))
                except (IndexError, ValueError):
                    pass
        for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_input")):
            txt = self._read(hw)
            if txt and txt.strip().isdigit():
                out["power_w"] = round(int(txt) / 1e6, 1)
                break
        txt = self._read(os.path.join(d, "gpu_busy_percent"))
        if txt and txt.strip().isdigit():
            out["busy_pct"] = int(txt)
        return out

    def _loop(self, t0):
        while not self._stop.is_set():
            if self.card is None and self.cards:
                busy = [(self._sample(d).get("busy_pct", -1), d) for d in self.cards]
                if max(busy)[0] > 0:
                    self.card, self.how = max(busy)[1], "the busiest card at the first sample"
            if self.card is not None:
                smp = self._sample(self.card)
                if smp:
                    smp["t_s"] = round(time.perf_counter() - t0, 2)
                    self.samples.append(smp)
            self._stop.wait(self.period)

    def __enter__(self):
        import threading
        self._thr = threading.Thread(target=self._loop, args=(time.perf_counter(),), daemon=True)
        self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thr.join(timeout=5)

    def summary(self):
        if not self.samples:
            return {"samples": 0, "note": "no readable amdgpu sysfs telemetry on this box (pp_dpm_sclk / hwmon power1_average / gpu_busy_percent)"}
        out = {"samples": len(self.samples), "period_s": self.period, "card": self.card, "card_chosen_by": self.how}
        for k in ("sclk_mhz", "power_w", "busy_pct"):
            v = [x[k] for x in self.samples if k in x]
            if v:
                out[k] = {"min": min(v), "max": max(v), "mean": round(sum(v) / len(v), 1), "series": v}
        out["t_s"] = [x["t_s"] for x in self.samples]
        return out


def pmc_families(cfg_id):
    """HBM-side bytes / MFMA busy fractions per kernel family from the committed rocprofv3 --pmc passes of this same command
    (tools/collect_profiles.sh -> tools/pmc_traffic.py / pmc_mfma.py; PMC collection cannot run inside the bench itself)."""
    sfx = "" if cfg_id == 2 else f"_cfg{cfg_id}"
    import glob
    rounds = sorted({os.path.basename(p_).split("_")[0] for p_ in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic*.json"))}, reverse=True)
    for rnd_ in rounds:          # the latest round's passes first
        tpath = os.path.join(ROOT, "profiles", f"{rnd_}_pmc_traffic{sfx}.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            return tj["families"], f"profiles/{os.path.basename(tpath)}" + (f" @ {tj['commit']}" if "commit" in tj else "")
    return None, None


def conv_roofline(wl, fam, dt, slices):
    """`achieved` / `frac` describe the TIMED mode (the family's algorithmic FLOPs of the timed slices over the timed region's
    wall time: kernels of several slices overlap there, so no per-launch duration exists -- this is the whole-chip fp32-MFMA
    utilisation over the timed region; it can exceed the isolated figure, whose launches leave half-empty last rounds that
    the other slices in flight fill); the *_isolated keys are the per-launch view (one slice in flight,
    hipEvent pairs around every launch of the family)."""
    cfg = wl.cfg
    conv_ms, conv_n = fam["conv3x3_mfma"]
    timed_tf = cfg["conv_flop"] * slices / wl.world / dt / 1e12
    iso_tf = cfg["conv_flop"] / (conv_ms * 1e-3) / 1e12
    mode_t = ("hipGraph replay" if wl.use_graph else "eager") + f", {wl.S} slices in flight (the timed region)"
    return {"bound": "mfma", "kernel": cfg["conv_kernel"],
            "achieved": timed_tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": timed_tf / MFMA_F32_PEAK_TFLOPS,
            "mode": "timed mode: " + mode_t + "; FLOPs of the 3x3 family / wall time of the timed region",
            "frac_timed_mode": timed_tf / MFMA_F32_PEAK_TFLOPS, "achieved_timed_mode": timed_tf,
            "frac_isolated": iso_tf / MFMA_F32_PEAK_TFLOPS, "achieved_isolated": iso_tf,
            "mode_isolated": "eager launches, one slice in flight, hipEvent pairs per launch",
            "traffic": None, "launches_per_slice": conv_n, "ms_per_slice_isolated": conv_ms,
            "avg_launch_us_isolated": conv_ms * 1e3 / max(conv_n, 1), "flop_per_slice": cfg["conv_flop"]}


def measure_other_config(cfg_id, args, dev, threads):
    """Another BASELINE configuration for the default line's `other_configs`: the same timed region, long enough to last >= 1 s
    (a 12-step region is 0.03 - 0.08 s: inside one clock / power state), the conv family's isolated and in-flight MFMA fractions,
    and the parity of a replayed output against the CPU oracle (1 warm-up forward, then as many timed ones as fit ~30 s, at most 3)."""
    from cine_hip import synth
    steps = 12
    wl = Workload(cfg_id, args, 1, 0, 0, dev, steps)
    wl.run(2, False)
    est = wl.timed(steps) / steps
    reps = max(1, int(1.05 / (est * steps)) + 1)              # whole rounds of the S streams, >= 1 s in total
    steps_short = steps
    steps = steps * reps
    short_dt = min(wl.timed(steps_short) for _ in range(2))
    wl.outs = torch.empty((steps * wl.B,) + tuple(wl.outs.shape[1:]), device=dev)     # the long region keeps every output too
    wl.steps = steps
    dt = min(wl.timed(steps) for _ in range(2))
    fam = profile_families(wl.forward, iters=2)
    roof = conv_roofline(wl, fam, dt, steps * wl.B)
    tr, tr_src = pmc_families(cfg_id)
    if tr and "conv3x3_mfma" in tr:
        roof["traffic"] = tr["conv3x3_mfma"]["hbm_MB_per_launch"] * 1e6
        roof["traffic_unit"] = f"HBM bytes per launch, PMC FETCH_SIZE x2 + WRITE_SIZE, {tr_src}"
    res = {"workload": wl.cfg["name"], "metric": wl.cfg["metric"], "value": steps * wl.B / dt, "unit": "cine slices/sec", "steps": steps,
           "ms_per_step": dt / steps * 1e3, "timed_region_s": dt, "value_12_step_region": steps_short * wl.B / short_dt,
           "slices_in_flight": wl.S, "launch": "hipGraph replay" if wl.use_graph else "eager",
           "roofline": {k: roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_timed_mode", "frac_isolated", "traffic",
                                             "ms_per_slice_isolated", "launches_per_slice")},
           "kernel_ms_per_slice": {k: round(v[0], 4) for k, v in fam.items() if v[1]}}
    try:
        lm = wl.latency_modes(reps=9)
        res["latency_modes_ms"] = lm
        res["latency_ms_one_slice"] = min(v[k] for v in lm.values() for k in ("eager_ms", "graph_ms") if k in v)
    except Exception as e:                                                # pragma: no cover
        res["latency_modes_ms"] = {"error": f"{type(e).__name__}: {e}"}
    chk = wl.replayed_output().cpu()
    ex0 = wl.exs[0][0]
    if not args.no_cpu_baseline:
        net = wl.cfg["ref"]().eval()
        synth.fill_parameters_(net, wl.cfg["wseed"], keep=wl.cfg["keep"])
        cargs = (ex0["masked_kspace"], ex0["mask"]) + ((ex0["sens_maps"],) if wl.cfg["needs_sens"] else ())
        torch.set_num_threads(threads)
        with torch.no_grad():
            t0 = time.perf_counter()
            ref_out = net(*cargs)
            warm = time.perf_counter() - t0
            nfw = max(1, min(3, int(30.0 / warm) - 1))             # ~30 s of CPU work per configuration, the warm-up included
            t0 = time.perf_counter()
            for _ in range(nfw):
                ref_out = net(*cargs)
            cdt = (time.perf_counter() - t0) / nfw
        res["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "cine slices/sec", "cores": threads, "kind": "port",
                               "sample": f"1 warm-up forward ({warm:.1f} s), then {nfw} timed forward(s) of the CPU oracle at {threads} threads: {cdt:.1f} s/slice"}
        res["parity_max_rel_err_vs_cpu_oracle"] = float((chk - ref_out).abs().max() / ref_out.abs().max())
        if cfg_id == 3:
            res["parity_note"] = ("10-cascade XPDNet with random weights: the reference's own fp32 output is 7.0e-4 of the peak from its fp64 "
                                  "evaluation (tests/golden/xpdnet_cfg3.npz); every cascade alone agrees with the oracle to 1.6e-6")
    wl.release()
    return res


def measure_training_step(args, dev, threads, cfg_id=2, cpu=True):
    """One training step of a BASELINE configuration (default configs[1]) through the HIP backward kernels: the body of reference
    pl_modules/varnet_module.py:97-113 (forward + SSIMLoss), loss.backward() and one Adam step (:151-154), k-space resident in HBM;
    and the same step of the CPU oracle on the host cores (ONE step: it takes tens of seconds)."""
    import reconstruction.models as M
    from reconstruction.utils import SSIMLoss
    from cine_hip import synth
    cfg = CONFIGS[cfg_id]()
    ex = synth.make_cine_slice(FRAMES, COILS, H, W, accel=cfg["accel"], seed=0, noise_std=cfg["noise"])

    def make(model, device):
        synth.fill_parameters_(model, cfg["wseed"], keep=cfg["keep"])
        model = model.to(device).train()
        lossf = SSIMLoss().to(device)
        opt = torch.optim.Adam(model.parameters(), lr=3e-4)
        mk, mask, target = ex["masked_kspace"].to(device), ex["mask"].to(device), ex["target"].to(device)
        extra = (ex["sens_maps"].to(device),) if cfg["needs_sens"] else ()

        def step():
            opt.zero_grad(set_to_none=True)
            out = model(mk, mask, *extra)
            loss = lossf(out.unsqueeze(1), target.unsqueeze(1), target.max())
            loss.backward()
            opt.step()
            return loss.detach()
        return step

    with torch.enable_grad():
        step = make(cfg["hip"](), dev)
        torch.cuda.reset_peak_memory_stats()                    # the peak of THIS training step, not of the workloads run before it
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            loss = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        fam = profile_families(step, iters=1)
        res = {"what": f"forward + SSIMLoss + backward (hand-written HIP gradient kernels) + Adam, one cfg-{cfg_id} slice per step, eager launches, "
                       "weight gradients on a side stream",
               "ms_per_step": ms, "steps_per_sec": 1e3 / ms, "loss_after_7_steps": float(loss),
               "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2 ** 30,
               "kernel_ms_per_step": {k: round(v[0], 3) for k, v in fam.items() if v[1]}}
        if cpu and not args.no_cpu_baseline:
            torch.set_num_threads(threads)
            cstep = make(cfg["ref"](), torch.device("cpu"))
            t0 = time.perf_counter()
            closs = cstep()
            cdt = time.perf_counter() - t0
            res["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "training steps/sec", "cores": threads, "kind": "port",
                                   "sample": f"ONE training step of the CPU oracle (torch autograd, fp32) at {threads} threads: {cdt:.1f} s",
                                   "loss_after_1_step": float(closs)}
    return res


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs")
    if world > 1 or args.pin_numa:
        # before the first GPU call: the runtime's threads and pinned buffers inherit the rank's CPU set
        AFFINITY.update(pin_rank_to_gpu_numa(local, world, os.environ.get("CINE_SYSFS_ROOT", "/sys")))
    if args.selftest_cpu:
        return selftest_cpu(args, world, rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        # RCCL ("nccl" on ROCm).  --force-dist initialises it at world 1 as well, so that the process group, the device barrier
        # and the all-gather of the volume assembly run on hardware even where only one GPU is visible.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.set_grad_enabled(False)                 # inference: the drop-in models build an autograd graph when gradients are enabled

    from cine_hip import shard
    shard.FORCE_COLLECTIVE = bool(args.force_dist)
    if args.no_conv_plane or args.conv_plane_mask >= 0:
        from cine_hip import ops as cine_ops
        cine_ops.set_conv_plane(0 if args.no_conv_plane else args.conv_plane_mask)
    if os.environ.get("CINE_EXTRA_STREAMS"):      # diagnostics: idle streams that have claimed hardware queues before anything else runs
        _idle = [torch.cuda.Stream() for _ in range(int(os.environ["CINE_EXTRA_STREAMS"]))]
        for s_ in _idle:
            with torch.cuda.stream(s_):
                torch.zeros(1, device=dev)
        torch.cuda.synchronize()
    wl = Workload(args.config, args, world, rank, local, dev, args.steps, args.inflight)
    cfg, S, B, use_graph = wl.cfg, wl.S, wl.B, wl.use_graph
    if args.latency_only:
        print(json.dumps({"config": args.config, "latency_ms_one_slice": wl.latency_one_slice(), "latency_modes_ms": wl.latency_modes()}))
        return
    wl.run(args.warmup, False)
    if dist.is_initialized() and (world > 1 or shard.FORCE_COLLECTIVE):
        # the warm-up also covers the timed region's collectives: RCCL sets up its channels on the first barrier / all-gather
        # (measured at world size 1: 151.4 slices/s in a first timed region against 154.3 in every later one)
        dist.barrier(device_ids=[local])
        shard.assemble_volume(wl.outs, world * args.steps * B)
        torch.cuda.synchronize()
    dt = wl.timed(args.steps)                                         # THE timed region: exactly K steps
    contract_per_rank_s = list(wl.per_rank_s)
    extra = sorted(wl.timed(args.steps) for _ in range(max(0, args.repeats)))
    dt_h2d = min(wl.timed(args.steps, h2d=True) for _ in range(2))
    dt_h2d_inline = wl.timed(args.steps, h2d="inline")
    # what the link alone delivers: the same pinned-host -> HBM copies back to back with nothing else on the GPU
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(wl.streams[0]):
        for k in range(2 * S):
            wl.mks[k % S].copy_(wl.host_mk[k % S], non_blocking=True)
    torch.cuda.synchronize()
    link_gbs = 2 * S * wl.host_mk[0].numel() * 4 / (time.perf_counter() - t0) / 1e9
    rccl_ranks = dist.get_world_size() if dist.is_initialized() else 1
    # the single-GPU diagnostics (one slice alone, the 60-s sustained region) belong to the N = 1 line: on N ranks they would keep N GPUs
    # busy for a minute each after the contract's region.  --sustained-seconds X forces the sustained region at N > 1 too (X > 0 given explicitly).
    extras = not args.headline_only and (world == 1 or args.sustained_explicit)
    lat_med, lat_min = wl.latency_one_slice() if extras else (None, None)
    lat_modes = wl.latency_modes() if (extras and rank == 0 and world == 1) else None
    sustained = wl.sustained(args.sustained_seconds, dt / args.steps) if args.sustained_seconds > 0 and extras else None
    best_lat = None
    if lat_modes:
        forms = [(v[k + "_ms"], v[k + "_min_ms"], f"{k} launches" if k == "eager" else "one hipGraph replay", nb)
                 for nb, v in lat_modes.items() for k in ("eager", "graph") if k + "_ms" in v]
        m = min(forms)
        best_lat = (m[0], m[1], f"{m[2]}, {m[3]} concurrent branch(es) per 2-D U-Net pass")

    if rank != 0:
        return leave_together(use_dist, lambda: dist.barrier(device_ids=[local]))

    # ---- rank 0: per-family device time, ISOLATED (eager launches, one slice in flight, hipEvents on the launch stream)
    fam = profile_families(wl.forward)
    fam = {k: (v[0] / B, v[1]) for k, v in fam.items()}            # per slice
    fft_ms = fam["fft_col_pass"][0] + fam["fft_row_pass"][0]
    slices = world * args.steps * B
    roofline = conv_roofline(wl, fam, dt, slices)
    tr, tr_src = pmc_families(args.config)
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
        "rccl_ranks": rccl_ranks, "rccl_initialised": bool(dist.is_initialized()),
        "timed_region_s": dt, "per_rank_timed_region_s": contract_per_rank_s, "cpu_affinity": AFFINITY,
        "latency_ms_one_slice": best_lat[0] if best_lat else lat_med, "latency_ms_one_slice_min": best_lat[1] if best_lat else lat_min,
        "latency_ms_one_slice_graph_one_stream": lat_med,
        "latency_note": ("one slice alone, nothing else in flight, host clock around enqueue + stream sync: what run_inference.py:53-61 times per slice "
                         "(median / min of 15); launch form: " + (best_lat[2] if best_lat else "one hipGraph replay on one stream") +
                         " (latency_modes_ms holds every form; _graph_one_stream = one hipGraph replay on one stream, round 5's figure)"),
        "latency_modes_ms": lat_modes,
        "latency_modes_note": "one slice alone, host clock around enqueue + stream sync, median / min of 15, per launch form: key = concurrent branches of "
                              "every 2-D U-Net pass (1 = one stream; 2 = x-f / y-f networks and coil halves beside each other on a side stream; 4 = "
                              "each halved again); eager launches or one hipGraph replay (the side streams are branches of the captured graph)",
        "sustained_value": sustained["value"] if sustained else None, "sustained": sustained,
        "repeat_values": [slices / d for d in extra], "repeat_median_value": (slices / extra[len(extra) // 2]) if extra else None,
        "value_with_h2d": slices / dt_h2d, "value_with_h2d_inline": slices / dt_h2d_inline,
        "h2d_link": {"GB_per_s_copies_alone": link_gbs, "MB_per_slice": wl.host_mk[0].numel() * 4 / 1e6 / B,
                     "slices_per_s_the_link_allows": link_gbs * 1e9 / (wl.host_mk[0].numel() * 4 / B)},
        "value_with_h2d_note": "same K steps with every slice's 72 MB k-space copied pinned-host -> HBM (what run_inference.py:53-61 times): one copy per replay "
                               "inside the region, on a copy stream, double-buffered (two device buffers and two captured graphs per stream: the copy of step "
                               "k + S runs while step k replays; the first round's copies are issued before the clock starts, as a running loop would have); "
                               "best of 2 regions; _inline = the copy on the slice's own stream in front of its replay (round 5's form)",
        "roofline": roofline,
        "kernel_ms_per_slice": {k: round(v[0], 4) for k, v in fam.items()},
    }
    line["forwards_run"] = None        # filled in just before the line is printed
    if cfg["fft_bytes"]:
        roof_fft = {"bound": "hbm", "kernel": "cine::imgdc200_kernel + imgdc_sum_kernel (x6: sens_expand + FFT2 + DC + IFFT2 + sens_reduce on the "
                                              "coil-combined image), col200_kernel + row200_reduce_kernel (first reduce, zero-filled term, sens prologue)",
                    "mode": roofline["mode_isolated"], "achieved": cfg["fft_bytes"] / (fft_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "algorithmic_bytes": cfg["fft_bytes"], "traffic": None, "ms_per_slice": fft_ms,
                    "note": "achieved = SURVEY 8(d) bytes at the reference's module boundaries / kernel time; the image-space chain "
                            "moves far fewer bytes than that (traffic), so this is the step's speed in the survey's units, not a bandwidth: "
                            "achieved_traffic_gbs is the bandwidth (PMC bytes / kernel time)"}
        roof_fft["frac"] = roof_fft["achieved"] / roof_fft["peak"]
        roof_fft["frac_survey_units"] = roof_fft["frac"]            # SURVEY 8(d) bytes / kernel time: the step's speed, not a bandwidth
        roof_fft["frac_real_bandwidth"] = None                      # PMC bytes / kernel time: the bandwidth the kernels really run at
        if tr and "fft_col_pass" in tr:
            roof_fft["traffic"] = (tr.get("fft_col_pass", {}).get("hbm_MB_per_slice", 0.0) + tr.get("fft_row_pass", {}).get("hbm_MB_per_slice", 0.0)) * 1e6 or None
            roof_fft["traffic_unit"] = f"HBM bytes per slice over all FFT / DC passes, PMC, {tr_src}"
            if roof_fft["traffic"]:
                roof_fft["achieved_traffic_gbs"] = roof_fft["traffic"] / (fft_ms * 1e-3) / 1e9
                roof_fft["frac_traffic"] = roof_fft["achieved_traffic_gbs"] / HBM_PEAK_GBS
                roof_fft["frac_real_bandwidth"] = roof_fft["frac_traffic"]
        line["roofline_fft_dc"] = roof_fft
    # ---- parity of a GRAPH-REPLAYED output (stream 0's slice) and the CPU baseline
    chk = wl.replayed_output()
    ex0 = wl.exs[0][0]
    tgt_d = ex0["target"].to(dev)
    try:
        from cine_hip import ops
        m = ops.image_metrics(tgt_d[0], chk[0])
        line["ssim"] = {"value": float(m["ssim"]), "nmse": float(m["nmse"]), "psnr": float(m["psnr"]),
                        "note": "device kernel (cine_image_metrics), graph-replayed output vs the synthetic target; untrained weights"}
    except (ImportError, AttributeError):
        pass
    best_threads = 16
    if not args.no_cpu_baseline and world == 1:        # the host-core baseline is timed on rank 0 of the 1-GPU run only
        ref_out, cb = cpu_baseline(cfg, ex0, args.cpu_forwards, args.cpu_threads)
        best_threads = cb["cores"]
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
    # ---- the default 1-GPU line also carries the other BASELINE configurations and the training step (shorter runs)
    if world == 1 and args.config == 2 and not args.headline_only:
        wl.release()
        line["other_configs"] = {}
        for cid in (3, 4, 5):
            try:
                line["other_configs"][str(cid)] = measure_other_config(cid, args, dev, best_threads)
            except Exception as e:                                        # pragma: no cover
                line["other_configs"][str(cid)] = {"error": f"{type(e).__name__}: {e}"}
        try:
            line["train_step"] = measure_training_step(args, dev, best_threads)
        except Exception as e:                                            # pragma: no cover
            line["train_step"] = {"error": f"{type(e).__name__}: {e}"}
        line["train_step_other_configs"] = {}     # XT-XPDNet (MWCNN backward), 3D CineNet, CRNN-VarNet (back-propagation through time): GPU only
        for cid in (3, 4, 5):
            try:
                line["train_step_other_configs"][str(cid)] = measure_training_step(args, dev, best_threads, cfg_id=cid, cpu=False)
            except Exception as e:                                        # pragma: no cover
                line["train_step_other_configs"][str(cid)] = {"error": f"{type(e).__name__}: {e}"}
    line["forwards_run"] = getattr(wl, "nforward", None)
    print(json.dumps(line))
    leave_together(use_dist, lambda: dist.barrier(device_ids=[local]))


if __name__ == "__main__":
    main()

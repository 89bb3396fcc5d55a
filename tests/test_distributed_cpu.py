"""N > 1 path on CPU: world_size-2 gloo, slice sharding + all-gather volume assembly."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _fake_recon(slice_id: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(1000 + slice_id)
    return torch.rand(3, 8, 6, generator=g)          # stands in for a (t, h, w) reconstruction


def _worker(rank, world, port, n_slices, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from cine_hip import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.slice_indices(n_slices, rank, world)
    per = shard.padded_count(n_slices, world)
    local = torch.zeros(per, 3, 8, 6)
    for j, s in enumerate(mine):
        local[j] = _fake_recon(s)
    vol = shard.assemble_volume(local, n_slices)
    want = torch.stack([_fake_recon(s) for s in range(n_slices)])
    q.put((rank, bool(torch.equal(vol, want)), mine))
    dist.barrier()
    dist.destroy_process_group()


def test_slice_sharding_and_assembly_gloo_world2():
    ctx = mp.get_context("spawn")
    for n_slices, port in ((7, 29611), (4, 29612)):
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, n_slices, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert all(ok for _, ok, _ in res)
        owned = sorted(i for _, _, mine in res for i in mine)
        assert owned == list(range(n_slices))            # every slice reconstructed by exactly one rank


def _train_worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from cine_hip import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(7 + rank)                                # DIFFERENT weights per rank: the constructor broadcasts rank 0's
    shared = torch.nn.Conv2d(2, 3, 3, padding=1)
    unused = torch.nn.Conv2d(1, 1, 1)                          # a sub-network this step does not use (VarNet with explicit sens_maps)
    net = torch.nn.ModuleList([shared, shared, torch.nn.Conv2d(3, 1, 1), unused])     # an aliased module, like the cascades' networks
    versions = [p._version for p in net.parameters()]
    sync = shard.GradientAllReduce(net)
    # the broadcast writes the parameters themselves: their version counters move, so packed-weight caches keyed on
    # (data_ptr, _version) that a forward BEFORE this constructor filled are invalidated on every rank
    bumped = all(p._version > v for p, v in zip(net.parameters(), versions))
    assert len(sync.params) == 6
    w0 = [torch.empty_like(p) for p in sync.params]
    for w, p in zip(w0, sync.params):
        w.copy_(p.detach())
        dist.broadcast(w, src=0)
    same_start = all(torch.equal(w, p.detach()) for w, p in zip(w0, sync.params))   # rank 0's values everywhere

    def loss_of(r):
        g = torch.Generator().manual_seed(50 + r)
        x = torch.randn(1, 2, 6, 5, generator=g)
        return net[2](net[1](x) + net[0](x)).square().mean()
    loss_of(rank).backward()
    sync()                                                                      # the unused parameters have grad None: no error
    used = [p for p in sync.params if p.grad is not None]
    got = [p.grad.clone() for p in used]
    net.zero_grad()
    (sum(loss_of(r) for r in range(world)) / world).backward()                 # the same average on one rank
    ok = bumped and same_start and len(used) == 4 and all(torch.allclose(a, p.grad, rtol=1e-5, atol=1e-7) for a, p in zip(got, used))
    # ranks that DISAGREE about which parameters are unused: only rank 0 runs the extra sub-network; every rank must end up with the
    # same (averaged) gradient for it, or the optimiser would update it on one replica only
    net.zero_grad(set_to_none=True)
    extra = unused(torch.ones(1, 1, 2, 2)).sum() if rank == 0 else 0.0
    (loss_of(rank) + extra).backward()
    sync()
    gu = [p.grad for p in unused.parameters()]
    ok = ok and all(g is not None for g in gu)
    if ok:
        want_w = torch.full_like(gu[0], 4.0 / world)          # d(sum of the conv over four ones)/dw = 4 on rank 0, nothing elsewhere
        want_b = torch.full_like(gu[1], 4.0 / world)
        ok = torch.allclose(gu[0], want_w) and torch.allclose(gu[1], want_b)
    try:                                                                        # broadcast=False only checks: equal now ...
        shard.GradientAllReduce(net, broadcast=False)
        with torch.no_grad():
            net[2].bias.add_(float(rank))                                       # ... and different after this
        try:
            shard.GradientAllReduce(net, broadcast=False)
            ok = False
        except RuntimeError:
            pass
    except RuntimeError:
        ok = False
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_all_reduce_averages_over_ranks_gloo_world2():
    """Data-parallel training: per-rank gradients -> one flat all-reduce -> the mean over the ranks' slices, aliased parameters once."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, 29621, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_single_process_assembly_is_identity():
    from cine_hip import shard
    x = torch.rand(5, 2, 3)
    assert torch.equal(shard.assemble_volume(x, 5), x)
    assert shard.slice_indices(10, 1, 4) == [1, 5, 9] and shard.padded_count(10, 4) == 3


def _bench(*argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                       timeout=300, cwd="/tmp")
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_flag_starts_that_many_ranks_gloo_world2():
    """`python bench.py --gpus 2` with WORLD_SIZE unset starts 2 ranks itself (child processes, no exec) and the real
    timed region -- barrier, K steps, slice-sharded volume assembly by all-gather, barrier, max over ranks -- runs on gloo
    with a stand-in for the forward: the assembled volume holds every slice of both ranks in slice order."""
    r, line = _bench("--gpus", "2", "--steps", "5", "--warmup", "1", "--selftest-cpu")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["volume_ok"] and line["volume_slices"] == 10
    assert line["valid"] is False and line["value"] is None            # the self-test line can never pass for a measurement
    # main()'s ordering behind the timed region (bench.leave_together): rank 1 waits at the last barrier until rank 0 has done its host-side work
    # (exit code 4 if it had left early), and an N > 1 run skips the single-GPU extras (one slice alone, the 60-s sustained region)
    assert line["single_gpu_extras"] is False


def test_bench_single_gpu_extras_only_when_asked_for_at_n_greater_than_1():
    r, line = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--selftest-cpu", "--sustained-seconds", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["single_gpu_extras"] is True


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    r, line = _bench("--gpus", "2", "--selftest-cpu", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and line is None and "refusing" in (r.stderr + r.stdout)


def test_bench_without_a_gpu_fails_loudly():
    if torch.cuda.is_available():
        return
    r, line = _bench("--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0 and line is None and "no CPU fallback" in (r.stderr + r.stdout)


def test_bench_selftest_eight_ranks_gloo():
    """The driver's N = 8 launch shape on CPU: eight ranks under torch.distributed.run, slice sharding + all-gather assembly of all
    8 K slices, every rank's own timed-region seconds on rank 0's line (a straggler is visible in SCALE_rNN.json)."""
    r, line = _bench("--gpus", "8", "--steps", "3", "--warmup", "1", "--selftest-cpu", env_extra={"OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8 and line["volume_ok"] and line["volume_slices"] == 24
    assert len(line["per_rank_timed_region_s"]) == 8 and line["timed_region_s"] == max(line["per_rank_timed_region_s"])


def _fake_sysfs(root, gpus, local_cpulists=None):
    """gpus: list of (bus, numa_node); node 0 of the KFD topology is the CPU (simd_count 0), as on a real box.  local_cpulists: per-GPU
    text of the PCI function's local_cpulist (None = the file does not exist)."""
    import pathlib
    root = pathlib.Path(root)
    nodes = root / "class/kfd/kfd/topology/nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (bus, numa) in enumerate(gpus):
        (nodes / str(i + 1)).mkdir()
        (nodes / str(i + 1) / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\n")
        dev = root / "bus/pci/devices" / f"0000:{bus:02x}:00.0"
        dev.mkdir(parents=True)
        (dev / "numa_node").write_text(f"{numa}\n")
        if local_cpulists and local_cpulists[i] is not None:
            (dev / "local_cpulist").write_text(local_cpulists[i] + "\n")
    for n, cl in ((0, "0-3,64-67"), (1, "4-7")):
        d = root / f"devices/system/node/node{n}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cl + "\n")


def test_rank_numa_affinity_from_sysfs(tmp_path):
    """bench.py pins a rank to its GPU's NUMA node from sysfs alone (no GPU call): KFD order -> PCI function -> numa_node -> cpulist."""
    sys.path.insert(0, ROOT)
    import bench
    _fake_sysfs(tmp_path, [(0x05, 0), (0x85, 1), (0xc5, -1)])
    assert bench.gpu_numa_cpus(0, str(tmp_path)) == (0, {0, 1, 2, 3, 64, 65, 66, 67})
    assert bench.gpu_numa_cpus(1, str(tmp_path)) == (1, {4, 5, 6, 7})
    assert bench.gpu_numa_cpus(2, str(tmp_path)) == (None, None)          # numa_node -1 and no local_cpulist either: no pinning
    assert bench.gpu_numa_cpus(7, str(tmp_path)) == (None, None)          # no such GPU
    assert bench.gpu_numa_cpus(0, str(tmp_path / "missing")) == (None, None)
    # applied in a child process (the affinity of the test runner itself stays untouched)
    import subprocess
    code = ("import os, sys, json; sys.path.insert(0, %r); import bench; before = os.sched_getaffinity(0); "
            "info = bench.pin_rank_to_gpu_numa(1, 2, %r); print(json.dumps([sorted(before), sorted(os.sched_getaffinity(0)), info]))" % (ROOT, str(tmp_path)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1500:]
    import json
    before, after, info = json.loads(out.stdout.strip().splitlines()[-1])
    want = sorted(set(before) & {4, 5, 6, 7})
    if want and want != before:
        assert after == want and info["applied"] and info["numa_node"] == 1
    else:
        assert after == before and not info["applied"] and info["why"]


def test_rank_numa_affinity_falls_back_to_local_cpulist(tmp_path):
    """A box whose firmware does not name the GPU's node (numa_node = -1, what the driver's GPU boxes report: `cpu_affinity.applied`
    was false there): the PCI function's local_cpulist gives the CPUs next to the GPU, marked as node -1."""
    sys.path.insert(0, ROOT)
    import bench
    _fake_sysfs(tmp_path, [(0x05, -1), (0x85, -1)], local_cpulists=["0-3", ""])
    assert bench.gpu_numa_cpus(0, str(tmp_path)) == (-1, {0, 1, 2, 3})
    assert bench.gpu_numa_cpus(1, str(tmp_path)) == (None, None)          # an empty mask: nothing to pin to
    import json
    import subprocess
    code = ("import os, sys, json; sys.path.insert(0, %r); import bench; before = os.sched_getaffinity(0); "
            "info = bench.pin_rank_to_gpu_numa(0, 2, %r); print(json.dumps([sorted(before), sorted(os.sched_getaffinity(0)), info]))" % (ROOT, str(tmp_path)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1500:]
    before, after, info = json.loads(out.stdout.strip().splitlines()[-1])
    want = sorted(set(before) & {0, 1, 2, 3})
    if want and want != before:
        assert after == want and info["applied"] and info["numa_node"] == -1 and "local_cpulist" in info["source"]
    else:
        assert after == before and not info["applied"] and info["why"]

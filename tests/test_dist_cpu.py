"""CPU, world_size 2, gloo: the N > 1 control flow of bench.py (replicas: barrier, max-over-ranks time,
whole-job throughput) without GPUs."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    steps = 10
    elapsed = 1.0 + rank            # rank 1 is the slow replica
    dist.barrier()
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    value = bench.aggregate_throughput([steps] * world, [float(t)])
    out[rank] = (float(t), value)
    dist.barrier()
    dist.destroy_process_group()


def _worker_records(rank, world, port, out):
    """The per-rank diagnostic records of bench.py (what an 8-GPU scaling curve would be debugged from) through the same
    gather main() uses, over gloo."""
    import json
    import types
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    cache = types.SimpleNamespace(v_cache_cpu=torch.zeros(4, 1024, dtype=torch.bfloat16))
    model = types.SimpleNamespace(batch_size=1, kv_cache=cache)
    head = dict(steps=20, elapsed_local=0.1 * (rank + 1), hit_rate=0.5 + 0.1 * rank)
    rec = bench.rank_record(rank, rank, model, head, t_build=3.0 + rank, numa_gpu=rank)
    recs = bench.gather_rank_records(rec, world)
    out[rank] = json.dumps(recs)            # (must be JSON-serialisable: it goes into the bench line)
    dist.barrier()
    dist.destroy_process_group()


def test_per_rank_records_are_gathered_to_every_rank():
    import json
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_records, args=(2, port, out), nprocs=2, join=True)
        out = dict(out)
    assert out[0] == out[1]
    recs = json.loads(out[0])
    assert [r["rank"] for r in recs] == [0, 1] and [r["gpu_index"] for r in recs] == [0, 1]
    assert recs[0]["tokens_per_s"] == 200.0 and recs[1]["tokens_per_s"] == 100.0          # 20 tokens / own elapsed time
    assert recs[1]["ms_per_step"] == 10.0 and recs[1]["chunk_hit_rate"] == 0.6 and recs[1]["state_build_s"] == 4.0
    for r in recs:
        assert r["gpu_numa_node"] == r["rank"] and "process_numa_node" in r and r["cpus_allowed"] >= 1
        nodes = r["pinned_v_numa_node"]                            # first / middle / last page of the host V table
        assert isinstance(nodes, list) and len(nodes) == 3 and all(n is None or n >= 0 for n in nodes)


def test_numa_queries_do_not_raise():
    import bench
    t = torch.zeros(4096, dtype=torch.uint8)
    n = bench.numa_node_of_address(t.data_ptr())
    assert n is None or n >= 0
    p = bench.process_numa_node()
    assert p is not None and p >= 0           # (round 3: null on the driver's box - Python 3.10 has no os.sched_getcpu)
    c = bench.physical_cores()
    assert c is None or 1 <= c <= os.cpu_count()


def test_replica_aggregation_two_ranks():
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
        out = dict(out)
    assert out[0][0] == out[1][0] == 2.0          # both ranks agree on the max time
    assert out[0][1] == out[1][1] == 2 * 10 / 2.0  # 2 replicas x 10 tokens / slowest rank


def test_aggregate_throughput_single():
    import bench
    assert bench.aggregate_throughput([64], [0.5]) == 128.0


# ----------------------------------------------------------------------------------------------------------------------
# bench.main()'s own control flow on two ranks (round 4): the rank gating (only rank 0 measures the extras and prints),
# the barrier / MAX all-reduce inside run_decode's timed region, the per-rank gather, the final barrier - over gloo with
# a stub model in place of the GPU work (the real main(), the real argument parsing, the real JSON assembly).
# ----------------------------------------------------------------------------------------------------------------------
def _stub_bench(bench, rank):
    import types
    cache = types.SimpleNamespace(resident_sets=256, select_sets=256, block_num=8, _early=None, rank=160, sparse_end=2496,
                                  k_landmark=torch.zeros(1, 1, 8, 16, 128), v_cache_cpu=torch.zeros(4, 1024, dtype=torch.bfloat16))
    cfg = types.SimpleNamespace(name="stub-llama", vocab_size=128)
    model = types.SimpleNamespace(batch_size=1, kv_cache=cache, num_layers=2, attn_mode="shadowkv_cpu", cfg=cfg)
    calls = dict(build=0, decode=0, score=0)

    def build_model(workload, args, rank_, dev, **kw):
        calls["build"] += 1
        return model, cfg, 124928, 2048, 1.5 + rank_

    def run_decode(model_, args, ctx, steps, warmup, walk_step, seed, world=1, pin_hit=None):
        # the timed region's collectives exactly as bench.run_decode issues them (barrier, barrier, MAX of the elapsed time)
        calls["decode"] += 1
        dist.barrier()
        elapsed_local = 0.5 * (rank + 1)                 # rank 1 is the slow replica
        dist.barrier()
        t = torch.tensor([elapsed_local], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return dict(value=bench.aggregate_throughput([steps] * world, [float(t)]), ms_per_step=float(t) / steps * 1e3,
                    hit_rate=0.6 + 0.1 * rank, mode=args.mode, slack_ring=False, elapsed_local=elapsed_local, steps=steps)

    def measure_score_kernel(model_, iters=3):
        calls["score"] += 1
        if os.environ.get("SKV_TEST_BREAK_SCORE"):
            raise RuntimeError("a reporting leg broke")
        return dict(kernel="stub", us_per_launch=8.0, algorithmic_bytes=32_000_000, gbs=4000.0)

    bench.DIST_BACKEND = "gloo"
    bench.setup_device = lambda local_rank: ("cpu", local_rank)
    bench.build_model, bench.run_decode, bench.measure_score_kernel = build_model, run_decode, measure_score_kernel
    return calls


def _worker_main(rank, world, port, out):
    import contextlib
    import io
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import bench
    calls = _stub_bench(bench, rank)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "10", "--warmup", "2"])
    # main() has destroyed the process group behind its final barrier: every rank got here
    out[rank] = (buf.getvalue(), dict(calls), dist.is_initialized())


def test_bench_main_on_two_ranks_prints_one_line_on_rank_zero():
    import json
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_main, args=(2, port, out), nprocs=2, join=True)
        out = dict(out)
    (text0, calls0, init0), (text1, calls1, init1) = out[0], out[1]
    assert text1 == "" and calls1 == dict(build=1, decode=1, score=0)       # rank 1: decodes, prints and measures nothing else
    assert calls0 == dict(build=1, decode=1, score=1) and not init0 and not init1
    lines = [l for l in text0.splitlines() if l.strip()]
    assert len(lines) == 1, text0                                           # exactly ONE JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 10 and rec["warmup"] == 2 and rec["scaling"] == "weak"
    assert rec["metric"].startswith("decode tokens/sec @122K ctx") and rec["unit"] == "tokens/s" and rec["vs_baseline"] is None
    assert rec["value"] == 2 * 10 / 1.0 and rec["ms_per_step"] == 100.0     # 2 replicas x 10 tokens / the slowest rank's 1.0 s
    assert [r["rank"] for r in rec["per_rank"]] == [0, 1]
    assert [r["tokens_per_s"] for r in rec["per_rank"]] == [20.0, 10.0]     # each rank's OWN rate
    assert rec["per_rank"][1]["state_build_s"] == 2.5 and rec["per_rank"][1]["chunk_hit_rate"] == 0.7
    assert "replicas x2" in rec["config"]["parallelism"] and rec["roofline"]["frac"] == 0.5
    for key in ("cpu_baseline", "secondary", "batched", "hit_rate_sweep"):  # N = 1 extras are not run at N > 1
        assert key not in rec


def _plain_bench(extra_env=None, args=("--gpus", "2", "--steps", "10", "--warmup", "2")):
    """`bench.main([...])` in a fresh interpreter with NO launcher variables: the parent must start its own replicas."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    code = ("import sys, os; sys.path.insert(0, %r); import bench; "
            "bench.CHILD_ENTRY = [sys.executable, os.path.join(%r, 'tests', '_bench_stub_child.py')]; "
            "import torch; bench.main(%r); "
            "assert not torch.cuda.is_initialized()" % (ROOT, ROOT, list(args)))
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)


def test_plain_bench_gpus_2_launches_its_own_replicas():
    """`python bench.py --gpus 2` with no torch.distributed.run around it.  The parent starts two ranks (gloo
    and the stub seams here), relays rank 0's single JSON line and exits 0 - without initialising the GPU runtime itself."""
    import json
    r = _plain_bench()
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 10 and rec["scaling"] == "weak"
    assert [x["rank"] for x in rec["per_rank"]] == [0, 1]
    assert rec["value"] == 2 * 10 / 1.0


def test_plain_bench_returns_the_worst_child_exit_code():
    """A rank that dies must not leave the other waiting in a barrier: the launcher stops it and reports the failure."""
    r = _plain_bench({"SKV_TEST_FAIL_RANK": "1"})
    assert r.returncode == 7 and r.stdout.strip() == "" and "rank 1 exited with code 7" in r.stderr


def test_a_failing_reporting_leg_still_leaves_the_line():
    """Everything behind the timed headline runs guarded: a leg that raises is reported in the line (`bench_error`), the headline
    value is still printed and the exit code stays 0."""
    import json
    r = _plain_bench({"SKV_TEST_BREAK_SCORE": "1"})
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["value"] == 2 * 10 / 1.0 and rec["n_gpus"] == 2 and "a reporting leg broke" in rec["bench_error"]
    assert "Traceback" in r.stderr


def test_launcher_form_with_a_wrong_world_size_is_refused():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 2 and r.stdout.strip() == "" and "WORLD_SIZE is 1" in r.stderr


def test_gpu_numa_cpulists_from_the_kfd_topology(tmp_path):
    """The launcher reads each GPU's NUMA CPU list from sysfs (KFD topology -> PCI address -> numa_node) without a HIP call."""
    import bench
    root = tmp_path
    nodes = root / "class/kfd/kfd/topology/nodes"
    spec = {0: (0, 0, 0), 1: (0, 0, 0), 2: (256, 0x0500, 0), 3: (256, 0x1500 | (0 << 3), 1)}      # two CPU nodes, two GPUs
    for n, (simd, loc, dom) in spec.items():
        (nodes / str(n)).mkdir(parents=True)
        (nodes / str(n) / "properties").write_text(f"cpu_cores_count 0\nsimd_count {simd}\nlocation_id {loc}\ndomain {dom}\n")
    for bdf, node in (("0000:05:00.0", 1), ("0001:15:00.0", 0)):
        (root / "bus/pci/devices" / bdf).mkdir(parents=True)
        (root / "bus/pci/devices" / bdf / "numa_node").write_text(f"{node}\n")
    for node, lst in ((0, "0-31,64-95"), (1, "32-63,96-127")):
        (root / f"devices/system/node/node{node}").mkdir(parents=True)
        (root / f"devices/system/node/node{node}/cpulist").write_text(lst + "\n")
    assert bench.gpu_numa_cpus_from_sysfs(2, str(root)) == ["32-63,96-127", "0-31,64-95"]
    assert bench.gpu_numa_cpus_from_sysfs(3, str(root)) == ["32-63,96-127", "0-31,64-95", None]
    assert bench.gpu_numa_cpus_from_sysfs(2, str(tmp_path / "missing")) == [None, None]


def test_process_numa_node_falls_back_to_the_node_cpulists(tmp_path):
    """The driver's N = 1 box in round 3 reported process_numa_node null: its /sys/devices/system/cpu/cpuN has no nodeM link.
    The fallback scans /sys/devices/system/node/node*/cpulist; a tree with neither reports node 0 for an online CPU."""
    import bench
    cpu = bench.current_cpu()
    assert cpu is not None and cpu in os.sched_getaffinity(0)
    bench.current_cpu = lambda: cpu                                       # (the scheduler may move the process between calls)
    root = tmp_path / "system"
    (root / "cpu" / f"cpu{cpu}").mkdir(parents=True)
    for n, lst in ((0, f"{cpu + 1}-{cpu + 3}"), (3, f"0-{cpu}" if cpu else "0"), (10, f"{cpu + 4}")):
        (root / "node" / f"node{n}").mkdir(parents=True)
        (root / "node" / f"node{n}" / "cpulist").write_text(lst + "\n")
    assert bench.process_numa_node(str(root)) == 3
    (root / "cpu" / f"cpu{cpu}" / "node7").mkdir()                        # the direct link wins when it exists
    assert bench.process_numa_node(str(root)) == 7
    bare = tmp_path / "bare"
    (bare / "cpu" / f"cpu{cpu}").mkdir(parents=True)
    (bare / "cpu" / "online").write_text(f"0-{cpu + 8}\n")
    assert bench.process_numa_node(str(bare)) == 0
    assert bench.process_numa_node(str(tmp_path / "missing")) is None
    assert bench._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}


def test_cgroup_cpu_quota_and_core_list(tmp_path):
    """cpu_baseline's thread count: min(physical cores allowed, cgroup CPU quota) - VERDICT r4 #7."""
    import bench
    root = tmp_path / "cg"
    (root / "pod" / "job").mkdir(parents=True)
    (root / "cpu.max").write_text("max 100000\n")
    (root / "pod" / "cpu.max").write_text("6400000 100000\n")
    (root / "pod" / "job" / "cpu.max").write_text("3200000 100000\n")
    proc = tmp_path / "cgroup"
    proc.write_text("0::/pod/job\n")
    assert bench.cgroup_cpu_quota(str(proc), str(root)) == 32.0                 # the tightest limit on the path
    proc.write_text("0::/pod\n")
    assert bench.cgroup_cpu_quota(str(proc), str(root)) == 64.0
    proc.write_text("0::/\n")
    assert bench.cgroup_cpu_quota(str(proc), str(root)) is None                 # "max": no limit
    v1 = tmp_path / "v1"
    (v1 / "cpu" / "docker").mkdir(parents=True)
    (v1 / "cpu" / "docker" / "cpu.cfs_quota_us").write_text("1650000\n")
    (v1 / "cpu" / "docker" / "cpu.cfs_period_us").write_text("100000\n")
    proc.write_text("4:cpu,cpuacct:/docker\n")
    assert bench.cgroup_cpu_quota(str(proc), str(v1)) == 16.5
    assert bench.cgroup_cpu_quota(str(tmp_path / "none"), str(tmp_path / "none")) is None
    cpus = tmp_path / "cpu"
    for cpu, (pkg, core) in {0: (0, 0), 1: (0, 1), 2: (1, 0), 3: (1, 1), 4: (0, 0), 5: (0, 1), 6: (1, 0), 7: (1, 1)}.items():
        (cpus / f"cpu{cpu}" / "topology").mkdir(parents=True)
        (cpus / f"cpu{cpu}" / "topology" / "physical_package_id").write_text(f"{pkg}\n")
        (cpus / f"cpu{cpu}" / "topology" / "core_id").write_text(f"{core}\n")
    assert bench.one_cpu_per_core(range(8), str(cpus)) == [0, 1, 2, 3]           # SMT siblings 4..7 counted once
    assert bench.one_cpu_per_core([1, 5, 6, 7], str(cpus)) == [1, 6, 7]
    assert bench.one_cpu_per_core([3, 1], str(tmp_path / "missing")) == [1, 3]


def test_oracle_thread_binding_round_trip():
    import oracle
    allowed = sorted(os.sched_getaffinity(0))
    n = min(4, len(allowed))
    before = oracle.num_threads()
    try:
        oracle.set_num_threads(n)
        assert oracle.bind_threads(allowed[:n]) == n
        assert sorted(oracle.thread_cpus()) == allowed[:n]                       # one thread per listed CPU
        assert oracle.bind_threads(allowed, whole_set=True) == n
    finally:
        os.sched_setaffinity(0, allowed)
        oracle.set_num_threads(before)
    t = torch.arange(100000, dtype=torch.float32)
    assert torch.equal(oracle.first_touch_clone(t), t)

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
    assert p is None or p >= 0
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

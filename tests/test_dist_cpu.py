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

"""world_size-2 (gloo, CPU) test of bench.py's multi-rank bookkeeping: layer-parallel
sharding has no data-path collective; only the timing is reduced (max over ranks)."""
import os
import socket
import sys

import pytest
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


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    mine = bench.shard_ring(10, rank, world)
    # gather every rank's shard: disjoint and complete
    got = [None] * world
    dist.all_gather_object(got, mine)
    # rank-dependent fake timings: everybody must end with the max
    wall, ev = bench.reduce_times(1.0 + rank, 10.0 * (world - rank), dist, "cpu")
    val = bench.job_throughput_gbps(world, bench.alg_bytes(8192), 32, 200, wall)
    dist.barrier()
    q.put((rank, got, wall, ev, val))
    dist.destroy_process_group()


def test_layer_parallel_bookkeeping_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, wall, ev, val in res:
        flat = sorted(i for shard in got for i in shard)
        assert flat == list(range(10))                       # complete, disjoint
        assert wall == 2.0 and ev == 20.0                    # max over ranks
        assert val == pytest.approx(2 * 16850944 * 32 * 200 / 2.0 / 1e9)
    assert res[0][4] == res[1][4]


def test_alg_bytes_matches_survey():
    import bench
    assert bench.alg_bytes(4096) == 4235264 and bench.alg_bytes(8192) == 16850944

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


def test_board_sampler_reads_the_card_of_this_device(tmp_path, monkeypatch):
    """bench.py's clock / power sampler: the hwmon files of the card whose PCI address is the HIP device's (a box of the
    pool shows every GPU of its node in /sys); nothing to read -> summaries are None, never an exception"""
    import time
    import bench
    assert bench.sysfs_card_of_device(0) is None        # no GPU here
    with bench.SclkSampler(0) as s:
        time.sleep(0.05)
    assert s.summary() is None and s.power_summary() is None
    # a fake card: freq1_input in Hz, power1_input / power1_cap in microwatts
    hw = tmp_path / "card7" / "device" / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (hw / "freq1_input").write_text("1620000000\n")
    (hw / "power1_input").write_text("1399000000\n")
    (hw / "power1_cap").write_text("1400000000\n")
    monkeypatch.setattr(bench, "sysfs_card_of_device", lambda i=0: str(tmp_path / "card7" / "device"))
    with bench.SclkSampler(0) as s:
        time.sleep(0.1)
    assert s.summary()["median_mhz"] == 1620.0
    p = s.power_summary()
    assert p["median_w"] == 1399.0 and p["cap_w"] == 1400.0

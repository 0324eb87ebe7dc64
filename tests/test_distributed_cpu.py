"""CPU: the N>1 sharding / status-gather path with two gloo ranks (no GPU involved)."""

import os
import socket

import torch
import torch.multiprocessing as mp

from biahub_amd import parallel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w = parallel.init("gloo")
    assert (r, w) == (rank, world)
    positions = [f"{row}/{col}/0" for row in "AB" for col in range(1, 6)]  # 10 positions

    def func(pos):
        if pos == "B/3/0":
            raise RuntimeError("corrupt position")
        return 1000.0

    st = parallel.process_positions(positions, func, r, w)
    parallel.barrier()
    slowest = parallel.max_over_ranks(float(rank + 1))
    rows = parallel.gather_stats(st)
    q.put((rank, parallel.shard_positions(positions, r, w), st.n_done, st.n_failed, slowest,
           [(x.n_done, x.n_failed, x.voxels) for x in rows]))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = [o[1] for o in out]
    assert sorted(shards[0] + shards[1]) == sorted(f"{row}/{col}/0" for row in "AB" for col in range(1, 6))
    assert not set(shards[0]) & set(shards[1])  # disjoint: no position is processed twice
    assert shards[0][0] == "A/1/0" and shards[1][0] == "A/2/0"  # round-robin
    assert sum(o[2] for o in out) == 9 and sum(o[3] for o in out) == 1  # one failure counted, plate continues
    assert all(o[4] == 2.0 for o in out)  # max over ranks reaches everyone
    assert out[0][5] == out[1][5] and sum(v for _, _, v in out[0][5]) == 9000.0  # identical gathered status


def test_single_process_defaults():
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert parallel.world_info() == (0, 0, 1)
    assert parallel.shard_positions(range(5), 0, 1) == [0, 1, 2, 3, 4]
    assert parallel.max_over_ranks(3.5) == 3.5
    st = parallel.process_positions(range(3), lambda p: 10, 0, 1)
    assert (st.n_done, st.n_failed, st.voxels) == (3, 0, 30.0)
    assert parallel.gather_stats(st)[0].n_done == 3

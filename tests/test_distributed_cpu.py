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
    cards = parallel.gather_objects({"rank": rank, "pci_bus_id": f"0000:{rank:02x}:00.0"})  # bench.py's "rccl" block
    assert [c["rank"] for c in cards] == list(range(world)) and len({c["pci_bus_id"] for c in cards}) == world
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


def _bind_worker(rank, world, port, q):
    """Rank r must make GPU r its current device before anything else touches a GPU (device count mocked: 8)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), BH_DIST_BACKEND="gloo")
    calls = []
    torch.cuda.device_count = lambda: 8
    torch.cuda.set_device = lambda d: calls.append(torch.device(d))
    r, w = parallel.init()  # what every CLI command calls first (cli._create_plate_once / _run_positions)
    dev = parallel.bound_device()
    parallel.barrier()  # gloo: must not try to run on the (mocked) GPU
    rows = parallel.gather_stats(parallel.RankStats(n_done=rank + 1))
    q.put((rank, str(dev), [str(c) for c in calls], [x.n_done for x in rows]))
    torch.distributed.destroy_process_group()


def test_two_ranks_bind_their_own_gpu():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bind_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, dev, calls, done in out:
        assert dev == f"cuda:{rank}" and calls == [f"cuda:{rank}"]  # bound exactly once, to LOCAL_RANK
        assert done == [1, 2]


def test_bind_device_wraps_and_handles_no_gpu(monkeypatch):
    monkeypatch.setenv("LOCAL_RANK", "5")
    seen = []
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: seen.append(str(d)))
    assert str(parallel.bind_device()) == "cuda:1" and seen == ["cuda:1"]  # more ranks than GPUs wrap around
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)
    assert parallel.bind_device() is None and parallel.bound_device() is None
    assert parallel._comm_device(None) == "cpu"


def _plate_worker(rank, world, port, store, q):
    """Both ranks run a CLI command's plate set-up at once: rank 0 creates, rank 1 waits; then both update attrs."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), BH_DIST_BACKEND="gloo")
    from biahub_amd import cli, io

    keys = [(row, str(col), "0") for row in "AB" for col in range(1, 5)]
    cli._create_plate_once(store, keys, ["ch0", "ch1"], (1, 2, 4, 8, 8))
    n = 0
    for k in keys:  # every rank can open every position straight after the barrier
        with io.open_ome_zarr(os.path.join(store, *k)) as p:
            n += int(p.data.shape == (1, 2, 4, 8, 8))
    for i in range(50):  # concurrent metadata writers on one group must not trip over a shared temp name
        io._write_json(io.Path(store) / f"probe_{rank}.json", {"i": i})
        io._write_json(io.Path(store) / "shared.json", {"rank": rank, "i": i})
    q.put((rank, n))
    torch.distributed.destroy_process_group()


def test_two_process_plate_creation(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    store = str(tmp_path / "plate.zarr")
    procs = [ctx.Process(target=_plate_worker, args=(r, world, port, store, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out == [(0, 8), (1, 8)]
    import json

    assert json.loads((tmp_path / "plate.zarr" / "shared.json").read_text())["i"] == 49
    assert not list((tmp_path / "plate.zarr").glob("*.tmp"))


def test_process_positions_logs_failures(capsys):
    def func(pos):
        if pos == 1:
            raise RuntimeError("corrupt chunk 3/0/0")
        return 1.0

    st = parallel.process_positions(range(3), func, 0, 1)
    assert (st.n_done, st.n_failed) == (2, 1) and st.first_error == "1: RuntimeError: corrupt chunk 3/0/0"
    err = capsys.readouterr().err
    assert "position 1 failed" in err and "Traceback" in err and "corrupt chunk 3/0/0" in err


def test_single_process_defaults():
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert parallel.world_info() == (0, 0, 1)
    assert parallel.shard_positions(range(5), 0, 1) == [0, 1, 2, 3, 4]
    assert parallel.max_over_ranks(3.5) == 3.5
    st = parallel.process_positions(range(3), lambda p: 10, 0, 1)
    assert (st.n_done, st.n_failed, st.voxels) == (3, 0, 30.0)
    assert parallel.gather_stats(st)[0].n_done == 3


def _rank0_fail_worker(rank, world, port, q, tmp):
    """Rank 0's plate lay-out fails: every rank must leave the collective promptly with an error, none may hang."""
    import time

    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), BH_DIST_BACKEND="gloo")
    parallel.init()
    t0 = time.perf_counter()
    ok = parallel.rank0_first(lambda: "laid out")  # the good path returns rank 0's value there, None elsewhere
    outcome = "no error"
    try:
        from biahub_amd.io import create_empty_plate

        bad = os.path.join(tmp, "not_a_dir")  # a regular file where the store should go
        parallel.rank0_first(create_empty_plate, bad, [("A", "1", "0")], ["c"], (1, 1, 2, 4, 4))
    except parallel.Rank0Error as e:
        outcome = f"Rank0Error: {e}"
    except Exception as e:  # noqa: BLE001 - rank 0 re-raises its own exception
        outcome = f"own: {type(e).__name__}"
    q.put((rank, ok, outcome, time.perf_counter() - t0))
    torch.distributed.destroy_process_group()


def test_rank0_failure_reaches_every_rank(tmp_path):
    (tmp_path / "not_a_dir").write_text("occupied")
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank0_fail_worker, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][1] == "laid out" and out[1][1] is None
    assert out[0][2].startswith("own: ")                       # rank 0 sees its own exception
    assert out[1][2].startswith("Rank0Error: rank 0 failed")   # rank 1 is told, instead of waiting in a barrier
    assert all(o[3] < 30 for o in out)


def test_four_rank_plate_rehearsal_cli_on_cpu(tmp_path):
    """The first 8-GPU run, rehearsed with what a box without a GPU has: `torchrun --nproc-per-node 4 -m biahub_amd deskew` on an
    8-position plate over gloo, `device: cpu` (libbhcore's host deskew) — the same sharding, plate lay-out, barrier and status
    gather as on GPUs (reference fan-out: biahub/deskew.py:715-749).  One plate writer (rank 0 lays the plate out, the others
    wait), disjoint shards that cover the plate, every position right, one job-id log."""
    import json
    import os
    import re
    import subprocess
    import sys
    from pathlib import Path

    import numpy as np
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: the CLI would take it; the GPU suite runs the 2-rank form (test_cli_two_ranks_under_torchrun)")
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from test_io_cli import DESKEW_YML, make_plate

    from biahub_amd import io
    from oracle import oracle_np as O

    src = tmp_path / "in.zarr"
    keys = [(row, str(col), "0") for row in "AB" for col in (1, 2, 3, 4)]
    data = make_plate(src, positions=keys, shape=(1, 1, 12, 16, 20))
    cfg = tmp_path / "deskew.yml"
    cfg.write_text(DESKEW_YML)
    out = tmp_path / "deskewed.zarr"
    env = dict(os.environ, BH_DIST_BACKEND="gloo", PYTHONPATH=str(Path(__file__).resolve().parent.parent), OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "biahub_amd", "deskew", "-i", *[str(src.joinpath(*k)) for k in keys],
           "-c", str(cfg), "-o", str(out), "--cluster", "debug"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    shares = re.findall(r"\[rank (\d)/4\] deskew: device (\S+), (\d) of 8 position\(s\)", res.stderr)
    assert sorted(int(r) for r, _, _ in shares) == [0, 1, 2, 3] and all(n == "2" for _, _, n in shares), res.stderr
    done = re.findall(r"Deskew complete: (\S+)", res.stdout)
    assert sorted(done) == sorted(str(src.joinpath(*k)) for k in keys)  # every position exactly once: the shards are disjoint
    # one writer of the plate-level metadata: it lists all eight wells once, and the job-id log has one line per position
    meta = json.loads((out / ".zattrs").read_text()) if (out / ".zattrs").exists() else json.loads((out / "zarr.json").read_text())
    text = json.dumps(meta)
    for row, col, _ in keys:
        assert text.count(f'"{row}/{col}"') >= 1
    assert len((tmp_path / "slurm_output" / "submitit_jobs_ids.log").read_text().splitlines()) == 8
    for k in keys:
        got = io.open_ome_zarr(out.joinpath(*k)).data[0, 0]
        want = O.fast_deskew_zyx(data[k + (0, 0)].astype(np.float32), 36.17, 0.371, True, 3, "mean")
        assert np.abs(got - want).max() <= 1e-5 * want.max()

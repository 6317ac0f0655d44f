"""N > 1 path on CPU: two gloo ranks shard the chains, each runs its block (the CPU oracle stands in for
the GPU runner here -- tests may use it), and the MIN / SUM all-reduce of the summary equals the
single-process summary.  Mirrors bench.py's multi-GPU flow (one process per GPU, RCCL) without a GPU."""
import os
import socket

import numpy as np
import pytest

import mcq_amd
from oracle import oracle

abi = mcq_amd.abi
dm = mcq_amd.distributed


def _oracle_runner(N, n_steps, init_mode, schedule_params, seeds, mcmc_type="full_3d", early_stop_patience=None, trace=False):
    p = abi.make_params(N, n_steps, init_mode, schedule_params, len(seeds), mcmc_type=mcmc_type,
                        early_stop_patience=early_stop_patience, trace=trace)
    return oracle.run(p, np.asarray(seeds, dtype=np.uint32), trace=trace, states=False), 0.0


CASE = dict(N=8, n_steps=600, init_mode="random", schedule_params={"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0},
            n_runs=37, base_seed=1234, mcmc_type="board", early_stop_patience=None)


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res, summary, (lo, hi), _ = dm.run_experiment_sharded(dist=dist, runner=_oracle_runner, **CASE)
        q.put((rank, lo, hi, res["best_energy"].tolist(), summary))
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 64, 65536, 100003):
        for world in (1, 2, 3, 8):
            blocks = [dm.shard_bounds(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
            seeds = np.concatenate([dm.shard_seeds(42, n, r, world)[0] for r in range(world)])
            np.testing.assert_array_equal(seeds, abi.seeds_for(42, n))


@pytest.mark.timeout(300)
def test_two_gloo_ranks_equal_one_process():
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    res, summary, _, _ = dm.run_experiment_sharded(runner=_oracle_runner, **CASE)  # single process, no dist
    best = sum((g[3] for g in got), [])
    assert best == res["best_energy"].tolist()                      # chain r's result does not depend on the rank count
    assert got[0][4] == got[1][4] == summary                        # every rank holds the reduced summary
    assert (got[0][1], got[0][2], got[1][1], got[1][2]) == (0, 19, 19, 37)
    assert summary["n_chains"] == 37 and summary["proposed"] == 37 * 600
    assert summary["min_best"] == min(best) and abs(summary["std_best"] - float(np.std(best))) < 1e-9


def test_summary_arithmetic_survives_large_counts():
    """10^6-step runs of 65 536 chains: n * sum(x^2) exceeds 2^63 -- the variance is taken from Python integers."""
    n, best = 65536, 30
    c = np.array([n, 10**12, n * 10**6, n * best, n * best * best + 4 * n, n * 900000], dtype=np.int64)
    s = dm.summary_from_counters(c, np.array([26, 0], dtype=np.int64))
    assert s["min_best"] == 25 and s["mean_best"] == best and abs(s["std_best"] - 2.0) < 1e-12 and s["mean_steps_to_best"] == 900000
    huge = np.array([2**31 - 1, 0, 0, (2**31 - 1) * 14000, (2**31 - 1) * 14000**2, 0], dtype=np.int64)
    assert dm.summary_from_counters(huge, np.array([1], dtype=np.int64))["std_best"] == 0.0


def _table_worker(rank, world, port, q):
    import importlib.util
    import sys

    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
        t = bench.per_rank_table(torch, dist, rank, world, torch.device("cpu"), {"sweep_ms": 100.0 + rank, "reduce_ms": 0.5 * (rank + 1)})
        q.put((rank, t))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_per_rank_table_of_the_bench_line_on_two_gloo_ranks():
    """What bench.py prints for reading an N > 1 run afterwards: every rank's own times side by side, identical on every rank."""
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_table_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1] == {"reduce_ms": [0.5, 1.0], "sweep_ms": [100.0, 101.0]}

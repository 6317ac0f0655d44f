"""The drivers of BASELINE configs 4 / 5 on a sharded, trace-free path (jobs.JobSet + ONE packed all-reduce):

* "stats" mode (trace = REDUCED: per-step integer sums instead of histories) gives the same CSV rows as the
  reference-shaped path that pulls every history to the host (experiments.py:593-608, 660-711), early stops included;
* two ranks give exactly what one rank gives -- per-pair minima, per-run best energies / steps-to-best, CSV rows
  (experiments.py:777-846, 1050-1117).

CPU part: gloo, world size 2, the CPU oracle injected as the chain runner.  GPU part (-m gpu): the real path, two ranks
sharing the one GPU of the box (gloo carries the packed reduce there; on a node it is RCCL)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import mcq_amd
from oracle import oracle

abi = mcq_amd.abi
dr = mcq_amd.drivers
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_runner(N, n_steps, init_mode, schedule_params, seeds, mcmc_type="full_3d", early_stop_patience=None, trace=True):
    p = abi.make_params(N, n_steps, init_mode, schedule_params, len(seeds), mcmc_type=mcmc_type,
                        early_stop_patience=early_stop_patience, trace=trace)
    return oracle.run(p, np.asarray(seeds, dtype=np.uint32), trace=trace, states=False), 0.0


PAIRS = dict(N=6, n_steps=700, beta_start_ends=[(0.5, 3.0), (1.0, 3.0), (0.1, 5.0)], annealing_type="sinusoidal_annealing",
             init_mode="random", n_runs=32, base_seed=42, verbose=False, mcmc_type="board", early_stop_patience=None)
CELLS = dict(Ns=[3, 4, 7], n_steps=500, init_modes=["random", "klarner"], n_runs=9, base_seed=42, verbose=False, mcmc_type="board",
             early_stop_patience=None, schedule_params={"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0})


def _csv(path):
    return np.loadtxt(path, delimiter=",", skiprows=1)


def _pairs(tmp, runner, histories, dist=None, **over):
    kw = dict(PAIRS, **over)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        r = dr.run_beta_start_end_pairs(plot=True, out_path_acceptance="a.png", runner=runner, histories=histories, dist=dist, **kw)
    finally:
        os.chdir(cwd)
    rows = {}
    if dist is None or dist.get_rank() == 0:
        for label in r["all_best_energies"]:
            rows[label] = (_csv(os.path.join(tmp, "results", f"{label}.csv")), _csv(os.path.join(tmp, "results", f"acceptance_rates_{label}.csv")))
    return r, rows


def _assert_same_products(a, b, rows_a, rows_b):
    assert a["all_best_energies"] == b["all_best_energies"] and a["min_energies"] == b["min_energies"]
    for label in rows_a:
        ea, aa = rows_a[label]
        eb, ab = rows_b[label]
        np.testing.assert_array_equal(ea[:, :2], eb[:, :2], err_msg=label)          # step, mean: exact
        np.testing.assert_allclose(ea[:, 2], eb[:, 2], rtol=1e-12, err_msg=label)   # std: integer numerator vs two-pass formula
        np.testing.assert_array_equal(np.isnan(aa[:, 1]), np.isnan(ab[:, 1]))
        np.testing.assert_array_equal(np.nan_to_num(aa), np.nan_to_num(ab), err_msg=label)


def test_stats_mode_equals_histories_mode(tmp_path):
    """trace-free statistics == the reference-shaped statistics of the full histories (no early stop: the reference's own
    mean / std need equal lengths, experiments.py:591-595)."""
    (tmp_path / "h").mkdir(), (tmp_path / "s").mkdir()
    h, rows_h = _pairs(str(tmp_path / "h"), oracle_runner, True)
    s, rows_s = _pairs(str(tmp_path / "s"), oracle_runner, False)
    assert all(v is None for v in s["all_histories"].values()) and all(len(v) == 32 for v in h["all_histories"].values())
    _assert_same_products(h, s, rows_h, rows_s)
    assert set(s["energy_stats"]) == set(s["acceptance"]) == set(h["all_best_energies"])


def test_acceptance_bins_with_early_stops():
    """Chains that stop early (experiments.py:349-353) still list their last step in accepted_steps / rejected_steps
    (329-332): the per-step sums carry it, so the binned rates equal the reference's binning of the step lists."""
    sp = {"type": "constant", "beta_const": 5.0}
    res, _ = oracle_runner(6, 900, "random", sp, abi.seeds_for(7, 24), mcmc_type="board", early_stop_patience=150)
    assert (res["hist_len"] < 901).any() and (res["hist_len"] == 901).sum() < 24
    st = mcq_amd.jobs.stats_from_trace(res, 900)
    centers, rates, a, p = mcq_amd.jobs.acceptance_bins_from_steps(st["step_accepted"], st["step_count"], st["step_stopped"], 900, n_bins=100)
    steps = [mcq_amd.experiments.accepted_rejected_steps(res, r) for r in range(24)]
    c2, r2 = dr.acceptance_rates_binned([s[0] for s in steps], [s[1] for s in steps], 900, n_bins=100)
    np.testing.assert_array_equal(centers, c2)
    np.testing.assert_array_equal(np.isnan(rates), np.isnan(r2))
    np.testing.assert_array_equal(np.nan_to_num(rates), np.nan_to_num(r2))
    assert int(p.sum()) == int(res["steps_executed"].sum()) and int(a.sum()) == int(res["n_accepted"].sum())


def _gloo_worker(rank, world, port, tmp, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r, rows = _pairs(os.path.join(tmp, f"rank{rank}"), oracle_runner, False, dist=dist)
        sched = mcq_amd.build_schedule_from_params("linear_annealing", CELLS["n_steps"], beta_start=1.0, beta_end=3.0)
        c = dr.measure_min_energy_vs_N(beta_schedule=sched, plot=False, runner=oracle_runner, dist=dist, **CELLS)
        q.put((rank, r["all_best_energies"], r["min_energies"], {k: [v[0].tolist(), np.nan_to_num(v[1]).tolist()] for k, v in rows.items()},
               {im: [x.tolist() for x in c["results"][im]["all_min_energies"]] + [x.tolist() for x in c["results"][im]["all_steps_to_best"]]
                for im in CELLS["init_modes"]}))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_gloo_ranks_equal_one_process(tmp_path):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    for r in range(2):
        (tmp_path / f"rank{r}").mkdir()
    (tmp_path / "one").mkdir()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=400) for _ in procs), key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one, rows_one = _pairs(str(tmp_path / "one"), oracle_runner, False)
    sched = mcq_amd.build_schedule_from_params("linear_annealing", CELLS["n_steps"], beta_start=1.0, beta_end=3.0)
    c_one = dr.measure_min_energy_vs_N(beta_schedule=sched, plot=False, runner=oracle_runner, **CELLS)
    for g in got:  # both ranks hold the full, identical result
        assert g[1] == one["all_best_energies"] and g[2] == one["min_energies"]
        for im in CELLS["init_modes"]:
            want = [x.tolist() for x in c_one["results"][im]["all_min_energies"]] + [x.tolist() for x in c_one["results"][im]["all_steps_to_best"]]
            assert g[4][im] == want
    assert got[1][3] == {}  # only rank 0 writes the CSVs
    for label, (e, a) in rows_one.items():
        assert got[0][3][label] == [e.tolist(), np.nan_to_num(a).tolist()]


def test_one_packed_collective(monkeypatch):
    """The node-level reduce is ONE all_reduce call (SUM) per job list, whatever the number of jobs."""
    calls = []

    class FakeDist:
        class ReduceOp:
            SUM = "sum"

        @staticmethod
        def is_initialized():
            return True

        @staticmethod
        def get_rank():
            return 0

        @staticmethod
        def get_world_size():
            return 2

        @staticmethod
        def get_backend():
            return "gloo"

        @staticmethod
        def all_reduce(t, op=None):
            calls.append((tuple(t.shape), op))

    jobs = [mcq_amd.jobs.make_job(5, 100, "random", {"type": "constant", "beta_const": 1.0}, 10, 42 + 1000 * i, "board", None) for i in range(4)]
    out = mcq_amd.jobs.JobSet(jobs, want="stats", dist=FakeDist, runner=oracle_runner).run()
    assert len(calls) == 1 and calls[0][1] == "sum"
    assert calls[0][0] == (4 * (6 + 2 + 2 * 10 + 5 * 101),)
    assert all(o["summary"]["n_chains"] == 5 for o in out)  # rank 0 of 2 ran chains [0, 5) of every job; the fake reduce adds nothing


# ---- GPU ------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_stats_mode_equals_histories_mode_on_the_gpu(tmp_path):
    (tmp_path / "h").mkdir(), (tmp_path / "s").mkdir(), (tmp_path / "o").mkdir()
    h, rows_h = _pairs(str(tmp_path / "h"), None, True)
    s, rows_s = _pairs(str(tmp_path / "s"), None, False)
    o, rows_o = _pairs(str(tmp_path / "o"), oracle_runner, False)
    _assert_same_products(h, s, rows_h, rows_s)
    _assert_same_products(o, s, rows_o, rows_s)
    # early stops: trace-free on the GPU vs the oracle's step lists
    s2, rows_s2 = _pairs(str(tmp_path / "s"), None, False, early_stop_patience=120, annealing_type="linear_annealing")
    o2, rows_o2 = _pairs(str(tmp_path / "o"), oracle_runner, False, early_stop_patience=120, annealing_type="linear_annealing")
    _assert_same_products(o2, s2, rows_o2, rows_s2)
    # ragged shard sizes (not a multiple of 16: one launch per pair instead of one batched launch)
    s3, rows_s3 = _pairs(str(tmp_path / "s"), None, False, n_runs=21)
    o3, rows_o3 = _pairs(str(tmp_path / "o"), oracle_runner, False, n_runs=21)
    _assert_same_products(o3, s3, rows_o3, rows_s3)


WORKER = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["MCQ_ROOT"])
import torch, torch.distributed as dist
import mcq_amd
from tests import test_sharded_drivers as t
torch.cuda.set_device(0)
dist.init_process_group("gloo")
tmp = os.path.join(os.environ["MCQ_TMP"], f"rank{dist.get_rank()}")
os.makedirs(tmp, exist_ok=True)
r, rows = t._pairs(tmp, None, False, dist=dist)
sched = mcq_amd.build_schedule_from_params("linear_annealing", t.CELLS["n_steps"], beta_start=1.0, beta_end=3.0)
c = mcq_amd.drivers.measure_min_energy_vs_N(beta_schedule=sched, plot=False, dist=dist, **t.CELLS)
if dist.get_rank() == 0:
    print("RESULT " + json.dumps({"best": r["all_best_energies"], "mins": r["min_energies"],
        "rows": {k: [v[0].tolist(), np.nan_to_num(v[1]).tolist()] for k, v in rows.items()},
        "cells": {im: [x.tolist() for x in c["results"][im]["all_min_energies"]] for im in t.CELLS["init_modes"]}}))
dist.destroy_process_group()
"""


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_equal_one_rank(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MCQ_ROOT=ROOT, MCQ_TMP=str(tmp_path))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    two = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    (tmp_path / "one").mkdir()
    one, rows_one = _pairs(str(tmp_path / "one"), None, False)
    sched = mcq_amd.build_schedule_from_params("linear_annealing", CELLS["n_steps"], beta_start=1.0, beta_end=3.0)
    c_one = dr.measure_min_energy_vs_N(beta_schedule=sched, plot=False, **CELLS)
    assert two["best"] == one["all_best_energies"] and two["mins"] == one["min_energies"]
    for label, (e, a) in rows_one.items():
        assert two["rows"][label] == [e.tolist(), np.nan_to_num(a).tolist()]
    for im in CELLS["init_modes"]:
        assert two["cells"][im] == [x.tolist() for x in c_one["results"][im]["all_min_energies"]]

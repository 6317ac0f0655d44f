"""The "best CPU" variant of the oracle (mcq_oracle_run_fast: O(1) dE from per-line occupancy counters, the north-star's
formulation) against the reference's golden chains and against the naive oracle -- it is bench.py's second CPU baseline
(SURVEY 8d), so it carries its own parity check.  CPU-only."""
import numpy as np

import mcq_amd
from oracle import oracle
from tests import util

abi = mcq_amd.abi


def test_every_golden_chain_fast(golden):
    for case in golden.chains:
        p = util.params_for_case(case)
        res = oracle.run(p, np.array([case["seed"]], dtype=np.uint32), fast=True)
        util.assert_chain_equals_golden(res, 0, case, golden.chain(case), f"fast oracle vs reference {case}")


def test_fast_equals_naive_on_fresh_seeds():
    rng = np.random.default_rng(2025)
    scheds = [{"type": "constant", "beta_const": 0.8}, {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0},
              {"type": "exponential_annealing", "beta_start": 0.5, "beta_end": 4.0}, {"type": "sinusoidal_annealing", "beta_start": 0.1, "beta_end": 5.0},
              {"type": "logarithmic_annealing", "beta_start": 0.5, "beta_end": 3.0}]
    for t in range(40):
        N = int(rng.integers(2, 33))
        mode = ("board", "full_3d")[t % 2]
        init = ("random", "latin", "klarner")[int(rng.integers(0, 3))]
        patience = (None, int(rng.integers(0, 200)))[int(rng.integers(0, 2))]
        n_steps = int(rng.integers(0, 600 if N <= 16 else 150))
        p = abi.make_params(N, n_steps, init, scheds[t % 5], 3, mcmc_type=mode, early_stop_patience=patience, rng=("mt19937", "philox")[t % 3 == 0])
        seeds = rng.integers(0, 2**32, size=3, dtype=np.uint64).astype(np.uint32)
        util.assert_results_equal(oracle.run(p, seeds, fast=True), oracle.run(p, seeds), f"fast vs naive N={N} {mode} {init} patience={patience}")


def test_fast_threads_and_sets():
    sets = [{"type": "sinusoidal_annealing", "beta_start": s, "beta_end": e} for s, e in ((0.1, 2.0), (0.5, 3.0))]
    p = abi.make_params_sets(9, 300, "random", sets, 16, mcmc_type="board")
    seeds = abi.seeds_for(5, 32)
    util.assert_results_equal(oracle.run(p, seeds, fast=True, n_threads=4), oracle.run(p, seeds), "fast, threads, schedule sets")

"""Several beta schedules in ONE launch (mcq_params.n_sets / sets): what run_beta_start_end_pairs does pair by pair
(experiments.py:741-846).  The batched run must equal the concatenation of the per-schedule runs, chain for chain."""
import numpy as np
import pytest

import mcq_amd
from mcq_amd import abi
from oracle import oracle
from tests import util

SETS = [
    {"type": "linear_annealing", "beta_start": 0.1, "beta_end": 2.0},
    {"type": "sinusoidal_annealing", "beta_start": 0.5, "beta_end": 5.0},
    {"type": "constant", "beta_const": 1.5},
    {"type": "exponential_annealing", "beta_start": 1.0, "beta_end": 3.0},
    {"type": "logarithmic_annealing", "beta_start": 2.0, "beta_end": 8.0},
]


def _seeds(n_sets, cps, base=42):
    # pair_seed = base_seed + idx * 1000, chain r of the pair: + r (experiments.py:791, 508)
    return np.concatenate([abi.seeds_for(base + 1000 * t, cps) for t in range(n_sets)])


def test_params_block_and_errors():
    p = abi.make_params_sets(8, 100, "random", SETS, 32, mcmc_type="board")
    assert p.n_sets == 5 and p.chains_per_set == 32 and p.n_chains == 160 and p.sets[1].sched == abi.SCHED["sinusoidal_annealing"]
    assert abi.copy_params(p).sets[4].beta_end == 8.0
    with pytest.raises(ValueError):
        abi.make_params_sets(8, 100, "random", SETS, 24, mcmc_type="board")  # not a multiple of 16
    with pytest.raises(ValueError):
        abi.make_params_sets(8, 100, "random", SETS + [{"type": "nope"}], 32, mcmc_type="board")
    with pytest.raises(ValueError):
        abi.make_params_sets(8, 100, "random", [{"type": "linear_annealing", "beta_start": 1.0}], 32, mcmc_type="board")


def test_oracle_batched_equals_per_schedule():
    cps, n_steps = 16, 300
    p = abi.make_params_sets(7, n_steps, "random", SETS, cps, mcmc_type="board", early_stop_patience=80)
    seeds = _seeds(len(SETS), cps)
    both = oracle.run(p, seeds, n_threads=4)
    for t, sp in enumerate(SETS):
        one = oracle.run(abi.make_params(7, n_steps, "random", sp, cps, mcmc_type="board", early_stop_patience=80), seeds[t * cps:(t + 1) * cps], n_threads=4)
        for k, v in one.items():
            np.testing.assert_array_equal(both[k][t * cps:(t + 1) * cps], v, err_msg=f"set {t}: {k}")


@pytest.mark.gpu
@pytest.mark.parametrize("mode,N,patience,lanes", [("board", 12, None, 0), ("board", 24, None, 0), ("board", 9, 60, 4), ("full_3d", 12, None, 0),
                                                   ("full_3d", 6, None, 16), ("board", 20, None, 8)])
def test_hip_batched_equals_oracle_and_per_schedule(mode, N, patience, lanes):
    cps, n_steps = 32, 700
    p = abi.make_params_sets(N, n_steps, "random", SETS, cps, mcmc_type=mode, early_stop_patience=patience, lanes_per_chain=lanes)
    seeds = _seeds(len(SETS), cps)
    got, _ = mcq_amd._lib.run_host(p, seeds)
    want = oracle.run(p, seeds, n_threads=8)
    util.assert_results_equal(got, want, f"batched {mode} N={N}")
    assert got["near_ties"].sum() == 0
    for t, sp in enumerate(SETS):  # and chain for chain what a launch of that schedule alone returns
        one, _ = mcq_amd._lib.run_host(abi.make_params(N, n_steps, "random", sp, cps, mcmc_type=mode, early_stop_patience=patience,
                                                       lanes_per_chain=lanes), seeds[t * cps:(t + 1) * cps])
        for k in ("accept_bits", "best_energy", "steps_to_best", "final_state", "hist_len"):
            np.testing.assert_array_equal(got[k][t * cps:(t + 1) * cps], one[k], err_msg=f"set {t}: {k}")
        for r in range(cps):  # history rows are defined up to hist_len (the row stride is padded)
            n = one["hist_len"][r]
            np.testing.assert_array_equal(got["energy_hist"][t * cps + r, :n], one["energy_hist"][r, :n], err_msg=f"set {t} chain {r}")


@pytest.mark.gpu
def test_hip_batched_reduced_trace_is_per_set():
    cps, n_steps, N = 48, 500, 12
    p = abi.make_params_sets(N, n_steps, "random", SETS, cps, mcmc_type="board")
    seeds = _seeds(len(SETS), cps)
    red, _ = mcq_amd._lib.run_host(p, seeds, trace="reduced", states=False)
    assert red["step_sum"].shape == (len(SETS), n_steps + 1)
    for t, sp in enumerate(SETS):
        one, _ = mcq_amd._lib.run_host(abi.make_params(N, n_steps, "random", sp, cps, mcmc_type="board"), seeds[t * cps:(t + 1) * cps],
                                       trace="reduced", states=False)
        for k in ("step_sum", "step_sumsq", "step_accepted", "step_count"):
            np.testing.assert_array_equal(red[k][t], one[k], err_msg=f"set {t}: {k}")


@pytest.mark.gpu
def test_hip_rejects_bad_sets():
    p = abi.make_params_sets(8, 10, "random", SETS, 16, mcmc_type="board")
    p.n_chains = 16 * len(SETS) - 16
    with pytest.raises(ValueError):
        mcq_amd._lib.run_host(p, abi.seeds_for(1, p.n_chains))


def test_sets_with_their_own_init_mode_cpu():
    """mcq_schedule.init_plus1: a set may start from its own init mode (the cells of measure_min_energy_vs_N that share N,
    experiments.py:1050-1067).  Oracle: equals the per-init launches chain for chain."""
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    inits = ["random", "latin", "klarner"]
    for mode in ("board", "full_3d"):
        p = abi.make_params_sets(8, 300, "random", [sp] * 3, 16, mcmc_type=mode, init_modes=inits)
        seeds = np.concatenate([abi.seeds_for(42 + 100 * t, 16) for t in range(3)])
        got = oracle.run(p, seeds)
        for t, im in enumerate(inits):
            one = oracle.run(abi.make_params(8, 300, im, sp, 16, mcmc_type=mode), seeds[16 * t:16 * t + 16])
            for k in ("initial_energy", "best_energy", "final_energy", "steps_to_best", "n_accepted", "final_state"):
                np.testing.assert_array_equal(got[k][16 * t:16 * t + 16], one[k], err_msg=f"{mode} {im} {k}")
    with pytest.raises(ValueError):
        abi.make_params_sets(8, 10, "random", [sp] * 2, 16, init_modes=["random", "nope"])


@pytest.mark.gpu
def test_sets_with_their_own_init_mode_gpu():
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    inits = ["klarner", "random", "latin", "random"]
    for mode, N in (("board", 8), ("board", 13), ("full_3d", 6), ("full_3d", 12)):
        p = abi.make_params_sets(N, 400, "latin", [sp] * 4, 32, mcmc_type=mode, init_modes=inits)
        seeds = np.concatenate([abi.seeds_for(7 + 100 * t, 32) for t in range(4)])
        got, _ = mcq_amd._lib.run_host(p, seeds)
        util.assert_results_equal(got, oracle.run(p, seeds, n_threads=8), f"sets with init modes {mode} N={N}")

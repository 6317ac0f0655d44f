"""GPU parity of the board sweep at TWO lanes of a wavefront per chain (32 chains per wavefront; lanes_per_chain = 2): the serial
half of a Metropolis step (experiments.py:308-358: draws, accept test, apply, history) is then issued once per 2 lanes instead of
once per 4.  Same bar as tests/test_hip_parity.py: the reference's golden chains and the CPU oracle, bit for bit, through the C-ABI."""
import itertools

import numpy as np
import pytest

import mcq_amd
from oracle import oracle
from tests import util
from tests.test_hip_parity import CASES, _group_key

abi = mcq_amd.abi
pytestmark = pytest.mark.gpu
BOARD_CASES = [c for c in CASES if c[1] == "board"]


def test_golden_board_chains_at_two_lanes(golden):
    """Every board chain of tests/golden (the reference's own outputs), batched by parameter set."""
    cases = sorted((c for c in golden.chains if c["mode"] == "board"), key=lambda c: str(_group_key(c)))
    n = 0
    for _, grp in itertools.groupby(cases, key=lambda c: str(_group_key(c))):
        grp = list(grp)
        p = util.params_for_case(grp[0], n_chains=len(grp), lanes_per_chain=2)
        res, _ = mcq_amd._lib.run_host(p, np.array([c["seed"] for c in grp], dtype=np.uint32))
        for r, c in enumerate(grp):
            util.assert_chain_equals_golden(res, r, c, golden.chain(c), f"hip G=2 vs reference {c}")
            n += 1
    assert n == len(cases) > 0


@pytest.mark.parametrize("case", BOARD_CASES, ids=lambda c: f"N{c[0]}-{c[2]}-{c[3]['type']}")
def test_two_lanes_equal_the_oracle(case):
    N, mode, init, sp, n_steps, n_chains, patience = case
    if N > 32:
        pytest.skip("32 chains of that size do not fit the LDS")
    p = abi.make_params(N, n_steps, init, sp, n_chains, mcmc_type=mode, early_stop_patience=patience, lanes_per_chain=2)
    seeds = abi.seeds_for(9000 + 17 * N, n_chains)
    want = oracle.run(p, seeds, n_threads=8)
    got, _ = mcq_amd._lib.run_host(p, seeds)
    util.assert_results_equal(got, want, f"hip G=2 vs oracle {case}")
    assert got["near_ties"].sum() == 0


def test_headline_shape_long_run_at_two_lanes():
    """BASELINE configs[1]'s parameters (N = 12 board, linear 1 -> 3) over ~200 MT19937 generations per chain, more chains than one
    wavefront holds, with and without early stopping and with the reduced trace."""
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    seeds = abi.seeds_for(42, 200)
    for patience in (None, 700):
        p = abi.make_params(12, 40000, "random", sp, 200, mcmc_type="board", early_stop_patience=patience, lanes_per_chain=2)
        want = oracle.run(p, seeds, n_threads=16)
        got, _ = mcq_amd._lib.run_host(p, seeds)
        util.assert_results_equal(got, want, f"G=2 long run patience={patience}")
        assert got["near_ties"].sum() == 0
        assert mcq_amd._lib.effective_lanes(p) == 2
    p = abi.make_params(12, 5000, "random", sp, 200, mcmc_type="board", lanes_per_chain=2)
    red, _ = mcq_amd._lib.run_host(p, seeds, trace="reduced", states=False)
    full = oracle.run(p, seeds, n_threads=16)
    st = mcq_amd.jobs.stats_from_trace(full, 5000)
    for k in ("step_sum", "step_sumsq", "step_accepted", "step_count"):
        np.testing.assert_array_equal(red[k], st[k], err_msg=k)


def test_two_lanes_ragged_counts_flags_and_streams():
    """Chain counts around the 32 chains of a wavefront, block edges of the trace, word-by-word draws, the all-float64 accept
    test, the Philox stream, replica exchange and schedule sets: every one equal to 4 lanes per chain (itself held to the oracle)."""
    sp = {"type": "constant", "beta_const": 1.5}
    for n_steps in (0, 1, 15, 16, 31, 63, 64, 65, 129):
        for n_chains in (1, 31, 33):
            p = abi.make_params(6, n_steps, "random", sp, n_chains, mcmc_type="board", lanes_per_chain=2)
            seeds = abi.seeds_for(77, n_chains)
            got, _ = mcq_amd._lib.run_host(p, seeds)
            util.assert_results_equal(got, oracle.run(p, seeds), f"n_steps={n_steps} n_chains={n_chains} G=2")
    sp = {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 3.0}
    seeds = abi.seeds_for(2024, 100)
    for N in (3, 9, 12, 17):
        for kw in (dict(), dict(flags=abi.FLAG_SEQUENTIAL_DRAWS), dict(flags=abi.FLAG_EXACT_EXP), dict(rng="philox"), dict(early_stop_patience=150)):
            a, _ = mcq_amd._lib.run_host(abi.make_params(N, 2500, "random", sp, 100, mcmc_type="board", lanes_per_chain=2, **kw), seeds)
            ref_kw = {k: v for k, v in kw.items() if k != "flags"}
            b, _ = mcq_amd._lib.run_host(abi.make_params(N, 2500, "random", sp, 100, mcmc_type="board", lanes_per_chain=4, **ref_kw), seeds)
            util.assert_results_equal(a, b, f"G=2 vs G=4 N={N} {kw}")
    for R in (4, 16):
        lad = np.linspace(0.7, 1.4, R)
        a, b = (abi.set_exchange(abi.make_params(12, 2000, "random", sp, 64, mcmc_type="board", lanes_per_chain=g), 8, lad) for g in (2, 4))
        ra, _ = mcq_amd._lib.run_host(a, seeds[:64])
        rb, _ = mcq_amd._lib.run_host(b, seeds[:64])
        util.assert_results_equal(ra, rb, f"replica exchange R={R}: G=2 vs G=4")
        np.testing.assert_array_equal(ra["exchange_rung"], rb["exchange_rung"])
        np.testing.assert_array_equal(ra["n_exchanges"], rb["n_exchanges"])
    sets = [{"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, {"type": "constant", "beta_const": 2.0},
            {"type": "sinusoidal_annealing", "beta_start": 0.1, "beta_end": 5.0}]
    for trace in (True, "reduced"):
        pa, pb = (abi.make_params_sets(12, 1500, "random", sets, 32, mcmc_type="board", lanes_per_chain=g, init_modes=["random", "latin", "klarner"],
                                       trace=trace) for g in (2, 4))
        ra, _ = mcq_amd._lib.run_host(pa, seeds[:96], trace=trace, states=trace is True)
        rb, _ = mcq_amd._lib.run_host(pb, seeds[:96], trace=trace, states=trace is True)
        util.assert_results_equal(ra, rb, f"schedule sets trace={trace}: G=2 vs G=4", trace=trace is True)
        if trace == "reduced":
            for k in ("step_sum", "step_sumsq", "step_accepted", "step_count"):
                np.testing.assert_array_equal(ra[k], rb[k], err_msg=k)


def test_two_lanes_limits_are_errors():
    sp = {"type": "constant", "beta_const": 1.0}
    with pytest.raises(ValueError, match="board"):
        mcq_amd._lib.run_host(abi.make_params(6, 10, "random", sp, 4, mcmc_type="full_3d", lanes_per_chain=2), abi.seeds_for(1, 4))
    with pytest.raises(ValueError, match="multiple of 32"):
        mcq_amd._lib.run_host(abi.make_params_sets(6, 10, "random", [sp, sp], 16, mcmc_type="board", lanes_per_chain=2), abi.seeds_for(1, 32))
    with pytest.raises(ValueError, match="does not fit in LDS"):
        mcq_amd._lib.run_host(abi.make_params(80, 10, "random", sp, 4, mcmc_type="board", lanes_per_chain=2), abi.seeds_for(1, 4))

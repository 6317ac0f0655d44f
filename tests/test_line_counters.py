"""GPU parity of the board sweep with dE taken from per-line occupancy counters in LDS (mcq_params.flags & MCQ_FLAG_LINE_COUNTERS: the
formulation BASELINE's north star names; boards up to N = 8 at 4 lanes per chain) -- the reference's golden chains and the CPU oracle
(whose mcq_oracle_run_fast is the same formulation on the CPU, oracle/mcq_oracle.c), bit for bit, through the C-ABI."""
import itertools

import numpy as np
import pytest

import mcq_amd
from oracle import oracle
from tests import util
from tests.test_hip_parity import _group_key

abi = mcq_amd.abi
pytestmark = pytest.mark.gpu
CNT = abi.FLAG_LINE_COUNTERS


def test_golden_small_boards_with_line_counters(golden):
    cases = sorted((c for c in golden.chains if c["mode"] == "board" and c["N"] <= 8), key=lambda c: str(_group_key(c)))
    n = 0
    for _, grp in itertools.groupby(cases, key=lambda c: str(_group_key(c))):
        grp = list(grp)
        p = util.params_for_case(grp[0], n_chains=len(grp), lanes_per_chain=4, flags=CNT)
        res, _ = mcq_amd._lib.run_host(p, np.array([c["seed"] for c in grp], dtype=np.uint32))
        for r, c in enumerate(grp):
            util.assert_chain_equals_golden(res, r, c, golden.chain(c), f"hip line counters vs reference {c}")
            n += 1
    assert n == len(cases) > 20


@pytest.mark.parametrize("N", [2, 3, 4, 5, 6, 7, 8])
def test_line_counters_equal_the_oracle(N):
    for init, sp, patience, n_chains in (("random", {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 3.0}, None, 37),
                                         ("klarner", {"type": "constant", "beta_const": 1.2}, 150, 20),
                                         ("latin", {"type": "sinusoidal_annealing", "beta_start": 0.1, "beta_end": 5.0}, None, 16)):
        p = abi.make_params(N, 4000, init, sp, n_chains, mcmc_type="board", early_stop_patience=patience, lanes_per_chain=4, flags=CNT)
        seeds = abi.seeds_for(100 * N + 7, n_chains)
        want = oracle.run(p, seeds, n_threads=8)
        got, _ = mcq_amd._lib.run_host(p, seeds)
        util.assert_results_equal(got, want, f"line counters N={N} {init} patience={patience}")
        assert got["near_ties"].sum() == 0
        lean, _ = mcq_amd._lib.run_host(p, seeds, trace=False)
        util.assert_results_equal(lean, want, f"line counters, no trace N={N}", trace=False)


def test_line_counters_long_run_sets_and_where_the_flag_does_not_apply():
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    seeds = abi.seeds_for(42, 96)
    p = abi.make_params(8, 50000, "random", sp, 96, mcmc_type="board", lanes_per_chain=4, flags=CNT)
    got, _ = mcq_amd._lib.run_host(p, seeds)
    util.assert_results_equal(got, oracle.run(p, seeds, n_threads=16), "line counters N=8 long run")
    sets = [sp, {"type": "constant", "beta_const": 2.0}, {"type": "exponential_annealing", "beta_start": 1.0, "beta_end": 3.0}]
    a, b = (abi.make_params_sets(6, 1500, "random", sets, 32, mcmc_type="board", lanes_per_chain=4, flags=f, init_modes=["random", "latin", "klarner"]) for f in (CNT, 0))
    ra, _ = mcq_amd._lib.run_host(a, seeds)
    rb, _ = mcq_amd._lib.run_host(b, seeds)
    util.assert_results_equal(ra, rb, "line counters with schedule sets")
    # ignored where it does not apply: larger boards, other lane counts, full_3d, the reduced trace
    for kw in (dict(N=12), dict(N=6, lanes_per_chain=8), dict(N=6, mcmc_type="full_3d")):
        N = kw.pop("N")
        mt = kw.pop("mcmc_type", "board")
        x, _ = mcq_amd._lib.run_host(abi.make_params(N, 800, "random", sp, 40, mcmc_type=mt, flags=CNT, **kw), seeds[:40])
        y, _ = mcq_amd._lib.run_host(abi.make_params(N, 800, "random", sp, 40, mcmc_type=mt, **kw), seeds[:40])
        util.assert_results_equal(x, y, f"flag ignored: N={N} {mt} {kw}")

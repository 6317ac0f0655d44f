"""GPU parity: the HIP kernels, called through the C-ABI (mcq_run_host / mcq_run_device), against
(a) the golden vectors captured from the reference and (b) the CPU oracle on fresh seeded inputs.
Integer outputs must be bit-identical: energy_history, accept bits, best / final energy and state,
steps_to_best, history length (experiments.py:270-279, 367-376)."""
import itertools

import numpy as np
import pytest

import mcq_amd
from oracle import oracle
from tests import util

abi = mcq_amd.abi
pytestmark = pytest.mark.gpu
LANES = (4, 8, 16)


def _group_key(c):
    return (c["mode"], c["init"], c["N"], c["n_steps"], c.get("patience"), tuple(sorted(c["schedule"].items())))


@pytest.mark.parametrize("lanes", LANES)
def test_golden_chains(golden, lanes):
    """Every reference chain of tests/golden, batched by parameter set (seeds differ inside a launch)."""
    cases = sorted(golden.chains, key=lambda c: str(_group_key(c)))
    n = 0
    for _, grp in itertools.groupby(cases, key=lambda c: str(_group_key(c))):
        grp = list(grp)
        p = util.params_for_case(grp[0], n_chains=len(grp), lanes_per_chain=lanes)
        res, _ = mcq_amd._lib.run_host(p, np.array([c["seed"] for c in grp], dtype=np.uint32))
        for r, c in enumerate(grp):
            util.assert_chain_equals_golden(res, r, c, golden.chain(c), f"hip G={lanes} vs reference {c}")
            n += 1
    assert n == len(golden.chains)


@pytest.mark.parametrize("lanes", LANES)
def test_golden_chains_with_other_queen_counts(golden, lanes):
    """metropolis_mcmc(..., Q=...) (experiments.py:199-203): the reference's chains with Q != N^2 queens."""
    for case in golden.chains_q:
        p = util.params_for_case(case, lanes_per_chain=lanes)
        res, _ = mcq_amd._lib.run_host(p, np.array([case["seed"]], dtype=np.uint32))
        util.assert_chain_equals_golden(res, 0, case, golden.chain(case), f"hip G={lanes} vs reference {case}")


def test_golden_boards_beyond_32(golden):
    """Boards up to N = 128 (compare-based probes, E0 family by family): the reference's chains at N = 33..100, every lane width that
    fits the LDS, and N = 101..128 against the oracle."""
    for case in golden.chains_big:
        for lanes in (0, 8, 16) + ((4,) if case["N"] <= 40 else ()):
            p = util.params_for_case(case, lanes_per_chain=lanes)
            res, _ = mcq_amd._lib.run_host(p, np.array([case["seed"]], dtype=np.uint32))
            util.assert_chain_equals_golden(res, 0, case, golden.chain(case), f"hip G={lanes} vs reference {case}")
    sp = {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 3.0}
    for N, init, n, rng, patience in ((128, "random", 9, "mt19937", None), (126, "klarner", 5, "mt19937", 40), (101, "latin", 6, "philox", None), (65, "random", 20, "mt19937", None)):
        p = abi.make_params(N, 600, init, sp, n, mcmc_type="board", rng=rng, early_stop_patience=patience)
        seeds = abi.seeds_for(400 + N, n)
        want = oracle.run(p, seeds, fast=True, n_threads=8)
        got, _ = mcq_amd._lib.run_host(p, seeds)
        util.assert_results_equal(got, want, f"N={N} {init} {rng}")
        assert got["near_ties"].sum() == 0
    with pytest.raises(ValueError, match="does not fit in LDS"):
        mcq_amd._lib.run_host(abi.make_params(128, 10, "random", sp, 4, mcmc_type="board", lanes_per_chain=4), abi.seeds_for(1, 4))


def test_golden_full_3d_beyond_32(golden):
    """full_3d up to N = 64 (64-bit column words, 16 lanes per chain, the queens and the init kernel's N^3 permutation in global memory): the
    reference's chains at N = 33..64 (mcmc.py:6-18 is unbounded), then against the oracle: more chains than the init kernel has permutation
    slices (its rounds come one after the other), the reduced trace, sets with different inits, Q != N^2."""
    for case in golden.chains_wide:
        for lanes in (0, 16):
            p = util.params_for_case(case, lanes_per_chain=lanes)
            res, _ = mcq_amd._lib.run_host(p, np.array([case["seed"]], dtype=np.uint32))
            util.assert_chain_equals_golden(res, 0, case, golden.chain(case), f"hip G={lanes} vs reference {case}")
    sp = {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 3.0}
    for N, init, n, n_steps, Q in ((33, "random", 301, 400, None), (64, "random", 6, 1500, None), (47, "klarner", 9, 1500, None), (64, "latin", 5, 3000, None),
                                   (50, "random", 7, 1500, 700), (36, "random", 10, 1500, 20000)):
        p = abi.make_params(N, n_steps, init, sp, n, mcmc_type="full_3d", Q=Q)
        seeds = abi.seeds_for(500 + N, n)
        want = oracle.run(p, seeds, fast=True, n_threads=8)
        got, _ = mcq_amd._lib.run_host(p, seeds)
        util.assert_results_equal(got, want, f"N={N} {init} Q={Q}")
        assert got["near_ties"].sum() == 0
    # more chains than one round of the init kernel takes (N = 64: the 1 GiB of permutation slices hold 1 024 chains), no trace
    p = abi.make_params(64, 40, "random", sp, 1030, mcmc_type="full_3d", trace=False)
    seeds = abi.seeds_for(7, 1030)
    want = oracle.run(p, seeds, fast=True, n_threads=8, trace=False)
    got, _ = mcq_amd._lib.run_host(p, seeds, trace=False)
    util.assert_results_equal(got, want, "N=64, 1030 chains: two init rounds", trace=False)
    # reduced trace
    p = abi.make_params(40, 2000, "random", sp, 21, mcmc_type="full_3d")
    seeds = abi.seeds_for(9, 21)
    want = oracle.run(p, seeds, fast=True, n_threads=8)
    got, _ = mcq_amd._lib.run_host(p, seeds, trace="reduced")
    util.assert_results_equal(got, want, "N=40 reduced", trace=False)
    st = mcq_amd.jobs.stats_from_trace(want, 2000)
    for k in ("step_sum", "step_sumsq", "step_accepted", "step_count"):
        np.testing.assert_array_equal(got[k], st[k], err_msg=k)
    # three schedule sets, each with its own init (the init kernel runs once per set, on its part of the queen table)
    scheds = [sp, {"type": "constant", "beta_const": 2.0}, {"type": "sinusoidal_annealing", "beta_start": 0.1, "beta_end": 5.0}]
    p = abi.make_params_sets(35, 800, "random", scheds, 16, mcmc_type="full_3d", init_modes=["klarner", "random", "latin"])
    seeds = np.concatenate([abi.seeds_for(70 + 1000 * t, 16) for t in range(3)])
    want = oracle.run(p, seeds, fast=True, n_threads=8)
    got, _ = mcq_amd._lib.run_host(p, seeds)
    util.assert_results_equal(got, want, "N=35 three sets")
    # what stays an explicit error
    for kw, msg in (({"lanes_per_chain": 8}, "16 lanes per chain"), ({"rng": "philox"}, "MT19937 stream only")):
        with pytest.raises(ValueError, match=msg):
            mcq_amd._lib.run_host(abi.make_params(33, 10, "random", sp, 4, mcmc_type="full_3d", **kw), abi.seeds_for(1, 4))


def test_golden_chains_that_continue_a_stream(golden):
    """metropolis_mcmc[_board](..., seed=None) (experiments.py:200-201, 287-288): the reference's chains started from NumPy states at every position
    class, at every lane width; mcq_outputs.stream_words must put the caller's stream where the reference left the global one.  Then against the
    oracle: many chains with positions of their own, sets with different inits, early stops, the reduced trace, replica exchange."""
    for case in golden.chains_stream:
        state, after = golden.stream_state(case)
        wide = case["mode"] == "full_3d" and case["N"] > 32
        for lanes in ((0, 16) if wide else (0, 8, 16) if case["N"] > 32 else (0, 2, 4, 8, 16) if case["mode"] == "board" else (0, 4, 8, 16)):
            p = abi.set_stream_states(util.params_for_case(case, lanes_per_chain=lanes), state[None, :])
            res, _ = mcq_amd._lib.run_host(p, np.array([0], dtype=np.uint32))
            util.assert_chain_equals_golden(res, 0, case, golden.chain(case), f"hip G={lanes} vs reference {case}")
            np.testing.assert_array_equal(util.words_after(state, res["stream_words"][0]), after, err_msg=f"G={lanes}: stream position after {case}")
    sp = {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 3.0}
    rs = np.random.RandomState(4321)

    def states(n):
        out = np.zeros((n, 625), dtype=np.uint32)
        for r in range(n):
            rs.randint(0, 2**32, size=int(rs.randint(1, 2000)), dtype=np.uint32)
            out[r, :624] = rs.get_state()[1]
            out[r, 624] = rs.choice([0, 1, 63, 64, 65, 300, 575, 576, 577, 623, 624, int(rs.randint(0, 625))])
        return out

    for N, mode, init, n, patience, trace in ((12, "board", "random", 150, None, True), (12, "full_3d", "random", 70, None, True), (24, "board", "klarner", 40, 60, True),
                                              (7, "board", "latin", 33, None, "reduced"), (20, "full_3d", "klarner", 9, None, False), (40, "full_3d", "random", 5, None, True),
                                              (3, "board", "random", 64, 25, True), (17, "board", "random", 19, None, True)):
        p = abi.set_stream_states(abi.make_params(N, 900, init, sp, n, mcmc_type=mode, early_stop_patience=patience), states(n))
        seeds = np.zeros(n, dtype=np.uint32)
        want = oracle.run(p, seeds, fast=True, n_threads=8)
        got, _ = mcq_amd._lib.run_host(p, seeds, trace=trace)
        util.assert_results_equal(got, want, f"continued streams: N={N} {mode} {init}", trace=trace is True)
        assert (got["stream_words"] > 0).all()
    scheds = [sp, {"type": "constant", "beta_const": 2.0}, {"type": "sinusoidal_annealing", "beta_start": 0.1, "beta_end": 5.0}]
    p = abi.set_stream_states(abi.make_params_sets(9, 700, "random", scheds, 16, mcmc_type="board", init_modes=["klarner", "random", "latin"]), states(48))
    want = oracle.run(p, np.zeros(48, dtype=np.uint32), n_threads=8)
    got, _ = mcq_amd._lib.run_host(p, np.zeros(48, dtype=np.uint32))
    util.assert_results_equal(got, want, "continued streams: three sets")
    p = abi.set_stream_states(abi.make_params(12, 2000, "random", sp, 32, mcmc_type="board", early_stop_patience=None), states(32))
    abi.set_exchange(p, 10, [1.0, 0.8, 0.6, 0.45])
    want = oracle.run(p, np.zeros(32, dtype=np.uint32), n_threads=8)
    got, _ = mcq_amd._lib.run_host(p, np.zeros(32, dtype=np.uint32))
    util.assert_results_equal(got, want, "continued streams: replica exchange")
    # a seeded run reports its words too, and they equal the oracle's (RESULT_FIELDS): here against NumPy itself for one chain
    p = abi.make_params(6, 200, "random", sp, 1, mcmc_type="board")
    got, _ = mcq_amd._lib.run_host(p, np.array([99], dtype=np.uint32))
    st = np.random.RandomState(99).get_state()
    q = abi.set_stream_states(abi.make_params(6, 200, "random", sp, 1, mcmc_type="board"), st)
    again, _ = mcq_amd._lib.run_host(q, np.array([0], dtype=np.uint32))
    util.assert_results_equal(again, got, "seeded == continued from the seeded state")


def test_seed_none_through_the_python_api(golden):
    """The drop-in call without a seed (the reference's default): metropolis_mcmc[_board] draw from np.random's global stream and leave it where
    the reference leaves it -- the chain's dict equals the reference's and so do the next words of the global stream; two calls in a row continue
    each other (the second equals a chain started from the state the first left)."""
    ex = mcq_amd.experiments
    for case in golden.chains_stream[::3]:
        state, after = golden.stream_state(case)
        gold = golden.chain(case)
        np.random.set_state(("MT19937", state[:624], int(state[624])))
        sched = ex.build_schedule_from_params(case["schedule"]["type"], case["n_steps"], case["schedule"].get("beta_const"),
                                              case["schedule"].get("beta_start"), case["schedule"].get("beta_end"))
        fn = ex.metropolis_mcmc_board if case["mode"] == "board" else ex.metropolis_mcmc
        d = fn(case["N"], case["n_steps"], case["init"], sched, verbose=False, schedule_params=case["schedule"])  # no seed
        np.testing.assert_array_equal(np.random.randint(0, 2**32, size=4, dtype=np.uint32), after, err_msg=f"global stream after {case}")
        np.testing.assert_array_equal(d["energy_history"], gold["hist"], err_msg=str(case))
        assert d["best_energy"] == int(gold["best_energy"]) and d["final_energy"] == int(gold["final_energy"]) and d["steps_to_best"] == int(gold["steps_to_best"])
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    sched = ex.build_schedule_from_params("linear_annealing", 400, None, 1.0, 3.0)
    np.random.seed(2024)
    first = ex.metropolis_mcmc_board(7, 400, "random", sched, verbose=False, schedule_params=sp)
    mid = np.random.get_state()
    second = ex.metropolis_mcmc_board(7, 400, "random", sched, verbose=False, schedule_params=sp)
    p = abi.set_stream_states(abi.make_params(7, 400, "random", sp, 1, mcmc_type="board", early_stop_patience=None), mid)
    want = oracle.run(p, np.zeros(1, dtype=np.uint32))
    np.testing.assert_array_equal(second["energy_history"], want["energy_hist"][0, : int(want["hist_len"][0])])
    seeded = ex.metropolis_mcmc_board(7, 400, "random", sched, verbose=False, seed=2024, schedule_params=sp)
    np.testing.assert_array_equal(first["energy_history"], seeded["energy_history"])  # np.random.seed(s) then seed=None == seed=s


def test_queen_counts_against_the_oracle():
    """Q != N^2 at sizes the golden chains do not reach: many chains per launch, reduced trace, Philox, sets."""
    sp = {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 3.0}
    for N, Q, n in ((12, 150, 70), (12, 1000, 24), (8, 3, 33), (16, 500, 17), (20, 40, 9), (32, 200, 5), (7, 342, 11)):
        for rng in ("mt19937", "philox"):
            p = abi.make_params(N, 1200, "random", sp, n, mcmc_type="full_3d", Q=Q, rng=rng)
            seeds = abi.seeds_for(77 + Q, n)
            want = oracle.run(p, seeds, fast=True, n_threads=8)
            got, _ = mcq_amd._lib.run_host(p, seeds)
            util.assert_results_equal(got, want, f"N={N} Q={Q} {rng}")
            assert got["near_ties"].sum() == 0
    p = abi.make_params(12, 800, "random", sp, 64, mcmc_type="full_3d", Q=100, trace="reduced")
    seeds = abi.seeds_for(5, 64)
    got, _ = mcq_amd._lib.run_host(p, seeds, trace="reduced", states=False)
    full = oracle.run(abi.make_params(12, 800, "random", sp, 64, mcmc_type="full_3d", Q=100), seeds, fast=True, n_threads=8)
    st = mcq_amd.jobs.stats_from_trace(full, 800)
    for k in ("step_sum", "step_sumsq", "step_accepted", "step_count"):
        np.testing.assert_array_equal(got[k], st[k], err_msg=k)


CASES = [
    # (N, mode, init, schedule, n_steps, n_chains, patience)
    (12, "board", "random", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, 3000, 150, None),
    (12, "full_3d", "random", {"type": "exponential_annealing", "beta_start": 1.0, "beta_end": 3.0}, 1500, 70, None),
    (24, "board", "random", {"type": "sinusoidal_annealing", "beta_start": 0.1, "beta_end": 5.0}, 1200, 40, None),
    (7, "board", "klarner", {"type": "logarithmic_annealing", "beta_start": 0.5, "beta_end": 3.0}, 2000, 33, 120),
    (5, "full_3d", "klarner", {"type": "constant", "beta_const": 0.7}, 1000, 21, None),
    (20, "full_3d", "latin", {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 2.0}, 400, 9, None),
    (3, "board", "latin", {"type": "constant", "beta_const": 0.0}, 700, 17, None),
    (32, "board", "random", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 2.0}, 300, 5, None),
    (2, "full_3d", "random", {"type": "constant", "beta_const": 2.0}, 500, 6, None),
    # board sizes on either edge of the unrolled-probe specialisations (ceil(N/4) = 3, 4, 6) and around them
    (9, "board", "random", {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 3.0}, 1500, 19, None),
    (10, "board", "latin", {"type": "constant", "beta_const": 1.0}, 1000, 12, None),
    (13, "board", "klarner", {"type": "constant", "beta_const": 0.3}, 1000, 11, None),
    (16, "board", "random", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, 800, 20, None),
    (21, "board", "random", {"type": "exponential_annealing", "beta_start": 1.0, "beta_end": 3.0}, 600, 9, None),
    (25, "board", "random", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, 400, 7, None),
    (11, "board", "random", {"type": "sinusoidal_annealing", "beta_start": 0.5, "beta_end": 3.0}, 900, 14, None),
    (14, "board", "random", {"type": "logarithmic_annealing", "beta_start": 0.5, "beta_end": 3.0}, 800, 13, None),
    (15, "board", "latin", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, 700, 10, None),
    # full_3d sizes of the 16-bit column-word variants (N = 9..16 at 8 lanes per chain), with and without clamped probe lanes
    (9, "full_3d", "random", {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 3.0}, 700, 13, None),
    (10, "full_3d", "klarner", {"type": "constant", "beta_const": 1.0}, 600, 9, None),
    (13, "full_3d", "latin", {"type": "exponential_annealing", "beta_start": 1.0, "beta_end": 3.0}, 600, 9, None),
    (15, "full_3d", "random", {"type": "sinusoidal_annealing", "beta_start": 0.5, "beta_end": 3.0}, 500, 7, None),
    (16, "full_3d", "random", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, 500, 10, None),
    (17, "full_3d", "random", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, 300, 6, None),
    # every unrolled variant: board ceil(N/4) = 1..6, full_3d ceil(N/4) = 1..4
    (4, "board", "random", {"type": "constant", "beta_const": 1.5}, 600, 9, None),
    (6, "board", "random", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, 800, 11, None),
    (8, "board", "klarner", {"type": "exponential_annealing", "beta_start": 1.0, "beta_end": 3.0}, 800, 10, None),
    (18, "board", "random", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, 500, 8, None),
    (20, "board", "klarner", {"type": "sinusoidal_annealing", "beta_start": 0.5, "beta_end": 3.0}, 500, 6, None),
    (3, "full_3d", "random", {"type": "constant", "beta_const": 1.0}, 600, 9, None),
    (4, "full_3d", "latin", {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 3.0}, 600, 9, None),
    (6, "full_3d", "random", {"type": "logarithmic_annealing", "beta_start": 0.5, "beta_end": 3.0}, 700, 12, None),
    (8, "full_3d", "random", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, 700, 10, None),
    # signs and magnitudes of beta: negative (uphill moves are the certain ones), through zero, and so large that exp underflows
    (6, "board", "random", {"type": "constant", "beta_const": -0.7}, 600, 9, None),
    (6, "full_3d", "random", {"type": "linear_annealing", "beta_start": -1.0, "beta_end": 2.0}, 600, 9, None),
    (8, "board", "random", {"type": "constant", "beta_const": 40.0}, 600, 9, None),
    (7, "full_3d", "random", {"type": "exponential_annealing", "beta_start": 1e-30, "beta_end": 1e-3}, 500, 7, None),
    # largest supported boards: 64 KB permutation array in the init kernel, 5-bit coordinate packing, bit 31 of the masks
    (32, "full_3d", "random", {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 2.0}, 150, 3, None),
    (31, "board", "klarner", {"type": "sinusoidal_annealing", "beta_start": 0.5, "beta_end": 3.0}, 300, 5, 80),
    (32, "full_3d", "klarner", {"type": "constant", "beta_const": 1.0}, 120, 2, None),
]


@pytest.mark.parametrize("lanes", LANES)
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"N{c[0]}-{c[1]}-{c[2]}-{c[3]['type']}")
def test_hip_equals_oracle(case, lanes):
    N, mode, init, sp, n_steps, n_chains, patience = case
    p = abi.make_params(N, n_steps, init, sp, n_chains, mcmc_type=mode, early_stop_patience=patience, lanes_per_chain=lanes)
    seeds = abi.seeds_for(9000 + 17 * N, n_chains)
    want = oracle.run(p, seeds, n_threads=8)
    got, secs = mcq_amd._lib.run_host(p, seeds)
    util.assert_results_equal(got, want, f"hip G={lanes} vs oracle {case}")
    assert got["near_ties"].sum() == 0 and want["near_ties"].sum() == 0
    assert secs > 0


def test_exact_exp_flag_changes_nothing():
    """The float32 bracket around exp(-beta*dE) never decides differently from float64 on every step."""
    sp = {"type": "linear_annealing", "beta_start": 0.2, "beta_end": 3.0}
    for mode in ("board", "full_3d"):
        seeds = abi.seeds_for(31337, 96)
        a, _ = mcq_amd._lib.run_host(abi.make_params(9, 4000, "random", sp, 96, mcmc_type=mode), seeds)
        b, _ = mcq_amd._lib.run_host(abi.make_params(9, 4000, "random", sp, 96, mcmc_type=mode, flags=abi.FLAG_EXACT_EXP), seeds)
        util.assert_results_equal(a, b, f"bracketed vs exact exp ({mode})")


def test_sequential_draw_flag_changes_nothing():
    """The batched proposal (accept-bit selection from the ring) equals word-by-word drawing."""
    sp = {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 3.0}
    seeds = abi.seeds_for(2024, 128)
    for N in (3, 12, 17):
        for lanes in LANES:
            a, _ = mcq_amd._lib.run_host(abi.make_params(N, 3000, "random", sp, 128, mcmc_type="board", lanes_per_chain=lanes), seeds)
            b, _ = mcq_amd._lib.run_host(abi.make_params(N, 3000, "random", sp, 128, mcmc_type="board", lanes_per_chain=lanes,
                                                         flags=abi.FLAG_SEQUENTIAL_DRAWS), seeds)
            util.assert_results_equal(a, b, f"batched vs sequential draws N={N} G={lanes}")


def test_trace_none_matches_trace_i32():
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    p = abi.make_params(12, 2500, "random", sp, 64, mcmc_type="board")
    seeds = abi.seeds_for(5, 64)
    full, _ = mcq_amd._lib.run_host(p, seeds, trace=True)
    lean, _ = mcq_amd._lib.run_host(p, seeds, trace=False)
    util.assert_results_equal(full, lean, "trace i32 vs none", trace=False)


def test_default_lane_choice_never_changes_a_result():
    """lanes_per_chain = 0 lets the library choose (4 for boards up to N = 12, 8 beyond and for full_3d; 8 or 16 for a board
    launch that leaves the device less than half full): whatever it chooses equals every explicit choice."""
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    import ctypes

    L = mcq_amd._lib.lib()
    simds = L.mcq_device_simds()
    for N, n, want in ((12, 48, 16), (12, 4 * simds, 16), (12, 8 * simds, 8), (12, 16 * simds, 4), (24, 48, 8), (24, 16 * simds, 8), (17, 48, 16), (8, 48, 4), (3, 48, 4)):
        p = abi.make_params(N, 10, "random", sp, n, mcmc_type="board")
        assert L.mcq_effective_lanes(ctypes.byref(p)) == want, (N, n)
    for N, mode, n_chains in ((12, "board", 48), (12, "board", 20000), (8, "board", 100), (12, "full_3d", 40), (24, "board", 48), (24, "board", 9000)):
        n_steps = 400 if n_chains > 1000 else 1500
        seeds = abi.seeds_for(31 + N, n_chains)
        ref, _ = mcq_amd._lib.run_host(abi.make_params(N, n_steps, "random", sp, n_chains, mcmc_type=mode, early_stop_patience=200), seeds)
        for lanes in LANES:
            got, _ = mcq_amd._lib.run_host(abi.make_params(N, n_steps, "random", sp, n_chains, mcmc_type=mode, early_stop_patience=200,
                                                           lanes_per_chain=lanes), seeds)
            util.assert_results_equal(ref, got, f"default lanes vs G={lanes}: N={N} {mode} n_chains={n_chains}")


def test_ragged_chain_counts_and_tail_blocks():
    """n_chains not a multiple of the chains per wavefront, n_steps around the 16/32/64-entry block edges."""
    sp = {"type": "constant", "beta_const": 1.5}
    for n_steps in (0, 1, 15, 16, 31, 63, 64, 65, 127, 128, 129):
        for n_chains in (1, 3, 5):
            p = abi.make_params(6, n_steps, "random", sp, n_chains, mcmc_type="board")
            seeds = abi.seeds_for(77, n_chains)
            want = oracle.run(p, seeds)
            for lanes in LANES:
                p.lanes_per_chain = lanes
                got, _ = mcq_amd._lib.run_host(p, seeds)
                util.assert_results_equal(got, want, f"n_steps={n_steps} n_chains={n_chains} G={lanes}")


def test_device_pointer_entry_point():
    """mcq_run_device with torch-allocated device buffers on a non-default stream."""
    import torch

    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    p = abi.make_params(12, 2000, "random", sp, 100, mcmc_type="board")
    seeds = abi.seeds_for(42, 100)
    run = mcq_amd._lib.DeviceRun(p, seeds)
    st = torch.cuda.Stream()
    run.launch(stream=st)
    st.synchronize()
    util.assert_results_equal(run.results(), oracle.run(p, seeds, n_threads=8), "mcq_run_device vs oracle")


def test_run_experiment_plumbing(golden):
    """BASELINE config 1 through the drop-in run_experiment: the reference's 6-tuple (experiments.py:573)."""
    pl = golden.manifest["plumbing"]
    sched = mcq_amd.build_schedule_from_params("constant", pl["n_steps"], beta_const=5.0)
    hist, best, times, acc, rej, stb = mcq_amd.run_experiment(
        N=pl["N"], n_steps=pl["n_steps"], init_mode=pl["init"], beta_schedule=sched, n_runs=pl["n_runs"],
        base_seed=pl["base_seed"], schedule_params=pl["schedule"], mcmc_type="board", early_stop_patience=None)
    assert [int(h[0]) for h in hist] == pl["E0"] and best == pl["best"] and stb == pl["steps_to_best"]
    assert [len(a) for a in acc] == pl["n_accepted"]
    assert all(len(a) + len(r) == pl["n_steps"] for a, r in zip(acc, rej)) and len(times) == pl["n_runs"]
    assert all(len(h) == pl["n_steps"] + 1 for h in hist)


def test_errors_cross_the_abi_as_exceptions():
    sp = {"type": "constant", "beta_const": 1.0}
    p = abi.make_params(6, 10, "random", sp, 2, mcmc_type="board")
    p.sched = 9
    with pytest.raises(ValueError):
        mcq_amd._lib.run_host(p, abi.seeds_for(0, 2))


def test_on_device_trace_statistics():
    """mcq_trace_stats_device: per-step sums / counts and binned acceptance from the resident trace equal what
    NumPy computes from the host copy (the reference's plot helpers, experiments.py:593-595, 660-695)."""
    dr = mcq_amd.drivers
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    for patience, n_steps in ((None, 3333), (60, 1500)):
        p = abi.make_params(8, n_steps, "random", sp, 77, mcmc_type="board", early_stop_patience=patience)
        run = mcq_amd._lib.DeviceRun(p, abi.seeds_for(11, 77))
        run.launch()
        st = run.trace_stats(n_bins=100)
        res = run.results()
        L = res["hist_len"]
        for e in (0, 1, 63, 64, n_steps // 2, n_steps):
            col = np.array([res["energy_hist"][r, e] for r in range(77) if e < L[r]], dtype=np.int64)
            assert st["step_count"][e] == len(col) and st["step_sum"][e] == col.sum() and st["step_sumsq"][e] == (col * col).sum()
        steps = [mcq_amd.experiments.accepted_rejected_steps(res, r) for r in range(77)]
        centers, rates = dr.acceptance_rates_binned([s[0] for s in steps], [s[1] for s in steps], n_steps, n_bins=100)
        got = dr.acceptance_rates_from_bins(st["bin_accepted"], st["bin_proposed"])
        np.testing.assert_array_equal(np.isnan(got), np.isnan(rates))
        np.testing.assert_allclose(got[~np.isnan(rates)], rates[~np.isnan(rates)], rtol=0, atol=0)
        np.testing.assert_array_equal(st["bin_centers"], centers)
        if patience is None:
            hist = [res["energy_hist"][r, : L[r]] for r in range(77)]
            mean, std = dr.energy_statistics(hist)
            m2, s2 = dr.mean_std_from_sums(st["step_sum"], st["step_sumsq"], st["step_count"])
            np.testing.assert_array_equal(m2, mean)
            np.testing.assert_allclose(s2, std, rtol=1e-12)


def test_reduced_trace_equals_statistics_of_the_full_trace():
    """trace == REDUCED (per-entry sums accumulated inside the sweep) against mcq_trace_stats_device on a full-trace
    run of the same chains, with and without early stops, for every lane width, ragged chain counts included."""
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    for mode, N, patience, n_steps, n_chains in (("board", 8, None, 1000, 77), ("board", 8, 60, 1500, 77), ("board", 12, None, 333, 16),
                                                   ("full_3d", 6, None, 700, 41), ("board", 4, None, 300, 9), ("board", 16, None, 300, 9),
                                                   ("board", 18, None, 200, 7), ("board", 22, None, 200, 6), ("full_3d", 12, None, 300, 10)):
        for lanes in LANES:
            p = abi.make_params(N, n_steps, "random", sp, n_chains, mcmc_type=mode, early_stop_patience=patience, lanes_per_chain=lanes)
            seeds = abi.seeds_for(11, n_chains)
            full = mcq_amd._lib.DeviceRun(p, seeds)
            full.launch()
            st = full.trace_stats(n_bins=10)
            res = full.results()
            red, _ = mcq_amd._lib.run_host(p, seeds, trace="reduced", states=False)
            what = f"{mode} N={N} patience={patience} G={lanes}"
            np.testing.assert_array_equal(red["step_sum"], st["step_sum"], err_msg=what)
            np.testing.assert_array_equal(red["step_sumsq"], st["step_sumsq"], err_msg=what)
            np.testing.assert_array_equal(red["step_count"], st["step_count"], err_msg=what)
            acc = np.zeros(n_steps + 1, dtype=np.int64)  # accepted[e] = chains whose step e - 1 was accepted
            for r in range(n_chains):
                a_steps, _ = mcq_amd.experiments.accepted_rejected_steps(res, r)
                acc[a_steps + 1] += 1  # the step a chain stopped at is executed (and listed) although it appends no entry
            np.testing.assert_array_equal(red["step_accepted"], acc, err_msg=what)
            for k in ("best_energy", "final_energy", "steps_to_best", "n_accepted", "hist_len"):
                np.testing.assert_array_equal(red[k], res[k], err_msg=f"{what}: {k}")


def test_long_runs_full_trace():
    """Tens of thousands of steps: ~300 MT19937 generations per chain, thousands of ring wrap-arounds and history
    blocks, the whole annealing range of beta -- full traces against the oracle, both modes."""
    for mode, sp, n_steps, n_chains in (("board", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, 60000, 192),
                                        ("full_3d", {"type": "exponential_annealing", "beta_start": 1.0, "beta_end": 3.0}, 25000, 96)):
        p = abi.make_params(12, n_steps, "random", sp, n_chains, mcmc_type=mode)
        seeds = abi.seeds_for(4242, n_chains)
        want = oracle.run(p, seeds, n_threads=16)
        got, _ = mcq_amd._lib.run_host(p, seeds)
        util.assert_results_equal(got, want, f"long run {mode}")
        assert got["near_ties"].sum() == 0


def test_device_beta_tables_against_the_reference(golden):
    """beta(step).  The sweep normally gets the reference's own values (abi.beta_values -> mcq_params.beta_table; bit-equal to the
    reference's closures, tests/test_oracle_golden.py).  Without a table the device evaluates the schedule itself (mcq_beta_kernel,
    observable through mcq_beta_table_device): exact for constant / linear, and within 2^-51 * max|beta| of the reference for the
    schedules that go through exp / log / cos (OCML instead of NumPy's loops; NumPy and glibc differ in the last place too)."""
    z = golden.npz("beta")
    for c in golden.manifest["beta"]:
        p = abi.make_params(6, c["n_steps"], "random", c["schedule"], 16, mcmc_type="board")
        tab, c32 = mcq_amd._lib.beta_table_device(p)
        steps = z[c["key"] + "_steps"]
        got, want = tab[steps], z[c["key"] + "_beta"]
        if c["schedule"]["type"] in ("constant", "linear_annealing"):
            np.testing.assert_array_equal(got, want, err_msg=str(c))
        else:  # one ulp of exp / log / cos, seen from beta (1 - cos x cancels near x = 0: several ulps of a small beta)
            bound = 2.0 ** -51 * max(abs(c["schedule"]["beta_start"]), abs(c["schedule"]["beta_end"]))
            assert float(np.abs(got - want).max()) <= bound, (c, float(np.abs(got - want).max()))
        np.testing.assert_array_equal(c32[steps], (-(got * 1.4426950408889634)).astype(np.float32), err_msg=str(c))
    # schedule sets: one table per set, equal to the single-schedule tables
    sets = [{"type": "sinusoidal_annealing", "beta_start": s, "beta_end": e} for s, e in ((0.1, 2.0), (0.5, 3.0), (2.0, 8.0))]
    ps = abi.make_params_sets(6, 777, "random", sets, 16, mcmc_type="board")
    tabs, _ = mcq_amd._lib.beta_table_device(ps)
    for t, sp in enumerate(sets):
        one, _ = mcq_amd._lib.beta_table_device(abi.make_params(6, 777, "random", sp, 16, mcmc_type="board"))
        np.testing.assert_array_equal(tabs[t], one)


def test_given_beta_table_is_what_the_sweep_uses():
    """mcq_params.beta_table: (a) the default Python path passes the reference's values and (b) a hand-filled struct without
    them (device-evaluated schedule) gives the same chains -- the two tables differ by at most an ulp and no uniform falls that
    close to an acceptance probability (near_ties == 0); (c) an arbitrary table is followed: a table of zeros accepts every move."""
    sp = {"type": "exponential_annealing", "beta_start": 0.5, "beta_end": 4.0}
    seeds = abi.seeds_for(808, 48)
    for mode in ("board", "full_3d"):
        p = abi.make_params(10, 3000, "random", sp, 48, mcmc_type=mode)
        a, _ = mcq_amd._lib.run_host(p, seeds)
        q = abi.copy_params(p)
        q._schedules = None  # no Python-side schedule: the library evaluates beta on the device
        b, _ = mcq_amd._lib.run_host(q, seeds)
        util.assert_results_equal(a, b, f"host beta table vs device-evaluated schedule ({mode})")
        assert a["near_ties"].sum() == 0
        util.assert_results_equal(a, oracle.run(p, seeds, n_threads=8), f"vs oracle ({mode})")
        z = abi.copy_params(p)
        zeros = np.zeros((1, 3000))
        z._schedules, z.beta_table = None, zeros.ctypes.data
        c, _ = mcq_amd._lib.run_host(z, seeds)
        assert (c["n_accepted"] == 3000).all()
    # schedule sets through the device-pointer entry point
    sets = [{"type": "logarithmic_annealing", "beta_start": 0.5, "beta_end": 3.0}, {"type": "sinusoidal_annealing", "beta_start": 0.1, "beta_end": 5.0}]
    ps = abi.make_params_sets(9, 2000, "random", sets, 32, mcmc_type="board")
    sd = abi.seeds_for(5, 64)
    run = mcq_amd._lib.DeviceRun(ps, sd)
    assert run.p.beta_table
    run.launch()
    util.assert_results_equal(run.results(), oracle.run(ps, sd, n_threads=8), "schedule sets with host beta tables")


def test_buffer_contract_is_checked():
    """include/mcq.h: hist_stride a multiple of 16, energy_hist and the workspace 64-byte aligned -- violations are MCQ_EINVAL."""
    import ctypes as C

    import torch

    sp = {"type": "constant", "beta_const": 1.0}
    p = abi.make_params(6, 100, "random", sp, 4, mcmc_type="board")
    run = mcq_amd._lib.DeviceRun(p, abi.seeds_for(1, 4))
    L = mcq_amd._lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    bad = abi.copy_params(run.p)
    bad.hist_stride = 101
    assert L.mcq_run_device(C.byref(bad), run.seeds.data_ptr(), C.byref(run.out), run.ws.data_ptr(), run.ws_bytes, st) == abi.EINVAL
    assert b"multiple of 16" in L.mcq_last_error()
    assert L.mcq_run_device(C.byref(run.p), run.seeds.data_ptr(), C.byref(run.out), run.ws.data_ptr() + 4, run.ws_bytes, st) == abi.EINVAL  # (rejected before anything is touched)
    out = abi.Outputs.from_buffer_copy(run.out)
    out.energy_hist = run.t["energy_hist"].data_ptr() + 4
    assert L.mcq_run_device(C.byref(run.p), run.seeds.data_ptr(), C.byref(out), run.ws.data_ptr(), run.ws_bytes, st) == abi.EINVAL
    run.launch()  # and the well-formed call still works
    torch.cuda.synchronize()
    util.assert_results_equal(run.results(), oracle.run(p, abi.seeds_for(1, 4)), "after rejected calls")

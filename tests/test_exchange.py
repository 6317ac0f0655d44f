"""Replica exchange (mcq_params.exchange_every > 0, include/mcq.h): NOT a mode of the reference (its report, section VI, names
better moves as future work), never a default.  Parity here means: the oracle's exchange mode equals an independent NumPy
restatement that draws from numpy.random.RandomState itself, reduces to the pinned plain chains when nothing can be exchanged,
and the HIP kernels equal the oracle bit for bit in this mode (-m gpu)."""
import numpy as np
import pytest

import mcq_amd
from oracle import oracle
from tests import util

abi = mcq_amd.abi
SP = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}


def _ladder(R, lo=0.7, hi=1.4):
    return lo * (hi / lo) ** (np.arange(R) / (R - 1))


def _params(N, n_steps, n, mode, every, ladder, init="random", sp=SP, **kw):
    p = abi.make_params(N, n_steps, init, sp, n, mcmc_type=mode, early_stop_patience=None, **kw)
    return abi.set_exchange(p, every, ladder)


# ---- an independent restatement: the board chain on numpy's own RandomState, ladders in lockstep -----------------------------
def _attacks(i, j, k, i2, j2, k2):
    di, dj, dk = abs(i2 - i), abs(j2 - j), abs(k2 - k)
    return ((i2 == i and k2 == k) or (j2 == j and k2 == k) or (k2 == k and di == dj) or (j2 == j and di == dk)
            or (i2 == i and dj == dk) or (di == dj == dk))


def _conflicts(h, N, i, j, k):
    return sum(_attacks(i, j, k, a, b, int(h[a, b])) for a in range(N) for b in range(N) if (a, b) != (i, j))


def _numpy_model(N, n_steps, seeds, every, ladder, beta):
    R = len(ladder)
    out = []
    for g in range(len(seeds) // R):
        rs = [np.random.RandomState(int(s)) for s in seeds[g * R:(g + 1) * R]]
        hs = [r.randint(0, N, size=(N, N)) for r in rs]
        E = [sum(_attacks(a // N, a % N, int(h[a // N, a % N]), b // N, b % N, int(h[b // N, b % N]))
                 for a in range(N * N) for b in range(a + 1, N * N)) for h in hs]
        hist = [[e] for e in E]
        rung = list(range(R))
        nex = [0] * R
        for s in range(n_steps):
            for c in range(R):
                r, h = rs[c], hs[c]
                i, j = r.randint(0, N), r.randint(0, N)
                old = int(h[i, j])
                new = r.randint(0, N)
                while new == old:
                    new = r.randint(0, N)
                dE = _conflicts(h, N, i, j, new) - _conflicts(h, N, i, j, old)
                b = beta[s] * ladder[rung[c]]
                if r.random_sample() < min(1.0, np.exp(-b * dE)):
                    h[i, j] = new
                    E[c] += dE
                hist[c].append(E[c])
            if (s + 1) % every == 0:
                n = (s + 1) // every
                for t in range(n & 1, R - 1, 2):
                    a, b2 = rung.index(t), rung.index(t + 1)
                    x = (beta[s] * ladder[t] - beta[s] * ladder[t + 1]) * float(E[a] - E[b2])
                    if rs[a].random_sample() < min(1.0, np.exp(x)):
                        rung[a], rung[b2] = t + 1, t
                        nex[a] += 1
                        nex[b2] += 1
        for c in range(R):
            out.append((hist[c], rung[c], nex[c], hs[c].reshape(-1)))
    return out


@pytest.mark.parametrize("every,R", [(1, 2), (3, 4), (7, 4)])
def test_oracle_exchange_equals_a_numpy_restatement(every, R):
    N, n_steps, n = 4, 150, 2 * R
    lad = _ladder(R)
    p = _params(N, n_steps, n, "board", every, lad)
    seeds = abi.seeds_for(1234, n)
    res = oracle.run(p, seeds)
    want = _numpy_model(N, n_steps, seeds, every, lad, abi.beta_values(SP, n_steps))
    assert int(res["near_ties"].sum()) == 0
    assert int(res["n_exchanges"].sum()) > 0
    for r, (hist, rung, nex, h) in enumerate(want):
        np.testing.assert_array_equal(res["energy_hist"][r, : n_steps + 1], hist, err_msg=f"chain {r}")
        assert int(res["exchange_rung"][r]) == rung and int(res["n_exchanges"][r]) == nex
        np.testing.assert_array_equal(res["final_state"][r], h)


@pytest.mark.parametrize("mode", ["board", "full_3d"])
@pytest.mark.parametrize("fast", [False, True])
def test_nothing_to_exchange_is_the_plain_chain(mode, fast):
    """A ladder of ones and a period beyond the run: the step function of the exchange mode against the pinned plain chains."""
    n, n_steps = 8, 300
    seeds = abi.seeds_for(42, n)
    for N, init in ((6, "random"), (7, "klarner"), (12, "latin")):
        plain = oracle.run(abi.make_params(N, n_steps, init, SP, n, mcmc_type=mode, early_stop_patience=None), seeds, fast=fast)
        ex = oracle.run(_params(N, n_steps, n, mode, n_steps + 1, np.ones(4), init=init), seeds, fast=fast)
        util.assert_results_equal(ex, plain, f"{mode} N={N}")
        assert not ex["n_exchanges"].any() and list(ex["exchange_rung"]) == [0, 1, 2, 3, 0, 1, 2, 3]


@pytest.mark.parametrize("mode", ["board", "full_3d"])
def test_exchange_properties_and_the_line_counter_variant(mode):
    n, n_steps, R = 32, 400, 8
    seeds = abi.seeds_for(7, n)
    p = _params(8, n_steps, n, mode, 5, _ladder(R))
    a = oracle.run(p, seeds)
    b = oracle.run(p, seeds, fast=True, n_threads=4)
    util.assert_results_equal(a, b, mode)
    for k in ("exchange_rung", "n_exchanges"):
        np.testing.assert_array_equal(a[k], b[k])
    rungs = a["exchange_rung"].reshape(-1, R)
    assert (np.sort(rungs, axis=1) == np.arange(R)).all()          # every ladder ends as a permutation of its rungs
    assert (a["n_exchanges"].reshape(-1, R).sum(axis=1) % 2 == 0).all() and a["n_exchanges"].sum() > 0
    assert int(a["near_ties"].sum()) == 0
    plain = oracle.run(abi.make_params(8, n_steps, "random", SP, n, mcmc_type=mode, early_stop_patience=None), seeds)
    np.testing.assert_array_equal(a["initial_energy"], plain["initial_energy"])
    assert not np.array_equal(a["energy_hist"], plain["energy_hist"])


def test_exchange_with_schedule_sets():
    """Two schedule sets of two ladders each: every ladder follows its set's beta."""
    sets = [SP, {"type": "constant", "beta_const": 2.0}]
    n_steps, R = 200, 4
    p = abi.make_params_sets(6, n_steps, "random", sets, 16, mcmc_type="board", early_stop_patience=None)
    abi.set_exchange(p, 4, _ladder(R))
    seeds = abi.seeds_for(99, 32)
    res = oracle.run(p, seeds)
    for t, sp in enumerate(sets):
        one = oracle.run(_params(6, n_steps, 16, "board", 4, _ladder(R), sp=sp), seeds[16 * t:16 * t + 16])
        np.testing.assert_array_equal(res["energy_hist"][16 * t:16 * t + 16], one["energy_hist"])
        np.testing.assert_array_equal(res["exchange_rung"][16 * t:16 * t + 16], one["exchange_rung"])


def test_exchange_parameter_errors():
    p = abi.make_params(6, 10, "random", SP, 8, mcmc_type="board", early_stop_patience=None)
    with pytest.raises(ValueError):
        abi.set_exchange(p, 0, np.ones(4))
    with pytest.raises(ValueError):
        abi.set_exchange(p, 2, np.ones(3))
    with pytest.raises(ValueError):
        abi.set_exchange(abi.make_params(6, 10, "random", SP, 6, mcmc_type="board"), 2, np.ones(4))
    q = abi.make_params(6, 10, "random", SP, 8, mcmc_type="board", early_stop_patience=5)
    abi.set_exchange(q, 2, np.ones(4))
    with pytest.raises(ValueError, match="early stopping off"):
        oracle.run(q, abi.seeds_for(1, 8))
    import ctypes

    L = mcq_amd._lib.lib()
    assert L.mcq_workspace_bytes(ctypes.byref(q)) == 0 and b"early stopping off" in L.mcq_last_error()
    r = abi.make_params(6, 10, "random", SP, 8, mcmc_type="board", early_stop_patience=None, trace="reduced")
    abi.set_exchange(r, 2, np.ones(4))
    assert L.mcq_workspace_bytes(ctypes.byref(r)) == 0 and b"trace none or i32" in L.mcq_last_error()
    # a rung runs at beta(step) * ladder[t]: multipliers that are not positive finite numbers are refused by both libraries
    for bad in (0.0, -1.0, float("nan"), float("inf")):
        b = abi.set_exchange(abi.make_params(6, 10, "random", SP, 8, mcmc_type="board", early_stop_patience=None), 2, [1.0, bad, 1.2, 1.4])
        with pytest.raises(ValueError, match="finite and positive"):
            oracle.run(b, abi.seeds_for(1, 8))
        assert L.mcq_workspace_bytes(ctypes.byref(b)) == 0 and b"finite and positive" in L.mcq_last_error()


# ---- HIP == oracle ---------------------------------------------------------------------------------------------------------
EX_FIELDS = ("exchange_rung", "n_exchanges")


def _gpu_vs_oracle(p, seeds, what, trace=True):
    got, _ = mcq_amd._lib.run_host(p, seeds, trace=trace)
    want = oracle.run(p, seeds, trace=trace, fast=True, n_threads=8)
    util.assert_results_equal(got, want, what, trace=trace)
    for k in EX_FIELDS:
        np.testing.assert_array_equal(got[k], want[k], err_msg=f"{what}: {k}")
    assert int(got["near_ties"].sum()) == 0 and int(want["near_ties"].sum()) == 0, what
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["board", "full_3d"])
@pytest.mark.parametrize("every", [1, 64, 1000])
def test_hip_exchange_equals_oracle(mode, every):
    for N, R, lanes in ((12, 16, 0), (12, 4, 0), (6, 2, 4), (17, 8, 8), (9, 4, 16)):
        if lanes and 64 // lanes < R:
            continue
        n = 96 if R != 16 else 128
        p = _params(N, 2500, n, mode, every, _ladder(R), lanes_per_chain=lanes)
        got = _gpu_vs_oracle(p, abi.seeds_for(1000 + N, n), f"{mode} N={N} R={R} K={every} lanes={lanes}")
        if every < 2500:
            assert got["n_exchanges"].sum() > 0


@pytest.mark.gpu
def test_hip_exchange_sets_philox_and_no_trace():
    sets = [SP, {"type": "sinusoidal_annealing", "beta_start": 0.5, "beta_end": 5.0}]
    p = abi.make_params_sets(12, 3000, "random", sets, 64, mcmc_type="board", early_stop_patience=None, init_modes=["random", "latin"])
    abi.set_exchange(p, 50, _ladder(16))
    _gpu_vs_oracle(p, abi.seeds_for(5, 128), "sets")
    q = _params(12, 3000, 64, "board", 10, _ladder(8), rng="philox")
    _gpu_vs_oracle(q, abi.seeds_for(6, 64), "philox")
    r = _params(12, 3000, 64, "full_3d", 10, _ladder(8), trace=False)
    _gpu_vs_oracle(r, abi.seeds_for(7, 64), "no trace", trace=False)


@pytest.mark.gpu
def test_hip_exchange_lane_count_changes_nothing():
    p4 = _params(10, 2000, 64, "board", 8, _ladder(4), lanes_per_chain=4)
    p16 = _params(10, 2000, 64, "board", 8, _ladder(4), lanes_per_chain=16)
    a, _ = mcq_amd._lib.run_host(p4, abi.seeds_for(3, 64))
    b, _ = mcq_amd._lib.run_host(p16, abi.seeds_for(3, 64))
    util.assert_results_equal(a, b, "lanes 4 vs 16")
    for k in EX_FIELDS:
        np.testing.assert_array_equal(a[k], b[k])


@pytest.mark.gpu
def test_hip_exchange_ladder_too_wide_for_the_lane_count():
    p = _params(12, 100, 64, "board", 8, _ladder(16), lanes_per_chain=8)
    with pytest.raises(ValueError, match="lanes_per_chain"):
        mcq_amd._lib.run_host(p, abi.seeds_for(3, 64))

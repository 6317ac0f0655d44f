"""Randomised differential test: parameter sets drawn from a fixed seed, HIP against the oracle through the C-ABI.
Small runs (the oracle stays under a second each) over the whole parameter space: every N, both modes, the three
initial states, the five schedules with ordinary and odd beta values, early stopping, all lane widths, the three trace
modes' shared outputs, ragged chain counts and lengths around the flush boundaries (16 / 32 / 64 steps)."""
import os

import numpy as np
import pytest

import mcq_amd
from mcq_amd import abi
from oracle import oracle
from tests import util

# Soak runs on demand: MCQ_FUZZ_CASES / MCQ_FUZZ_SEED / MCQ_FUZZ_LONG=1 (longer chains) widen the same generator;
# the defaults are what the suite runs.
N_CASES = int(os.environ.get("MCQ_FUZZ_CASES", "400"))
FUZZ_SEED = int(os.environ.get("MCQ_FUZZ_SEED", "20251004"))
FUZZ_LONG = os.environ.get("MCQ_FUZZ_LONG", "") == "1"


def _cases():
    rng = np.random.default_rng(FUZZ_SEED)
    scheds = ["constant", "linear_annealing", "exponential_annealing", "logarithmic_annealing", "sinusoidal_annealing"]
    out = []
    for c in range(N_CASES):
        mode = "board" if rng.random() < 0.55 else "full_3d"
        N = int(rng.choice([2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 23, 24, 25, 28, 32]))
        if mode == "full_3d" and N > 20:
            N = int(rng.integers(2, 21))  # the init permutation of N^3 cells keeps the oracle slow beyond that
        init = str(rng.choice(["random", "latin", "klarner"]))
        st = str(rng.choice(scheds))
        if st == "constant":
            sp = {"type": st, "beta_const": float(rng.choice([0.0, 0.3, 1.0, 2.5, 8.0, -0.5]))}
        else:
            lo = float(rng.choice([0.05, 0.5, 1.0, 2.0]))
            hi = float(rng.choice([0.5, 2.0, 3.0, 6.0]))
            if st == "exponential_annealing" and rng.random() < 0.2:
                lo, hi = hi, lo  # cooling and heating
            sp = {"type": st, "beta_start": lo, "beta_end": hi}
        n_steps = int(rng.choice([0, 1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 257, 400]))
        if FUZZ_LONG and rng.random() < 0.3:
            n_steps = int(rng.choice([1500, 3000, 5000]))
        n_chains = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 24]))
        patience = None
        if mode == "board" and rng.random() < 0.35:
            patience = int(rng.choice([0, 1, 2, 10, 40, 1000]))
        lanes = int(rng.choice([0, 4, 8, 16]))
        seed0 = int(rng.integers(0, 2**32 - 1 - n_chains))
        out.append((c, N, mode, init, sp, n_steps, n_chains, patience, lanes, seed0))
    return out


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_random_parameter_sets():
    for c, N, mode, init, sp, n_steps, n_chains, patience, lanes, seed0 in _cases():
        what = f"case {c}: N={N} {mode} {init} {sp} n_steps={n_steps} n_chains={n_chains} patience={patience} lanes={lanes} seed0={seed0}"
        p = abi.make_params(N, n_steps, init, sp, n_chains, mcmc_type=mode, early_stop_patience=patience, lanes_per_chain=lanes)
        seeds = abi.seeds_for(seed0, n_chains)
        want = oracle.run(p, seeds, n_threads=8)
        got, _ = mcq_amd._lib.run_host(p, seeds)
        util.assert_results_equal(got, want, what)
        assert got["near_ties"].sum() == 0, what


def _feature_cases(n=160):
    rng = np.random.default_rng(77001)
    scheds = ["constant", "linear_annealing", "exponential_annealing", "logarithmic_annealing", "sinusoidal_annealing"]

    def sched():
        st = str(rng.choice(scheds))
        if st == "constant":
            return {"type": st, "beta_const": float(rng.choice([0.3, 1.0, 2.5]))}
        return {"type": st, "beta_start": float(rng.choice([0.1, 0.5, 1.0])), "beta_end": float(rng.choice([2.0, 3.0, 6.0]))}

    out = []
    for c in range(n):
        mode = "board" if rng.random() < 0.55 else "full_3d"
        N = int(rng.choice([2, 3, 5, 6, 8, 9, 12, 13, 16, 17, 20, 24]))
        if mode == "full_3d" and N > 17:
            N = int(rng.integers(2, 18))
        n_steps = int(rng.choice([0, 1, 15, 16, 17, 33, 64, 100, 300]))
        patience = int(rng.choice([0, 3, 25, 120])) if mode == "board" and rng.random() < 0.4 else None
        n_sets = int(rng.choice([1, 1, 2, 3]))
        cps = 16 * int(rng.integers(1, 3)) if n_sets > 1 else int(rng.choice([1, 3, 16, 19]))
        sets = [sched() for _ in range(n_sets)]
        inits = [str(rng.choice(["random", "latin", "klarner"])) for _ in range(n_sets)]
        out.append(dict(c=c, N=N, mode=mode, n_steps=n_steps, patience=patience, sets=sets, inits=inits, cps=cps,
                        rng="philox" if rng.random() < 0.5 else "mt19937", reduced=bool(rng.random() < 0.5),
                        lanes=int(rng.choice([0, 4, 8, 16])), seed0=int(rng.integers(0, 2**31))))
    return out


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_random_feature_combinations():
    """The round-2 features crossed at random: Philox stream, reduced trace (against the sums of the oracle's full trace, early
    stops included), schedule sets with their own init modes, every lane width."""
    for k in _feature_cases():
        what = str(k)
        if len(k["sets"]) > 1:
            p = abi.make_params_sets(k["N"], k["n_steps"], k["inits"][0], k["sets"], k["cps"], mcmc_type=k["mode"], early_stop_patience=k["patience"],
                                     lanes_per_chain=k["lanes"], rng=k["rng"], init_modes=k["inits"])
            seeds = np.concatenate([abi.seeds_for(k["seed0"] + 1000 * t, k["cps"]) for t in range(len(k["sets"]))])
        else:
            p = abi.make_params(k["N"], k["n_steps"], k["inits"][0], k["sets"][0], k["cps"], mcmc_type=k["mode"], early_stop_patience=k["patience"],
                                lanes_per_chain=k["lanes"], rng=k["rng"])
            seeds = abi.seeds_for(k["seed0"], k["cps"])
        want = oracle.run(p, seeds, n_threads=8)
        if not k["reduced"]:
            got, _ = mcq_amd._lib.run_host(p, seeds)
            util.assert_results_equal(got, want, what)
            continue
        got, _ = mcq_amd._lib.run_host(p, seeds, trace="reduced")
        util.assert_results_equal(got, want, what, trace=False)
        for t in range(len(k["sets"])):
            sl = slice(t * k["cps"], (t + 1) * k["cps"])
            st = mcq_amd.jobs.stats_from_trace({f: v[sl] for f, v in want.items()}, k["n_steps"])
            for f in ("step_sum", "step_sumsq", "step_accepted", "step_count"):
                g = got[f][t] if len(k["sets"]) > 1 else got[f]
                np.testing.assert_array_equal(g, st[f], err_msg=f"{what}: {f} of set {t}")


def _round3_cases(n=int(os.environ.get("MCQ_FUZZ_R3_CASES", "140"))):
    rng = np.random.default_rng(FUZZ_SEED + 3)
    scheds = ["constant", "linear_annealing", "exponential_annealing", "logarithmic_annealing", "sinusoidal_annealing"]
    out = []
    for c in range(n):
        mode = "board" if rng.random() < 0.5 else "full_3d"
        N = int(rng.choice([2, 3, 3, 4, 4, 5, 5, 6, 9, 12, 17, 19, 24, 33, 40, 57, 64, 100]))  # boards beyond 32: compare-based probes
        if mode == "full_3d" and N > 17:
            N = int(rng.integers(2, 18))
        st = str(rng.choice(scheds))
        sp = {"type": st, "beta_const": float(rng.choice([0.3, 1.0, 2.5]))} if st == "constant" else \
            {"type": st, "beta_start": float(rng.choice([0.1, 0.5, 1.0])), "beta_end": float(rng.choice([2.0, 3.0, 6.0]))}
        n_steps = int(rng.choice([1, 16, 33, 100, 300, 700])) if not FUZZ_LONG else int(rng.choice([300, 1500, 4000]))
        Q = None
        if mode == "full_3d" and N >= 2 and rng.random() < 0.5:
            Q = int(rng.integers(2, min(N ** 3, 1500)))
        init = "random" if Q is not None and Q != N * N else str(rng.choice(["random", "latin", "klarner"]))
        exch = None
        if rng.random() < 0.45:
            R = int(rng.choice([2, 4, 8, 16]))
            exch = (int(rng.choice([1, 2, 7, 50])), R, float(rng.choice([0.5, 0.8, 1.0])), float(rng.choice([1.0, 1.3, 2.0])))
        lanes = int(rng.choice([0, 4, 8, 16]))
        if exch and lanes and 64 // lanes < exch[1]:
            lanes = 0
        if N > 32:  # the heights of 64 / lanes chains must fit the LDS
            lanes = 0 if (exch and exch[1] > 4) or N <= 64 else 16
            if exch and exch[1] > 8 and N > 64:
                exch = None
        n_chains = (exch[1] if exch else 1) * int(rng.choice([1, 2, 5]))
        out.append(dict(c=c, N=N, mode=mode, init=init, sp=sp, n_steps=n_steps, Q=Q, exch=exch, lanes=lanes, n_chains=n_chains,
                        rng="philox" if rng.random() < 0.3 else "mt19937", trace=bool(rng.random() < 0.7), seed0=int(rng.integers(0, 2**31))))
    return out


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_random_round3_feature_combinations():
    """The round-3 features crossed at random: full_3d with Q != N^2 queens, replica exchange (ladders of 2..16, periods 1..50),
    the small boards whose kernels look at five candidates, both streams, with and without a trace, every lane width."""
    for k in _round3_cases():
        what = str(k)
        p = abi.make_params(k["N"], k["n_steps"], k["init"], k["sp"], k["n_chains"], mcmc_type=k["mode"], early_stop_patience=None,
                            lanes_per_chain=k["lanes"], rng=k["rng"], Q=k["Q"], trace=k["trace"])
        if k["exch"]:
            every, R, lo, hi = k["exch"]
            abi.set_exchange(p, every, lo * (hi / lo) ** (np.arange(R) / (R - 1)))
        seeds = abi.seeds_for(k["seed0"], k["n_chains"])
        want = oracle.run(p, seeds, trace=k["trace"], n_threads=8, fast=bool(k["c"] & 1))
        got, _ = mcq_amd._lib.run_host(p, seeds, trace=k["trace"])
        util.assert_results_equal(got, want, what, trace=k["trace"])
        if k["exch"]:
            for f in ("exchange_rung", "n_exchanges"):
                np.testing.assert_array_equal(got[f], want[f], err_msg=f"{what}: {f}")
        assert got["near_ties"].sum() == 0 and want["near_ties"].sum() == 0, what


def _round4_cases(n=int(os.environ.get("MCQ_FUZZ_R4_CASES", "220"))):
    rng = np.random.default_rng(FUZZ_SEED + 4)
    scheds = ["constant", "linear_annealing", "exponential_annealing", "logarithmic_annealing", "sinusoidal_annealing"]
    out = []
    for c in range(n):
        kind = str(rng.choice(["two_lanes", "slim", "counters", "lanes16"]))
        st = str(rng.choice(scheds))
        sp = {"type": st, "beta_const": float(rng.choice([0.0, 0.3, 1.0, 2.5]))} if st == "constant" else \
            {"type": st, "beta_start": float(rng.choice([0.1, 0.5, 1.0])), "beta_end": float(rng.choice([2.0, 3.0, 6.0]))}
        n_steps = int(rng.choice([0, 1, 15, 16, 17, 31, 33, 64, 65, 100, 300, 700])) if not FUZZ_LONG else int(rng.choice([300, 1500, 4000]))
        k = dict(c=c, kind=kind, sp=sp, n_steps=n_steps, mode="board", Q=None, patience=None, flags=0, lanes=0, rng="mt19937", trace=True,
                 init=str(rng.choice(["random", "latin", "klarner"])), seed0=int(rng.integers(0, 2**31)))
        if kind == "two_lanes":  # boards at 32 chains per wavefront: chain counts around 32 and 64, every stream / trace / early-stop form
            k.update(N=int(rng.choice([2, 3, 5, 8, 9, 11, 12, 12, 13, 16, 17, 24, 32])), lanes=2, n_chains=int(rng.choice([1, 2, 31, 32, 33, 63, 65])),
                     rng="philox" if rng.random() < 0.25 else "mt19937", trace=[True, False, "reduced"][int(rng.integers(0, 3))])
            if rng.random() < 0.35:
                k["patience"] = int(rng.choice([0, 3, 25, 120]))
        elif kind == "slim":  # full_3d N = 9..12 at the library's lane choice (4: queens in global memory), any queen count, trace or none
            N = int(rng.choice([9, 10, 11, 12, 12]))
            k.update(N=N, mode="full_3d", n_chains=int(rng.choice([1, 3, 15, 16, 17, 40])), trace=bool(rng.random() < 0.7), lanes=int(rng.choice([0, 0, 4])))
            if rng.random() < 0.4:
                k.update(Q=int(rng.integers(2, min(N ** 3, 1400))), init="random")
        elif kind == "counters":  # dE from LDS line counters: boards up to N = 8 at 4 lanes
            k.update(N=int(rng.integers(2, 9)), lanes=4, flags=abi.FLAG_LINE_COUNTERS, n_chains=int(rng.choice([1, 5, 16, 17, 33])), trace=bool(rng.random() < 0.7))
            if rng.random() < 0.4:
                k["patience"] = int(rng.choice([0, 3, 25, 120]))
        else:  # the unrolled 16-lane boards (one packed pass up to N = 16, two unpacked ones up to 32), with a launch priority
            k.update(N=int(rng.choice([3, 8, 12, 16, 17, 20, 24, 25, 32])), lanes=16, n_chains=int(rng.choice([1, 4, 5, 9])), flags=abi.flag_priority(int(rng.integers(0, 4))) | (abi.FLAG_SHARED_PACING if rng.random() < 0.5 else 0),
                     trace=[True, False, "reduced"][int(rng.integers(0, 3))])
        out.append(k)
    return out


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_random_round4_feature_combinations():
    """The round-4 variants crossed at random: two lanes per chain, the slim full_3d kernels (queen table in global memory), dE from line
    counters, the unrolled 16-lane boards, launch priorities."""
    for k in _round4_cases():
        what = str(k)
        p = abi.make_params(k["N"], k["n_steps"], k["init"], k["sp"], k["n_chains"], mcmc_type=k["mode"], early_stop_patience=k["patience"],
                            lanes_per_chain=k["lanes"], rng=k["rng"], Q=k["Q"], trace=k["trace"], flags=k["flags"])
        seeds = abi.seeds_for(k["seed0"], k["n_chains"])
        want = oracle.run(p, seeds, n_threads=8, fast=bool(k["c"] & 1))
        if k["trace"] == "reduced":
            got, _ = mcq_amd._lib.run_host(p, seeds, trace="reduced")
            util.assert_results_equal(got, want, what, trace=False)
            st = mcq_amd.jobs.stats_from_trace(want, k["n_steps"])
            for f in ("step_sum", "step_sumsq", "step_accepted", "step_count"):
                np.testing.assert_array_equal(got[f], st[f], err_msg=f"{what}: {f}")
        else:
            got, _ = mcq_amd._lib.run_host(p, seeds, trace=k["trace"])
            util.assert_results_equal(got, want, what, trace=bool(k["trace"]))
        assert got["near_ties"].sum() == 0, what


def _wide_cases(n=int(os.environ.get("MCQ_FUZZ_WIDE_CASES", "40"))):
    rng = np.random.default_rng(FUZZ_SEED + 5)
    scheds = ["constant", "linear_annealing", "exponential_annealing", "logarithmic_annealing", "sinusoidal_annealing"]
    out = []
    for c in range(n):
        st = str(rng.choice(scheds))
        sp = {"type": st, "beta_const": float(rng.choice([0.0, 0.3, 1.0, 2.5]))} if st == "constant" else \
            {"type": st, "beta_start": float(rng.choice([0.1, 0.5, 1.0])), "beta_end": float(rng.choice([2.0, 3.0, 6.0]))}
        N = int(rng.choice([33, 34, 40, 41, 47, 48, 56, 63, 64]))
        k = dict(c=c, N=N, sp=sp, n_steps=int(rng.choice([0, 1, 16, 17, 100, 300, 700])) if not FUZZ_LONG else int(rng.choice([1500, 4000])),
                 init=str(rng.choice(["random", "latin", "klarner"])), n_chains=int(rng.choice([1, 3, 4, 5, 9])), Q=None,
                 trace=[True, False, "reduced"][int(rng.integers(0, 3))], lanes=int(rng.choice([0, 16])), seed0=int(rng.integers(0, 2**31)))
        if rng.random() < 0.3:
            k.update(Q=int(rng.integers(2, 32768)), init="random")
        out.append(k)
    return out


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_random_full_3d_beyond_32():
    """full_3d at N = 33..64 (64-bit column words; queens and the init kernel's permutation in global memory) at random: every init, schedule
    and trace mode, queen counts up to 32 767, chain counts around a wavefront's four."""
    for k in _wide_cases():
        what = str(k)
        p = abi.make_params(k["N"], k["n_steps"], k["init"], k["sp"], k["n_chains"], mcmc_type="full_3d", lanes_per_chain=k["lanes"], Q=k["Q"], trace=k["trace"])
        seeds = abi.seeds_for(k["seed0"], k["n_chains"])
        want = oracle.run(p, seeds, n_threads=8, fast=bool(k["c"] & 1))
        if k["trace"] == "reduced":
            got, _ = mcq_amd._lib.run_host(p, seeds, trace="reduced")
            util.assert_results_equal(got, want, what, trace=False)
            st = mcq_amd.jobs.stats_from_trace(want, k["n_steps"])
            for f in ("step_sum", "step_sumsq", "step_accepted", "step_count"):
                np.testing.assert_array_equal(got[f], st[f], err_msg=f"{what}: {f}")
        else:
            got, _ = mcq_amd._lib.run_host(p, seeds, trace=k["trace"])
            util.assert_results_equal(got, want, what, trace=bool(k["trace"]))
        assert got["near_ties"].sum() == 0, what


def _stream_cases(n=int(os.environ.get("MCQ_FUZZ_STREAM_CASES", "60"))):
    rng = np.random.default_rng(FUZZ_SEED + 6)
    scheds = ["constant", "linear_annealing", "exponential_annealing", "logarithmic_annealing", "sinusoidal_annealing"]
    out = []
    for c in range(n):
        st = str(rng.choice(scheds))
        sp = {"type": st, "beta_const": float(rng.choice([0.0, 0.3, 1.0, 2.5]))} if st == "constant" else \
            {"type": st, "beta_start": float(rng.choice([0.1, 0.5, 1.0])), "beta_end": float(rng.choice([2.0, 3.0, 6.0]))}
        mode = "board" if rng.random() < 0.55 else "full_3d"
        N = int(rng.choice([2, 3, 5, 8, 9, 12, 13, 16, 17, 24, 33, 40])) if mode == "board" else int(rng.choice([2, 3, 6, 9, 10, 12, 13, 16, 20, 33, 41]))
        wide = mode == "full_3d" and N > 32
        lanes = int(rng.choice([0, 16])) if wide else int(rng.choice([0, 8, 16])) if N > 32 else int(rng.choice([0, 2, 4, 8, 16] if mode == "board" else [0, 4, 8, 16]))
        n_chains = int(rng.choice([1, 3, 16, 17, 33]))
        if lanes == 2 and n_chains > 1:
            n_chains = int(rng.choice([31, 32, 33, 64]))
        k = dict(c=c, N=N, mode=mode, sp=sp, lanes=lanes, n_chains=n_chains, init=str(rng.choice(["random", "latin", "klarner"])),
                 n_steps=int(rng.choice([0, 1, 16, 17, 100, 300, 700])) if not FUZZ_LONG else int(rng.choice([1500, 4000])),
                 patience=int(rng.choice([0, 3, 25, 120])) if mode == "board" and rng.random() < 0.3 else None,
                 trace=[True, False, "reduced"][int(rng.integers(0, 3))], seed0=int(rng.integers(0, 2**31)))
        out.append(k)
    return out


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_random_continued_streams():
    """Chains that continue MT19937 states (mcq_params.stream_states; the reference's seed=None) at random: states at random positions of random
    generations -- the block edges 0 / 64k / 624 among them --, both modes, every lane width, early stops, every trace mode; results AND the words
    taken from the stream (stream_words) against the oracle."""
    for k in _stream_cases():
        what = str(k)
        rs = np.random.RandomState(k["seed0"] % (2**32))
        states = np.zeros((k["n_chains"], 625), dtype=np.uint32)
        for r in range(k["n_chains"]):
            rs.randint(0, 2**32, size=int(rs.randint(1, 1500)), dtype=np.uint32)
            states[r, :624] = rs.get_state()[1]
            states[r, 624] = rs.choice([0, 64, 128, 576, 623, 624, int(rs.randint(0, 625)), int(rs.randint(0, 625))])
        p = abi.set_stream_states(abi.make_params(k["N"], k["n_steps"], k["init"], k["sp"], k["n_chains"], mcmc_type=k["mode"], early_stop_patience=k["patience"],
                                                  lanes_per_chain=k["lanes"], trace=k["trace"]), states)
        seeds = np.zeros(k["n_chains"], dtype=np.uint32)
        want = oracle.run(p, seeds, n_threads=8, fast=bool(k["c"] & 1))
        if k["trace"] == "reduced":
            got, _ = mcq_amd._lib.run_host(p, seeds, trace="reduced")
            util.assert_results_equal(got, want, what, trace=False)
            st = mcq_amd.jobs.stats_from_trace(want, k["n_steps"])
            for f in ("step_sum", "step_sumsq", "step_accepted", "step_count"):
                np.testing.assert_array_equal(got[f], st[f], err_msg=f"{what}: {f}")
        else:
            got, _ = mcq_amd._lib.run_host(p, seeds, trace=k["trace"])
            util.assert_results_equal(got, want, what, trace=bool(k["trace"]))
        assert got["near_ties"].sum() == 0, what

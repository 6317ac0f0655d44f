"""rng = MCQ_RNG_PHILOX4X32_10, the counter-based fast mode (include/mcq.h).  It is NOT a stream of the reference
(which only has NumPy's MT19937, experiments.py:201-239, 288-327), so parity here means: the block function equals
Random123's published known answers, the oracle's word stream equals an independent NumPy restatement, and the HIP
kernels equal the oracle bit for bit in this mode.  The bench and the drivers never default to it."""
import numpy as np
import pytest

import mcq_amd
from oracle import oracle
from tests import util

abi = mcq_amd.abi

# Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key) -> output
KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
    ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
]


def _philox_numpy(c0, key):
    """Vectorised Philox-4x32-10 for counters (c0, 0, 0, 0) and key (key, 0): returns [len(c0), 4] uint32."""
    c = [np.asarray(c0, dtype=np.uint64), np.zeros(len(c0), np.uint64), np.zeros(len(c0), np.uint64), np.zeros(len(c0), np.uint64)]
    k0, k1 = np.uint64(key), np.uint64(0)
    M = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & M, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & M]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    return np.stack(c, axis=1).astype(np.uint32)


def test_block_function_known_answers():
    for ctr, key, want in KAT:
        np.testing.assert_array_equal(oracle.philox_block(ctr, key), np.array(want, dtype=np.uint32))


def test_word_stream_against_numpy_restatement():
    for seed in (0, 1, 42, 2**32 - 1):
        words = _philox_numpy(np.arange(600), seed).reshape(-1)  # word w = block w // 4, element w % 4
        np.testing.assert_array_equal(oracle.rng_stream(seed, "u32", 2400, rng="philox"), words)
        for m in (1, 2, 5, 11, 23, 143, 1727):  # RandomState.randint semantics on this stream: masked rejection
            mask = (1 << int(m).bit_length()) - 1
            v = words & mask
            np.testing.assert_array_equal(oracle.rng_stream(seed, "bounded", 300, arg=m, rng="philox"), v[v <= m][:300])
        a, b = words[0::2] >> 5, words[1::2] >> 6  # RandomState.random semantics: 53 bits from two words
        want = (a.astype(np.float64) * 67108864.0 + b) / 9007199254740992.0
        np.testing.assert_array_equal(oracle.rng_stream(seed, "double", 1200, rng="philox"), want)


def test_philox_chains_differ_from_the_reference_stream_only_in_the_draws():
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    seeds = abi.seeds_for(42, 6)
    for mode in ("board", "full_3d"):
        mt = oracle.run(abi.make_params(7, 400, "latin", sp, 6, mcmc_type=mode), seeds)
        ph = oracle.run(abi.make_params(7, 400, "latin", sp, 6, mcmc_type=mode, rng="philox"), seeds)
        np.testing.assert_array_equal(mt["initial_energy"], ph["initial_energy"])  # latin init draws nothing
        assert not np.array_equal(mt["energy_hist"], ph["energy_hist"])
        ph2 = oracle.run(abi.make_params(7, 400, "latin", sp, 6, mcmc_type=mode, rng="philox"), seeds)
        util.assert_results_equal(ph, ph2, "philox oracle is deterministic")
    with pytest.raises(ValueError):
        abi.make_params(7, 10, "latin", sp, 1, rng="xoshiro")


PHILOX_CASES = [
    # (N, mode, init, schedule, n_steps, n_chains, patience)
    (12, "board", "random", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, 3000, 150, None),
    (12, "full_3d", "random", {"type": "exponential_annealing", "beta_start": 1.0, "beta_end": 3.0}, 1500, 70, None),
    (24, "board", "random", {"type": "sinusoidal_annealing", "beta_start": 0.1, "beta_end": 5.0}, 800, 40, None),
    (7, "board", "klarner", {"type": "logarithmic_annealing", "beta_start": 0.5, "beta_end": 3.0}, 2000, 33, 120),
    (5, "full_3d", "klarner", {"type": "constant", "beta_const": 0.7}, 1000, 21, None),
    (20, "full_3d", "latin", {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 2.0}, 300, 9, None),
    (2, "board", "random", {"type": "constant", "beta_const": 2.0}, 500, 6, None),
    (2, "full_3d", "random", {"type": "constant", "beta_const": 2.0}, 500, 6, None),
    (9, "board", "random", {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 3.0}, 1500, 19, 0),
    (16, "board", "random", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, 800, 20, None),
    (17, "full_3d", "random", {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, 300, 6, None),
    (32, "full_3d", "random", {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 2.0}, 150, 3, None),
    (31, "board", "klarner", {"type": "sinusoidal_annealing", "beta_start": 0.5, "beta_end": 3.0}, 300, 5, 80),
]


@pytest.mark.gpu
@pytest.mark.parametrize("lanes", (4, 8, 16))
@pytest.mark.parametrize("case", PHILOX_CASES, ids=lambda c: f"N{c[0]}-{c[1]}-{c[2]}-{c[3]['type']}")
def test_hip_equals_oracle_philox(case, lanes):
    N, mode, init, sp, n_steps, n_chains, patience = case
    p = abi.make_params(N, n_steps, init, sp, n_chains, mcmc_type=mode, early_stop_patience=patience, lanes_per_chain=lanes, rng="philox")
    seeds = abi.seeds_for(7000 + 13 * N, n_chains)
    want = oracle.run(p, seeds, n_threads=8)
    got, _ = mcq_amd._lib.run_host(p, seeds)
    util.assert_results_equal(got, want, f"philox hip G={lanes} vs oracle {case}")
    assert got["near_ties"].sum() == 0


@pytest.mark.gpu
def test_philox_long_run_and_flags():
    """25 000 steps (thousands of ring wrap-arounds), the sequential-draw path, and the reduced trace in Philox mode."""
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    for mode, n_chains in (("board", 192), ("full_3d", 96)):
        p = abi.make_params(12, 25000, "random", sp, n_chains, mcmc_type=mode, rng="philox")
        seeds = abi.seeds_for(99, n_chains)
        want = oracle.run(p, seeds, n_threads=16)
        got, _ = mcq_amd._lib.run_host(p, seeds)
        util.assert_results_equal(got, want, f"philox long run {mode}")
        seq, _ = mcq_amd._lib.run_host(abi.make_params(12, 25000, "random", sp, n_chains, mcmc_type=mode, rng="philox",
                                                       flags=abi.FLAG_SEQUENTIAL_DRAWS), seeds)
        util.assert_results_equal(seq, want, f"philox sequential draws {mode}")
        red, _ = mcq_amd._lib.run_host(p, seeds, trace="reduced", states=False)
        L = want["hist_len"]
        h = want["energy_hist"][:, : p.n_steps + 1].astype(np.int64)
        assert (L == p.n_steps + 1).all()
        np.testing.assert_array_equal(red["step_sum"], h.sum(axis=0))
        np.testing.assert_array_equal(red["step_sumsq"], (h * h).sum(axis=0))
        np.testing.assert_array_equal(red["best_energy"], want["best_energy"])

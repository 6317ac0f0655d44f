"""The oracle's restatement of NumPy's legacy RandomState against (a) fixtures captured in the build
container and (b) NumPy itself, which is importable wherever the tests run (it is a third-party
dependency of the reference, requirements.txt:1, not part of it)."""
import numpy as np

from oracle import oracle


def test_fixture_streams(golden):
    z = golden.npz("rng")
    for seed in golden.manifest["rng"]["seeds"]:
        for N in (2, 3, 6, 12, 16, 17, 20, 24):
            np.testing.assert_array_equal(oracle.rng_stream(seed, "bounded", 2000, arg=N - 1), z[f"randint_s{seed}_N{N}"])
        np.testing.assert_array_equal(oracle.rng_stream(seed, "double", 2000), z[f"random_s{seed}"])
        for N in (3, 6, 12):
            np.testing.assert_array_equal(oracle.rng_stream(seed, "bounded", N * N, arg=N - 1), z[f"grid_s{seed}_N{N}"].reshape(-1))


def test_against_live_numpy():
    for seed in (0, 7, 123456789, 2**32 - 1):
        rs = np.random.RandomState(seed)
        want = rs.randint(0, 2**32, size=1500, dtype=np.uint64).astype(np.uint32)  # raw 32-bit words
        np.testing.assert_array_equal(oracle.rng_stream(seed, "u32", 1500), want)
        np.random.seed(seed)
        want = np.array([np.random.randint(0, 144) for _ in range(700)])
        np.testing.assert_array_equal(oracle.rng_stream(seed, "bounded", 700, arg=143), want)
        np.random.seed(seed)
        want = np.array([np.random.random() for _ in range(700)])
        np.testing.assert_array_equal(oracle.rng_stream(seed, "double", 700), want)


def test_choice_without_replacement_is_a_tail_shuffle(golden):
    """np.random.choice(n, Q, replace=False) == permutation(n)[:Q]: the full_3d random init (mcmc.py:97)
    decoded from the oracle's final_state equals the fixture's flat indices."""
    import mcq_amd

    abi = mcq_amd.abi
    z = golden.npz("rng")
    sp = {"type": "constant", "beta_const": 1.0}
    for seed in golden.manifest["rng"]["seeds"]:
        for N in (3, 6, 12):
            p = abi.make_params(N, 0, "random", sp, 1, mcmc_type="full_3d")
            st = oracle.run(p, np.array([seed], dtype=np.uint32))["final_state"][0].reshape(-1, 3).astype(np.int64)
            flat = st[:, 0] * N * N + st[:, 1] * N + st[:, 2]
            np.testing.assert_array_equal(flat, z[f"choice_s{seed}_N{N}"])

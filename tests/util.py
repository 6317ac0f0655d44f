"""Shared helpers of the parity tests: golden-fixture access and field-by-field comparison."""
import json
import os

import numpy as np

import mcq_amd

abi = mcq_amd.abi
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    """tests/golden/*: vectors captured from the reference by tools/gen_golden.py."""

    def __init__(self):
        with open(os.path.join(GOLDEN_DIR, "manifest.json")) as f:
            self.manifest = json.load(f)
        self._npz = {}

    def npz(self, name):
        if name not in self._npz:
            self._npz[name] = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        return self._npz[name]

    @property
    def chains(self):
        return self.manifest["chains"]

    @property
    def chains_q(self):
        """full_3d chains with Q != N^2 queens (metropolis_mcmc(..., Q=...), experiments.py:199-203)"""
        return self.manifest["chains_q"]

    @property
    def chains_big(self):
        """board chains beyond N = 32 (the reference is unbounded, mcmc_board.py:12)"""
        return self.manifest["chains_big"]

    @property
    def chains_wide(self):
        """full_3d chains beyond N = 32 (the reference is unbounded, mcmc.py:6-18)"""
        return self.manifest["chains_wide"]

    @property
    def chains_stream(self):
        """chains that continue NumPy's global stream (metropolis_mcmc[_board](..., seed=None), experiments.py:200-201, 287-288)"""
        return self.manifest["chains_stream"]

    def stream_state(self, case):
        """(uint32[625] key words + position the chain started from, the four words the global stream yields after the chain)"""
        z = self.npz("chains_stream")
        return z[f"{case['key']}_state"], z[f"{case['key']}_after"]

    def chain(self, case):
        key = case["key"]
        z = self.npz("chains_q" if key.startswith("q") else "chains_big" if key.startswith("big") else "chains_wide" if key.startswith("wide") else
                     "chains_stream" if key.startswith("stream") else "chains")
        return {k: z[f"{case['key']}_{k}"] for k in
                ("hist", "accept", "n_executed", "best_energy", "final_energy", "steps_to_best", "best_state", "final_state")}


def params_for_case(case, n_chains=1, **kw):
    return abi.make_params(case["N"], case["n_steps"], case["init"], case["schedule"], n_chains,
                           mcmc_type=case["mode"], early_stop_patience=case.get("patience"), Q=case.get("Q"), **kw)


def words_after(state625, n_words, count=4):
    """The `count` 32-bit words an MT19937 stream yields after `n_words` more have been taken from the given state (NumPy itself does the walking)."""
    rs = np.random.RandomState()
    rs.set_state(("MT19937", np.asarray(state625[:624], dtype=np.uint32), int(state625[624])))
    left = int(n_words)
    while left > 0:
        step = min(left, 1 << 22)
        rs.randint(0, 2**32, size=step, dtype=np.uint32)
        left -= step
    return rs.randint(0, 2**32, size=count, dtype=np.uint32)


def accept_bytes(bits_row, n_steps):
    """uint64 accept words -> the little-endian packed bytes the fixtures store."""
    b = np.ascontiguousarray(bits_row).view(np.uint8)
    return b[: (n_steps + 7) // 8]


def assert_chain_equals_golden(res, r, case, gold, what):
    """Compare chain r of a result dict (oracle or HIP) with a golden chain, bit for bit."""
    n = case["n_steps"]
    L = int(res["hist_len"][r])
    assert L == len(gold["hist"]), f"{what}: history length {L} != {len(gold['hist'])}"
    np.testing.assert_array_equal(res["energy_hist"][r, :L], gold["hist"], err_msg=f"{what}: energy_history")
    assert int(res["steps_executed"][r]) == int(gold["n_executed"]), f"{what}: steps executed"
    np.testing.assert_array_equal(accept_bytes(res["accept_bits"][r], n), gold["accept"], err_msg=f"{what}: accept bits")
    assert int(res["initial_energy"][r]) == int(gold["hist"][0]), what
    assert int(res["best_energy"][r]) == int(gold["best_energy"]), f"{what}: best_energy"
    assert int(res["final_energy"][r]) == int(gold["final_energy"]), f"{what}: final_energy"
    assert int(res["steps_to_best"][r]) == int(gold["steps_to_best"]), f"{what}: steps_to_best"
    np.testing.assert_array_equal(res["best_state"][r], gold["best_state"], err_msg=f"{what}: best_state")
    np.testing.assert_array_equal(res["final_state"][r], gold["final_state"], err_msg=f"{what}: final_state")
    assert int(res["near_ties"][r]) == 0, f"{what}: near tie between uniform and acceptance probability"


RESULT_FIELDS = ("hist_len", "steps_executed", "initial_energy", "best_energy", "final_energy", "steps_to_best",
                 "n_accepted", "best_state", "final_state", "stream_words")


def assert_results_equal(a, b, what, trace=True):
    """Two result dicts (e.g. HIP vs oracle) agree on every integer output."""
    for k in RESULT_FIELDS:
        if k in a and k in b:
            np.testing.assert_array_equal(a[k], b[k], err_msg=f"{what}: {k}")
    if trace:
        for r in range(len(a["hist_len"])):
            L = int(a["hist_len"][r])
            np.testing.assert_array_equal(a["energy_hist"][r, :L], b["energy_hist"][r, :L], err_msg=f"{what}: energy_hist[{r}]")
        np.testing.assert_array_equal(a["accept_bits"], b["accept_bits"], err_msg=f"{what}: accept_bits")

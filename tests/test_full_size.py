"""BASELINE.json's full problem sizes on the GPU, checked through properties that do not need a second full run:

* the energy of the returned final / best state, recounted pair by pair (mcmc_board.py:82-122, mcmc.py:134-169), equals
  the final / best energy the sweep accumulated from 10^5 incremental dE values per chain;
* energy_history is consistent with itself and with the accept bits (entry 0 = E0, last entry = final energy, minimum
  and its first index = best energy / steps_to_best, every change of energy sits on an accepted step, popcount of the
  bits = n_accepted);
* a sample of chains from both ends of the batch equals the oracle bit for bit.

Everything stays on the device (the 26 GB trace never crosses PCIe); the checks run in chunks of chains.
"""
import numpy as np
import pytest

import mcq_amd
from mcq_amd import abi
from oracle import oracle

CHUNK = 4096


def _pair_tables(torch, N, dev):
    """index pairs a < b of the N*N columns (board) with their in-plane offsets"""
    q = np.arange(N * N)
    a, b = np.triu_indices(N * N, k=1)
    di = np.abs(q[a] // N - q[b] // N)
    dj = np.abs(q[a] % N - q[b] % N)
    to = lambda x: torch.from_numpy(x.astype(np.int64)).to(dev)
    return to(a), to(b), to(di).to(torch.int16), to(dj).to(torch.int16)


def _attacks(torch, di, dj, dk):
    """two distinct cells attack iff every non-zero coordinate offset has the same magnitude"""
    m = torch.maximum(torch.maximum(di, dj), dk)
    return ((di == 0) | (di == m)) & ((dj == 0) | (dj == m)) & ((dk == 0) | (dk == m))


def _board_energy(torch, heights, N, tabs):
    a, b, di, dj = tabs
    h = heights.to(torch.int16)
    dk = (h[:, a] - h[:, b]).abs()
    return _attacks(torch, di[None, :], dj[None, :], dk).sum(dim=1)


def _full3d_energy(torch, coords, Q):
    c = coords.reshape(-1, Q, 3).to(torch.int16)
    a, b = np.triu_indices(Q, k=1)
    a, b = torch.from_numpy(a).to(c.device), torch.from_numpy(b).to(c.device)
    d = (c[:, a, :] - c[:, b, :]).abs()
    assert bool((d.sum(dim=2) > 0).all()), "two queens on one cell"
    return _attacks(torch, d[:, :, 0], d[:, :, 1], d[:, :, 2]).sum(dim=1)


def _check_run(mode, N, n_steps, n_chains, sp, n_sample):
    import torch

    p = abi.make_params(N, n_steps, "random", sp, n_chains, mcmc_type=mode, early_stop_patience=None)
    seeds = abi.seeds_for(42, n_chains)
    run = mcq_amd._lib.DeviceRun(p, seeds, trace=True, states=True)
    run.launch()
    torch.cuda.synchronize()
    t = run.t
    dev = t["best_energy"].device
    Q = N * N
    tabs = _pair_tables(torch, N, dev) if mode == "board" else None
    bits8 = t["accept_bits"].view(torch.uint8)  # little-endian: step s is bit s % 8 of byte s // 8
    shifts = torch.arange(8, device=dev, dtype=torch.uint8)
    assert bool((t["hist_len"] == n_steps + 1).all()) and bool((t["steps_executed"] == n_steps).all())
    for lo in range(0, n_chains, CHUNK):
        sl = slice(lo, min(n_chains, lo + CHUNK))
        hist = t["energy_hist"][sl, : n_steps + 1]
        # states against energies
        if mode == "board":
            e_final = _board_energy(torch, t["final_state"][sl], N, tabs)
            e_best = _board_energy(torch, t["best_state"][sl], N, tabs)
        else:
            e_final = _full3d_energy(torch, t["final_state"][sl], Q)
            e_best = _full3d_energy(torch, t["best_state"][sl], Q)
        assert torch.equal(e_final, t["final_energy"][sl].to(torch.int64)), f"final state energy, chains {lo}.."
        assert torch.equal(e_best, t["best_energy"][sl].to(torch.int64)), f"best state energy, chains {lo}.."
        # the trace against the scalars
        assert torch.equal(hist[:, 0], t["initial_energy"][sl])
        assert torch.equal(hist[:, n_steps], t["final_energy"][sl])
        best = hist.min(dim=1).values
        assert torch.equal(best, t["best_energy"][sl])
        first = (hist == best[:, None]).to(torch.int8).argmax(dim=1)  # first index of the minimum (experiments.py:364-365)
        assert torch.equal(first, t["steps_to_best"][sl])
        # the trace against the accept bits
        acc = ((bits8[sl, :, None] >> shifts) & 1).reshape(hist.shape[0], -1)[:, :n_steps].bool()
        changed = hist[:, 1:] != hist[:, :-1]
        assert not bool((changed & ~acc).any()), "energy changed on a rejected step"
        assert torch.equal(acc.sum(dim=1), t["n_accepted"][sl])
        del acc, changed, first, best
    # both ends of the batch against the oracle
    idx = np.r_[0:n_sample, n_chains - n_sample:n_chains]
    ps = abi.make_params(N, n_steps, "random", sp, len(idx), mcmc_type=mode, early_stop_patience=None)
    want = oracle.run(ps, seeds[idx], n_threads=8)
    ti = torch.from_numpy(idx).to(dev)
    for k in ("initial_energy", "best_energy", "final_energy", "steps_to_best", "n_accepted", "near_ties", "best_state", "final_state"):
        np.testing.assert_array_equal(t[k][ti].cpu().numpy(), want[k], err_msg=k)
    np.testing.assert_array_equal(t["energy_hist"][ti].cpu().numpy()[:, : n_steps + 1], want["energy_hist"][:, : n_steps + 1])
    np.testing.assert_array_equal(t["accept_bits"][ti].cpu().numpy().view(np.uint64), want["accept_bits"])
    assert int(t["near_ties"].sum().item()) == 0


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_headline_size_board():
    """BASELINE.json configs[1]: N=12 board, linear 1 -> 3, 65 536 chains x 100 000 steps, seeds 42 + r, full trace"""
    _check_run("board", 12, 100000, 65536, {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, n_sample=16)


@pytest.mark.gpu
@pytest.mark.timeout(1500)
def test_config3_size_full3d():
    """BASELINE.json configs[2] at a fifth of its length: N=12 full_3d, exponential 1 -> 3, 65 536 chains x 20 000 steps"""
    _check_run("full_3d", 12, 20000, 65536, {"type": "exponential_annealing", "beta_start": 1.0, "beta_end": 3.0}, n_sample=8)

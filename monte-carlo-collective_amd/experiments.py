"""Drop-in for the sweep path of the reference's experiments.py.

Same entry points, argument meaning, return shapes and error behaviour as the reference
(cited per function), but every chain of a call runs in ONE launch of the HIP kernels in
csrc/ through the C-ABI of include/mcq.h -- there is no process pool and no CPU path.

    from mcq_amd.experiments import run_experiment, build_schedule_from_params
"""
import time

import numpy as np

from . import _lib, abi


# --------------------------------------------------------------------------------------------
# beta schedules (experiments.py:13-105).  The reference returns closures; these are callable
# objects that also carry the picklable `params` dict the GPU path consumes.  Their __call__
# evaluates the same float64 expressions on the host (for plots / inspection); the sweep itself
# evaluates beta on the device.
# --------------------------------------------------------------------------------------------
class BetaSchedule:
    def __init__(self, params, n_steps):
        self.params = dict(params)
        self.n_steps = int(n_steps)

    def __call__(self, step):
        t, n = self.params["type"], self.n_steps
        if t == "constant":
            return self.params["beta_const"]
        bs, be = self.params["beta_start"], self.params["beta_end"]
        if n <= 1:
            return be
        if t == "linear_annealing":
            frac = step / (n - 1)
            return bs + frac * (be - bs)
        if t == "exponential_annealing":
            s = min(max(step, 0), n - 1)
            return bs * np.exp(np.log(be / bs) * (s / (n - 1)))
        if t == "logarithmic_annealing":
            s = min(max(step, 0), n)
            return bs + (be - bs) * (np.log(1 + s) / np.log(1 + n))
        s = min(max(step, 0), n)
        return bs + (be - bs) * (1 - np.cos(np.pi * s / n)) / 2

    def __repr__(self):
        return f"BetaSchedule({self.params}, n_steps={self.n_steps})"


def constant_beta(beta):
    return BetaSchedule({"type": "constant", "beta_const": beta}, 0)


def linear_annealing_beta(beta_start, beta_end, n_steps):
    return BetaSchedule({"type": "linear_annealing", "beta_start": beta_start, "beta_end": beta_end}, n_steps)


def exponential_annealing_beta(beta_start, beta_end, n_steps):
    return BetaSchedule({"type": "exponential_annealing", "beta_start": beta_start, "beta_end": beta_end}, n_steps)


def logarithmic_annealing_beta(beta_start, beta_end, n_steps):
    return BetaSchedule({"type": "logarithmic_annealing", "beta_start": beta_start, "beta_end": beta_end}, n_steps)


def sinusoidal_annealing_beta(beta_start, beta_end, n_steps):
    return BetaSchedule({"type": "sinusoidal_annealing", "beta_start": beta_start, "beta_end": beta_end}, n_steps)


def build_schedule_from_params(sched_type, n_steps, beta_const=None, beta_start=None, beta_end=None):
    """experiments.py:79-105, including its ValueErrors."""
    if sched_type == "constant":
        if beta_const is None:
            raise ValueError("beta_const required for constant schedule")
        return BetaSchedule({"type": "constant", "beta_const": beta_const}, n_steps)
    if sched_type in ("linear_annealing", "exponential_annealing", "logarithmic_annealing", "sinusoidal_annealing"):
        if beta_start is None or beta_end is None:
            raise ValueError(f"beta_start and beta_end required for {sched_type} schedule")
        return BetaSchedule({"type": sched_type, "beta_start": beta_start, "beta_end": beta_end}, n_steps)
    raise ValueError(f"Unknown betta_scheduling type: {sched_type}")


_DESC = {
    "linear_annealing": ("linear beta: {s}->{e}", "Linear {s}->{e}"),
    "exponential_annealing": ("exp beta: {s}->{e}", "Exponential {s}->{e}"),
    "logarithmic_annealing": ("log beta: {s}->{e}", "Logarithmic {s}->{e}"),
    "sinusoidal_annealing": ("sinusoidal beta: {s}->{e}", "Sinusoidal {s}->{e}"),
}


def _describe(sched_type, beta_const, beta_start, beta_end):
    if sched_type == "constant":
        return {"type": "constant", "beta_const": beta_const}, f"constant beta={beta_const}", f"Constant beta={beta_const}"
    if sched_type in _DESC:
        d, l = _DESC[sched_type]
        return ({"type": sched_type, "beta_start": beta_start, "beta_end": beta_end},
                d.format(s=beta_start, e=beta_end), l.format(s=beta_start, e=beta_end))
    raise ValueError(f"Unknown betta_scheduling type: {sched_type}")


def build_schedule_from_common(common_cfg, n_steps):
    """experiments.py:108-152: common['betta_scheduling'] -> (schedule, base_seed, desc, schedule_params)."""
    cfg = common_cfg["betta_scheduling"]
    st = cfg["type"]
    base_seed = cfg.get("base_seed", 0)
    if st == "constant":
        params, desc, _ = _describe(st, cfg["beta_const"], None, None)
    elif st in _DESC:
        params, desc, _ = _describe(st, None, cfg["beta_start"], cfg["beta_end"])
    else:
        raise ValueError(f"Unknown betta_scheduling type: {st}")
    sched = build_schedule_from_params(st, n_steps, params.get("beta_const"), params.get("beta_start"), params.get("beta_end"))
    return sched, base_seed, desc, params


def build_schedules_from_types(sched_types, sched_cfg, n_steps):
    """experiments.py:155-196: requires base_seed, beta_start, beta_end and beta_const keys."""
    base_seed = sched_cfg["base_seed"]
    beta_start, beta_end, beta_const = sched_cfg["beta_start"], sched_cfg["beta_end"], sched_cfg["beta_const"]
    out = []
    for st in sched_types:
        params, desc, label = _describe(st, beta_const, beta_start, beta_end)
        sched = build_schedule_from_params(st, n_steps, params.get("beta_const"), params.get("beta_start"), params.get("beta_end"))
        out.append((sched, base_seed, desc, label, params))
    return out


# --------------------------------------------------------------------------------------------
# states handed back to callers (competition.py:179-187 reads best_state.heights)
# --------------------------------------------------------------------------------------------
class BoardState:
    """Stand-in for State3DQueensBoard: N, Q and the heights array (mcmc_board.py:23-28)."""

    def __init__(self, N, heights, energy):
        self.N, self.Q = N, N * N
        self.heights = np.asarray(heights, dtype=np.int64).reshape(N, N)
        self._energy = int(energy)

    def energy(self, recompute=False):
        return self._energy


class QueensState:
    """Stand-in for State3DQueens: N, Q, queens[Q,3] and the occupied-cell set (mcmc.py:15-18, 101, 113-118)."""

    def __init__(self, N, queens, energy):
        self.queens = np.asarray(queens, dtype=np.int64).reshape(-1, 3)
        self.N, self.Q = N, len(self.queens)
        self.occ_set = {tuple(int(v) for v in q) for q in self.queens}
        self._energy = int(energy)

    def energy(self, recompute=False):
        return self._energy


def _state(mode, N, row, energy):
    return BoardState(N, row, energy) if mode == abi.MODE_BOARD else QueensState(N, row, energy)


# --------------------------------------------------------------------------------------------
# the sweep
# --------------------------------------------------------------------------------------------
def run_chains(N, n_steps, init_mode, schedule_params, seeds, mcmc_type="full_3d", early_stop_patience=None,
               trace=True, states=True, flags=0, lanes_per_chain=0, Q=None, stream_states=None):
    """Lowest Python level: every chain of `seeds` in one GPU launch.  `stream_states` (one np.random.get_state() tuple or uint32[625] row per
    chain): the chains continue those MT19937 streams instead of seeding their own; `stream_words` of the result says how far each went.

    Returns (result dict of NumPy arrays as described in include/mcq.h, kernel seconds)."""
    seeds = np.asarray(seeds)
    params = abi.make_params(N, n_steps, init_mode, schedule_params, len(seeds), mcmc_type=mcmc_type,
                             early_stop_patience=early_stop_patience, trace=trace, flags=flags,
                             lanes_per_chain=lanes_per_chain, Q=Q)
    if stream_states is not None:
        abi.set_stream_states(params, stream_states)
    if seeds.size and (seeds.min() < 0 or seeds.max() > 2**32 - 1):
        raise ValueError("Seed must be between 0 and 2**32 - 1")
    return _lib.run_host(params, seeds.astype(np.uint32), trace=trace, states=states)


def accepted_rejected_steps(res, r):
    """Step indices of accepted / rejected proposals of chain r (experiments.py:329-332)."""
    n = int(res["steps_executed"][r])
    bits = np.unpackbits(res["accept_bits"][r].view(np.uint8), bitorder="little")[:n].astype(bool)
    idx = np.arange(n)
    return idx[bits], idx[~bits]


def _chain_dict(res, r, mode, N):
    """The dict metropolis_mcmc[_board] returns (experiments.py:270-279, 367-376)."""
    L = int(res["hist_len"][r])
    acc, rej = accepted_rejected_steps(res, r)
    return {
        "final_state": _state(mode, N, res["final_state"][r], res["final_energy"][r]),
        "final_energy": int(res["final_energy"][r]),
        "best_state": _state(mode, N, res["best_state"][r], res["best_energy"][r]),
        "best_energy": int(res["best_energy"][r]),
        "energy_history": res["energy_hist"][r, :L],
        "accepted_steps": acc,
        "rejected_steps": rej,
        "steps_to_best": int(res["steps_to_best"][r]),
    }


def _params_of(beta_schedule, schedule_params):
    if schedule_params is not None:
        return schedule_params
    p = getattr(beta_schedule, "params", None)
    if p is None:
        raise ValueError("the GPU path needs schedule_params (or a schedule built by build_schedule_from_params)")
    return p


def _one_chain(seed, **kw):
    """One chain, seeded (np.random.seed(seed), experiments.py:200-201, 287-288) or -- seed=None, the reference's default -- drawn from NumPy's
    global stream where it stands: the chain runs on the device from a copy of np.random.get_state(), and the global stream is then moved past
    the words the chain took (mcq_outputs.stream_words), so whatever draws from it next sees what it would see after the reference's call."""
    if seed is not None:
        return run_chains(seeds=[seed], **kw)[0]
    st = np.random.get_state()
    res, _ = run_chains(seeds=[0], stream_states=st, **kw)
    left = int(res["stream_words"][0])
    while left > 0:  # RandomState.randint over the full 32-bit range takes exactly one word per element
        n = min(left, 1 << 24)
        np.random.randint(0, 2**32, size=n, dtype=np.uint32)
        left -= n
    return res


def metropolis_mcmc_board(N, n_steps, init_mode, beta_schedule, verbose=True, seed=None, run_idx=None,
                          early_stop_patience=None, schedule_params=None):
    """experiments.py:282-376 for one chain; seed=None continues NumPy's global stream like the reference (_one_chain)."""
    res = _one_chain(seed, N=N, n_steps=n_steps, init_mode=init_mode, schedule_params=_params_of(beta_schedule, schedule_params), mcmc_type="board",
                     early_stop_patience=early_stop_patience)
    d = _chain_dict(res, 0, abi.MODE_BOARD, N)
    if verbose and n_steps > 0:
        print(d["final_energy"])
        print(d["best_energy"])
    return d


def metropolis_mcmc(N, n_steps, init_mode, beta_schedule, verbose=True, seed=None, Q=None, run_idx=None,
                    early_stop_patience=None, schedule_params=None):
    """experiments.py:199-279 for one chain; early_stop_patience is accepted and ignored, as there; seed=None continues NumPy's global stream."""
    res = _one_chain(seed, N=N, n_steps=n_steps, init_mode=init_mode, schedule_params=_params_of(beta_schedule, schedule_params), mcmc_type="full_3d", Q=Q)
    d = _chain_dict(res, 0, abi.MODE_FULL3D, N)
    if verbose and n_steps > 0:
        print(d["final_energy"])
        print(d["best_energy"])
    return d


def run_single_chain(N, n_steps, init_mode, beta_schedule, seed=None, verbose=False, run_idx=None, early_stop_patience=None):
    """experiments.py:379-389"""
    return metropolis_mcmc(N=N, n_steps=n_steps, init_mode=init_mode, beta_schedule=beta_schedule, verbose=verbose,
                           seed=seed, run_idx=run_idx, early_stop_patience=early_stop_patience)


def run_single_chain_board(N, n_steps, init_mode, beta_schedule, seed=None, verbose=False, run_idx=None, early_stop_patience=None):
    """experiments.py:392-402"""
    return metropolis_mcmc_board(N=N, n_steps=n_steps, init_mode=init_mode, beta_schedule=beta_schedule, verbose=verbose,
                                 seed=seed, run_idx=run_idx, early_stop_patience=early_stop_patience)


def _multithread(args, board):
    (N, n_steps, init_mode, schedule_params, seed, verbose, run_idx, early_stop_patience) = args
    sched = build_schedule_from_params(schedule_params["type"], n_steps, schedule_params.get("beta_const"),
                                       schedule_params.get("beta_start"), schedule_params.get("beta_end"))
    t0 = time.time()
    fn = run_single_chain_board if board else run_single_chain
    res = fn(N=N, n_steps=n_steps, init_mode=init_mode, beta_schedule=sched, seed=seed, verbose=verbose, run_idx=run_idx,
             early_stop_patience=early_stop_patience)
    return {"run_idx": run_idx, "best_state": res["best_state"], "energy_history": res["energy_history"],
            "best_energy": res["best_energy"], "duration": time.time() - t0, "accepted_steps": res["accepted_steps"],
            "rejected_steps": res["rejected_steps"], "steps_to_best": res["steps_to_best"]}


def run_single_chain_multithread(args):
    """experiments.py:405-437"""
    return _multithread(args, board=False)


def run_single_chain_board_multithread(args):
    """experiments.py:440-472"""
    return _multithread(args, board=True)


def run_experiment(N, n_steps, init_mode, beta_schedule, n_runs, base_seed=0, verbose=False, n_workers=None,
                   schedule_params=None, mcmc_type="full_3d", early_stop_patience=100000, return_steps=True,
                   lanes_per_chain=0):
    """experiments.py:475-573.  Chain r is seeded with base_seed + r and results come back ordered by r.

    Returns the reference's 6-tuple (all_histories, best_energies, run_times, all_accepted_steps,
    all_rejected_steps, all_steps_to_best); histories and step lists are NumPy arrays (one per chain,
    ragged after an early stop), which every consumer in the reference accepts
    (np.array(histories), len(histories[0]), list.extend).  `n_workers` is accepted and unused: all
    chains run in one kernel launch.  With return_steps=False the two step-index lists are empty
    (they are O(n_runs * n_steps) on the host).

    Reference behaviours kept: n_runs > 1 requires schedule_params (experiments.py:505-506);
    n_runs == 1 drops early_stop_patience (experiments.py:550-558); full_3d ignores it."""
    if n_runs > 1:
        if schedule_params is None:
            raise ValueError("schedule_params is required for parallel execution when n_runs > 1")
        patience = early_stop_patience
    else:
        schedule_params = _params_of(beta_schedule, schedule_params) if n_runs == 1 else schedule_params
        patience = None
    if n_runs <= 0:
        return [], [], [], [], [], []
    seeds = abi.seeds_for(base_seed, n_runs)
    res, secs = run_chains(N, n_steps, init_mode, schedule_params, seeds, mcmc_type=mcmc_type, early_stop_patience=patience,
                           states=False, lanes_per_chain=lanes_per_chain)
    all_histories = [res["energy_hist"][r, : int(res["hist_len"][r])] for r in range(n_runs)]
    best_energies = [int(b) for b in res["best_energy"]]
    run_times = [secs / n_runs] * n_runs
    all_acc, all_rej = [], []
    if return_steps:
        for r in range(n_runs):
            a, j = accepted_rejected_steps(res, r)
            all_acc.append(a)
            all_rej.append(j)
    else:
        all_acc, all_rej = [[] for _ in range(n_runs)], [[] for _ in range(n_runs)]
    all_steps_to_best = [int(s) for s in res["steps_to_best"]]
    if verbose:
        for b in best_energies:
            print(b)
    return all_histories, best_energies, run_times, all_acc, all_rej, all_steps_to_best

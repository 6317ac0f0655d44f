"""ctypes mirror of include/mcq.h (the C-ABI under run_experiment).

Everything here is plain data: the structs, the enum values, and the mapping from the
reference's string vocabulary (`mcmc_type`, `init_mode`, `betta_scheduling.type`;
experiments.py:79-105, 497-502; mcmc_board.py:26-59) to those enums.
"""
import ctypes as C

import numpy as np

ABI_VERSION = 5

OK, EINVAL, EDEVICE, ENOMEM = 0, -1, -2, -3

MODE_BOARD, MODE_FULL3D = 0, 1
INIT = {"random": 0, "latin": 1, "klarner": 2}
SCHED = {
    "constant": 0,
    "linear_annealing": 1,
    "exponential_annealing": 2,
    "logarithmic_annealing": 3,
    "sinusoidal_annealing": 4,
}
RNG_MT19937_NUMPY = 0
RNG_PHILOX4X32_10 = 1
RNG = {"mt19937": RNG_MT19937_NUMPY, "numpy": RNG_MT19937_NUMPY, "philox": RNG_PHILOX4X32_10}
TRACE_NONE, TRACE_I32, TRACE_REDUCED = 0, 1, 2
FLAG_EXACT_EXP = 1
FLAG_SEQUENTIAL_DRAWS = 2
FLAG_LINE_COUNTERS = 4  # HIP: dE from per-line occupancy counters in LDS (boards up to N = 8 at 4 lanes per chain)
FLAG_SHARED_PACING = 16  # HIP: pace against every launch of the process that sets the flag (one progress table per device)
FLAG_PRIORITY_SHIFT = 8  # bits 8..9: s_setprio level of a small launch's wavefronts (include/mcq.h: MCQ_FLAG_PRIORITY)


def flag_priority(p):
    return (int(p) & 3) << FLAG_PRIORITY_SHIFT


MIN_N, MAX_N, MAX_N_BOARD = 2, 64, 128  # include/mcq.h: full_3d up to 64, boards up to 128
MAX_HIST_STRIDE = 1 << 24  # a full trace row (hist_stride entries) must stay below this: include/mcq.h


class Schedule(C.Structure):
    """include/mcq.h: mcq_schedule -- one beta schedule of a batched run"""
    _fields_ = [
        ("sched", C.c_int32),
        ("init_plus1", C.c_int32),
        ("beta_const", C.c_double),
        ("beta_start", C.c_double),
        ("beta_end", C.c_double),
    ]


class Params(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("N", C.c_int32),
        ("mode", C.c_int32),
        ("init", C.c_int32),
        ("sched", C.c_int32),
        ("rng", C.c_int32),
        ("trace", C.c_int32),
        ("flags", C.c_uint32),
        ("beta_const", C.c_double),
        ("beta_start", C.c_double),
        ("beta_end", C.c_double),
        ("n_steps", C.c_int64),
        ("n_chains", C.c_int64),
        ("patience", C.c_int64),
        ("hist_stride", C.c_int64),
        ("bits_stride", C.c_int64),
        ("lanes_per_chain", C.c_int32),
        ("device", C.c_int32),
        ("n_sets", C.c_int64),
        ("chains_per_set", C.c_int64),
        ("sets", C.POINTER(Schedule)),
        ("beta_table", C.c_void_p),
        ("exchange_every", C.c_int64),
        ("exchange_replicas", C.c_int32),
        ("n_queens", C.c_int32),
        ("exchange_ladder", C.POINTER(C.c_double)),
        ("stream_states", C.c_void_p),
    ]


class PackSlot(C.Structure):
    """include/mcq.h: mcq_pack_slot -- where one job's fields sit in the packed summary tensor (word offsets, -1 = absent)"""
    _fields_ = [("counters", C.c_int64), ("min_slot", C.c_int64), ("best", C.c_int64), ("stb", C.c_int64), ("stats", C.c_int64)]


class Outputs(C.Structure):
    _fields_ = [
        ("energy_hist", C.c_void_p),
        ("accept_bits", C.c_void_p),
        ("hist_len", C.c_void_p),
        ("steps_executed", C.c_void_p),
        ("initial_energy", C.c_void_p),
        ("best_energy", C.c_void_p),
        ("final_energy", C.c_void_p),
        ("steps_to_best", C.c_void_p),
        ("n_accepted", C.c_void_p),
        ("near_ties", C.c_void_p),
        ("best_state", C.c_void_p),
        ("final_state", C.c_void_p),
        ("step_sum", C.c_void_p),
        ("step_sumsq", C.c_void_p),
        ("step_accepted", C.c_void_p),
        ("step_count", C.c_void_p),
        ("exchange_rung", C.c_void_p),
        ("n_exchanges", C.c_void_p),
        ("stream_words", C.c_void_p),
    ]


# field -> (dtype, per-chain shape builder)
OUTPUT_DTYPES = {
    "energy_hist": np.int32,
    "accept_bits": np.uint64,
    "hist_len": np.int64,
    "steps_executed": np.int64,
    "initial_energy": np.int32,
    "best_energy": np.int32,
    "final_energy": np.int32,
    "steps_to_best": np.int64,
    "n_accepted": np.int64,
    "near_ties": np.int64,
    "best_state": np.uint8,
    "final_state": np.uint8,
    "step_sum": np.int64,
    "step_sumsq": np.int64,
    "step_accepted": np.int64,
    "step_count": np.int64,
    "exchange_rung": np.int32,
    "n_exchanges": np.int64,
    "stream_words": np.uint32,
}


def state_bytes(N, mode, n_queens=0):
    """Bytes of one chain's state record: N*N heights (board) or Q*3 coordinates (full_3d; Q = N*N unless n_queens names a count)."""
    return N * N if mode == MODE_BOARD else 3 * (n_queens if n_queens else N * N)


def hist_stride_for(n_steps):
    """Row stride of energy_hist: n_steps + 1 entries rounded up to 64 (256-byte rows, so every
    64-entry block a wavefront flushes is one aligned 256-byte store)."""
    return ((n_steps + 1 + 63) // 64) * 64


def bits_stride_for(n_steps):
    return max(1, (n_steps + 63) // 64)


def output_shapes(p, trace=True, states=True):
    """name -> shape for the arrays a call with parameters `p` fills."""
    n = p.n_chains
    shapes = {k: (n,) for k in ("hist_len", "steps_executed", "initial_energy", "best_energy", "final_energy",
                                "steps_to_best", "n_accepted", "near_ties", "stream_words")}
    if isinstance(trace, str) and trace == "reduced":
        for k in ("step_sum", "step_sumsq", "step_accepted", "step_count"):
            shapes[k] = (p.n_sets, p.n_steps + 1) if p.n_sets > 1 else (p.n_steps + 1,)
    elif trace:
        shapes["energy_hist"] = (n, p.hist_stride)
        shapes["accept_bits"] = (n, p.bits_stride)
    if states:
        sb = state_bytes(p.N, p.mode, p.n_queens)
        shapes["best_state"] = (n, sb)
        shapes["final_state"] = (n, sb)
    if p.exchange_every > 0:
        shapes["exchange_rung"] = (n,)
        shapes["n_exchanges"] = (n,)
    return shapes


def trace_mode(trace):
    """True / False / "reduced" -> MCQ_TRACE_*"""
    if trace == "reduced":
        return TRACE_REDUCED
    return TRACE_I32 if trace else TRACE_NONE


def mode_of(mcmc_type):
    """experiments.py:497-502: "board" selects the board chain, anything else the full_3d chain."""
    return MODE_BOARD if mcmc_type == "board" else MODE_FULL3D


def normalise_patience(early_stop_patience):
    """experiments.py:284-285: None, 'None' and 'null' all disable early stopping."""
    if early_stop_patience in (None, "None", "null"):
        return -1
    v = int(early_stop_patience)
    return v if v >= 0 else 0  # a negative patience stops at the first step, like 0


def _check_schedule(schedule_params):
    """(type, beta_const, beta_start, beta_end) of a betta_scheduling dict; raises like the reference (experiments.py:85-105)"""
    if schedule_params is None:
        raise ValueError("schedule_params is required")
    st = schedule_params.get("type")
    if st not in SCHED:
        raise ValueError(f"Unknown betta_scheduling type: {st}")
    bc = schedule_params.get("beta_const")
    bs = schedule_params.get("beta_start")
    be = schedule_params.get("beta_end")
    if st == "constant":
        if bc is None:
            raise ValueError("beta_const required for constant schedule")
    elif bs is None or be is None:
        raise ValueError(f"beta_start and beta_end required for {st} schedule")
    f = lambda v: float(v) if v is not None else 0.0
    return st, f(bc), f(bs), f(be)


def make_params_sets(N, n_steps, init_mode, schedule_sets, chains_per_set, mcmc_type="full_3d", early_stop_patience=None,
                     trace=True, flags=0, lanes_per_chain=0, device=-1, rng="mt19937", init_modes=None):
    """Parameters of ONE launch that runs `chains_per_set` chains under each schedule of `schedule_sets` (a list of
    betta_scheduling dicts): chains [t * chains_per_set, (t + 1) * chains_per_set) follow schedule t.  What
    run_beta_start_end_pairs does pair by pair (experiments.py:741-846), batched.  `init_modes` (optional, one per set) gives
    every set its own initial state: the (init_mode, N) cells of measure_min_energy_vs_N that share N."""
    sets = [_check_schedule(sp) for sp in schedule_sets]
    if not sets:
        raise ValueError("at least one schedule")
    if chains_per_set <= 0 or chains_per_set % 16:
        raise ValueError("chains_per_set must be a positive multiple of 16")
    p = make_params(N, n_steps, init_mode, schedule_sets[0], chains_per_set * len(sets), mcmc_type=mcmc_type,
                    early_stop_patience=early_stop_patience, trace=trace, flags=flags, lanes_per_chain=lanes_per_chain, device=device,
                    rng=rng)
    arr = (Schedule * len(sets))()
    if init_modes is not None:
        if len(init_modes) != len(sets):
            raise ValueError("init_modes needs one entry per schedule set")
        for im in init_modes:
            if im not in INIT:
                raise ValueError(f"Unknown init_mode: {im}")
    for t, (a, (st, bc, bs, be)) in enumerate(zip(arr, sets)):
        a.sched, a.init_plus1, a.beta_const, a.beta_start, a.beta_end = SCHED[st], 0 if init_modes is None else INIT[init_modes[t]] + 1, bc, bs, be
    p.n_sets, p.chains_per_set = len(sets), chains_per_set
    p.sets = C.cast(arr, C.POINTER(Schedule))
    p._sets_keepalive = arr  # the struct only holds a pointer
    p._schedules = [dict(sp) for sp in schedule_sets]
    return p


def make_params(N, n_steps, init_mode, schedule_params, n_chains, mcmc_type="full_3d", early_stop_patience=None,
                trace=True, flags=0, lanes_per_chain=0, device=-1, rng="mt19937", Q=None):
    """Build a Params from the reference's vocabulary.  Raises ValueError exactly where the
    reference does: unknown schedule type (experiments.py:105), missing beta parameters
    (experiments.py:85-102), unknown init_mode (mcmc_board.py:59, mcmc.py:104)."""
    if schedule_params is None:
        raise ValueError("schedule_params is required")
    st = schedule_params.get("type")
    if st not in SCHED:
        raise ValueError(f"Unknown betta_scheduling type: {st}")
    bc = schedule_params.get("beta_const")
    bs = schedule_params.get("beta_start")
    be = schedule_params.get("beta_end")
    if st == "constant":
        if bc is None:
            raise ValueError("beta_const required for constant schedule")
    elif bs is None or be is None:
        raise ValueError(f"beta_start and beta_end required for {st} schedule")
    if init_mode not in INIT:
        raise ValueError(f"Unknown init_mode: {init_mode}")
    N = int(N)
    top = MAX_N_BOARD if mode_of(mcmc_type) == MODE_BOARD else MAX_N
    if not (MIN_N <= N <= top):
        raise ValueError(f"N must be in [{MIN_N}, {top}] for mcmc_type {mcmc_type}, got {N}")
    n_steps = int(n_steps)
    if n_steps < 0:
        raise ValueError("n_steps must be >= 0")
    p = Params()
    p.abi_version = ABI_VERSION
    p.N = N
    p.mode = mode_of(mcmc_type)
    p.init = INIT[init_mode]
    p.sched = SCHED[st]
    if rng not in RNG:
        raise ValueError(f"Unknown rng: {rng}")
    p.rng = RNG[rng]
    p.trace = trace_mode(trace)
    p.flags = flags
    p.beta_const = float(bc) if bc is not None else 0.0
    p.beta_start = float(bs) if bs is not None else 0.0
    p.beta_end = float(be) if be is not None else 0.0
    p.n_steps = n_steps
    p.n_chains = int(n_chains)
    p.patience = normalise_patience(early_stop_patience) if p.mode == MODE_BOARD else -1
    p.hist_stride = hist_stride_for(n_steps)
    p.bits_stride = bits_stride_for(n_steps)
    p.lanes_per_chain = lanes_per_chain
    p.device = device
    p._schedules = [dict(schedule_params)]  # what abi.beta_table evaluates (the struct itself only holds enums and doubles)
    if Q is not None and int(Q) != N * N:  # State3DQueens(N, Q=...): mcmc.py:6-18
        Q = int(Q)
        if p.mode != MODE_FULL3D:
            raise ValueError("Q applies to mcmc_type full_3d (a board has one queen per column)")
        if init_mode in ("latin", "klarner"):
            raise ValueError(f"{init_mode} initialization assumes Q = N^2, got Q={Q}, N^2={N * N}.")  # mcmc.py:21-25
        if Q > N ** 3:
            raise ValueError(f"Q={Q} cannot exceed N^3={N ** 3}.")  # mcmc.py:94-95
        if Q < 2 or Q == N ** 3 or Q > 32767:
            raise ValueError(f"this build runs 2 <= Q < N^3 and Q <= 32767 queens, got Q={Q}")
        p.n_queens = Q
    return p


def beta_values(schedule_params, n_steps):
    """beta(step) for step = 0 .. n_steps - 1 exactly as the reference evaluates it (experiments.py:13-77): the same NumPy
    functions on float64 in the same order, vectorised (NumPy's scalar and array exp / log / cos agree bit for bit).  This is
    what the sweep is given as mcq_params.beta_table, so that beta is the reference's own value and not a second math
    library's opinion of it."""
    st, bc, bs, be = _check_schedule(schedule_params)
    n = int(n_steps)
    step = np.arange(n, dtype=np.float64)
    if st == "constant":
        return np.full(n, float(bc), dtype=np.float64)
    if n <= 1:
        return np.full(n, float(be), dtype=np.float64)
    if st == "linear_annealing":
        frac = step / (n - 1)
        return bs + frac * (be - bs)
    if st == "exponential_annealing":
        log_ratio = np.log(be / bs)
        t = np.clip(step, 0, n - 1) / (n - 1)
        return bs * np.exp(log_ratio * t)
    if st == "logarithmic_annealing":
        log_norm = np.log(1 + n)
        return bs + (be - bs) * (np.log(1 + np.clip(step, 0, n)) / log_norm)
    x = (np.pi * np.clip(step, 0, n)) / n
    return bs + ((be - bs) * (1 - np.cos(x))) / 2


def beta_table(params, schedule_sets=None, schedule_params=None):
    """The [n_sets][n_steps] float64 table of a Params block (beta_values per schedule)."""
    sps = schedule_sets if schedule_sets is not None else [schedule_params]
    return np.ascontiguousarray(np.stack([beta_values(sp, params.n_steps) for sp in sps]).reshape(len(sps), int(params.n_steps)))


def seeds_for(base_seed, n_chains):
    """Chain r is seeded with base_seed + r (experiments.py:508); NumPy's legacy seed() accepts
    only 0 <= seed <= 2**32 - 1 and raises ValueError otherwise."""
    s = np.arange(n_chains, dtype=np.int64) + int(base_seed)
    if n_chains and (s[0] < 0 or s[-1] > 2**32 - 1):
        raise ValueError("Seed must be between 0 and 2**32 - 1")
    return s.astype(np.uint32)


def copy_params(params):
    """A private copy of a Params block (the schedule-set array it may point to is kept alive with the copy)."""
    p = Params.from_buffer_copy(params)
    keep = getattr(params, "_sets_keepalive", None)
    if keep is not None:
        p._sets_keepalive = keep
    for k in ("_schedules", "_beta_keepalive", "_ladder_keepalive", "_stream_keepalive"):
        if hasattr(params, k):
            setattr(p, k, getattr(params, k))
    return p


def host_beta_table(params):
    """float64 [n_sets][n_steps] of a Params built by make_params / make_params_sets, or None when the schedules are not
    known on the Python side (`_schedules` absent or None: a hand-filled struct -- the device then evaluates them itself).
    The values are derived from the struct's own fields (sched / beta_* and sets[t]), the ones the library would read, so a
    Params edited after make_params runs the schedule it now describes, not the one it was built with."""
    if not getattr(params, "_schedules", None) or params.n_steps <= 0:
        return None
    names = {v: k for k, v in SCHED.items()}

    def as_dict(s):
        if s.sched not in names:
            raise ValueError(f"Unknown betta_scheduling type: {s.sched}")
        return {"type": names[s.sched], "beta_const": float(s.beta_const), "beta_start": float(s.beta_start), "beta_end": float(s.beta_end)}

    sch = [as_dict(params.sets[t]) for t in range(int(params.n_sets))] if params.n_sets > 1 else [as_dict(params)]
    return np.ascontiguousarray(np.stack([beta_values(sp, params.n_steps) for sp in sch]))


def set_stream_states(params, states):
    """Chains that continue MT19937 streams instead of seeding them (include/mcq.h: stream_states; metropolis_mcmc(seed=None),
    experiments.py:200-201, 287-288): `states` is one np.random.get_state() tuple -- ('MT19937', key[624], pos, ...) -- per chain, or an
    array [n_chains][625] of key words + position.  Returns params."""
    if isinstance(states, np.ndarray):
        arr = np.ascontiguousarray(states, dtype=np.uint32)
    else:
        if isinstance(states, tuple) and len(states) >= 3 and isinstance(states[0], str):
            states = [states]
        rows = []
        for st in states:
            if st[0] != "MT19937":
                raise ValueError("only MT19937 states can be continued")
            rows.append(np.concatenate([np.asarray(st[1], dtype=np.uint32), np.array([int(st[2])], dtype=np.uint32)]))
        arr = np.ascontiguousarray(np.stack(rows))
    if arr.shape != (params.n_chains, 625):
        raise ValueError("one MT19937 state (624 key words + position) per chain")
    if (arr[:, 624] > 624).any():
        raise ValueError("MT19937 position out of range")
    params.stream_states = arr.ctypes.data
    params._stream_keepalive = arr
    return params


def set_exchange(params, every, ladder):
    """Turn on replica exchange (include/mcq.h: exchange_every / exchange_replicas / exchange_ladder) on a Params block:
    every `every` steps neighbouring rungs of each ladder of len(ladder) consecutive chains are offered a swap of their beta
    multipliers.  NOT a mode of the reference.  Returns params."""
    lad = np.ascontiguousarray(ladder, dtype=np.float64)
    if int(every) <= 0:
        raise ValueError("exchange_every must be positive")
    if lad.ndim != 1 or len(lad) not in (2, 4, 8, 16):
        raise ValueError("the exchange ladder has 2, 4, 8 or 16 rungs")
    if params.n_chains % len(lad) or (params.n_sets > 1 and params.chains_per_set % len(lad)):
        raise ValueError("n_chains (and chains_per_set) must be multiples of the number of rungs")
    params.exchange_every, params.exchange_replicas = int(every), len(lad)
    params.exchange_ladder = lad.ctypes.data_as(C.POINTER(C.c_double))
    params._ladder_keepalive = lad
    return params

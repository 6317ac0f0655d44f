"""The ctypes stub printed in INTEGRATION.md section 2 (extracted verbatim by hand below the marker), run against the package.
Regenerate with: python - < tools/check_integration_stub.py after editing INTEGRATION.md."""
# --- verbatim from INTEGRATION.md (library path made absolute) ---
import ctypes as C, numpy as np

_L = C.CDLL(__import__("os").path.join(__import__("os").environ.get("GRAFT_REPO_ROOT", __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))),"monte-carlo-collective_amd/csrc/libmcq_hip.so"))
_L.mcq_run_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]   # pointers are 64-bit: never let ctypes default to int
_L.mcq_last_error.restype = C.c_char_p

class McqParams(C.Structure):      # include/mcq.h: mcq_params
    _fields_ = [("abi_version", C.c_int32), ("N", C.c_int32), ("mode", C.c_int32), ("init", C.c_int32),
                ("sched", C.c_int32), ("rng", C.c_int32), ("trace", C.c_int32), ("flags", C.c_uint32),
                ("beta_const", C.c_double), ("beta_start", C.c_double), ("beta_end", C.c_double),
                ("n_steps", C.c_int64), ("n_chains", C.c_int64), ("patience", C.c_int64),
                ("hist_stride", C.c_int64), ("bits_stride", C.c_int64),
                ("lanes_per_chain", C.c_int32), ("device", C.c_int32),
                ("n_sets", C.c_int64), ("chains_per_set", C.c_int64), ("sets", C.c_void_p)]   # several schedules in one launch; 0 / NULL here

class McqOutputs(C.Structure):     # include/mcq.h: mcq_outputs (all caller-allocated)
    _fields_ = [(n, C.c_void_p) for n in ("energy_hist", "accept_bits", "hist_len", "steps_executed",
                "initial_energy", "best_energy", "final_energy", "steps_to_best", "n_accepted",
                "near_ties", "best_state", "final_state",
                "step_sum", "step_sumsq", "step_accepted", "step_count")]   # last four: trace == REDUCED only

_SCHED = {"constant": 0, "linear_annealing": 1, "exponential_annealing": 2,
          "logarithmic_annealing": 3, "sinusoidal_annealing": 4}
_INIT = {"random": 0, "latin": 1, "klarner": 2}

def run_experiment(N, n_steps, init_mode, beta_schedule, n_runs, base_seed=0, verbose=False, n_workers=None,
                   schedule_params=None, mcmc_type="full_3d", early_stop_patience=100000):
    p = McqParams(abi_version=2, N=N, mode=0 if mcmc_type == "board" else 1, init=_INIT[init_mode],
                  sched=_SCHED[schedule_params["type"]], rng=0, trace=1, flags=0,
                  beta_const=schedule_params.get("beta_const") or 0.0,
                  beta_start=schedule_params.get("beta_start") or 0.0, beta_end=schedule_params.get("beta_end") or 0.0,
                  n_steps=n_steps, n_chains=n_runs,
                  patience=-1 if early_stop_patience in (None, "None", "null") else int(early_stop_patience),
                  hist_stride=(n_steps + 64) // 64 * 64, bits_stride=max(1, (n_steps + 63) // 64),
                  lanes_per_chain=0, device=-1, n_sets=0, chains_per_set=0, sets=None)
    seeds = (np.arange(n_runs, dtype=np.int64) + base_seed).astype(np.uint32)   # experiments.py:508
    hist = np.zeros((n_runs, p.hist_stride), np.int32); bits = np.zeros((n_runs, p.bits_stride), np.uint64)
    i64 = lambda: np.zeros(n_runs, np.int64); i32 = lambda: np.zeros(n_runs, np.int32)
    hl, ex, e0, be, fe, sb, na = i64(), i64(), i32(), i32(), i32(), i64(), i64()
    out = McqOutputs(hist.ctypes.data, bits.ctypes.data, hl.ctypes.data, ex.ctypes.data, e0.ctypes.data,
                     be.ctypes.data, fe.ctypes.data, sb.ctypes.data, na.ctypes.data, None, None, None, None, None, None, None)
    secs = C.c_double()
    rc = _L.mcq_run_host(C.addressof(p), seeds.ctypes.data, C.addressof(out), C.addressof(secs))
    if rc:   # -1 -> ValueError like the reference's own checks, others -> RuntimeError
        raise (ValueError if rc == -1 else RuntimeError)(_L.mcq_last_error().decode())
    acc = [np.flatnonzero(np.unpackbits(bits[r].view(np.uint8), bitorder="little")[: ex[r]]) for r in range(n_runs)]
    rej = [np.setdiff1d(np.arange(ex[r]), acc[r]) for r in range(n_runs)]
    return ([hist[r, : hl[r]] for r in range(n_runs)], be.tolist(), [secs.value / n_runs] * n_runs, acc, rej, sb.tolist())

if __name__ == "__main__":
    import sys, os
    sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import mcq_amd
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    a = run_experiment(12, 3000, "random", None, 24, base_seed=42, schedule_params=sp, mcmc_type="board", early_stop_patience=None)
    b = mcq_amd.run_experiment(12, 3000, "random", None, 24, base_seed=42, schedule_params=sp, mcmc_type="board", early_stop_patience=None)
    assert all((np.asarray(x) == np.asarray(y)).all() for x, y in zip(a[0], b[0])), "histories"
    assert list(a[1]) == list(b[1]) and list(a[5]) == list(b[5])
    assert all((np.asarray(x) == np.asarray(y)).all() for x, y in zip(a[3], b[3]))
    print("INTEGRATION.md stub == mcq_amd.run_experiment:", a[1][:6])

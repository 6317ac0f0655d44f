#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (build container only).

The reference (galgantar/monte-carlo-collective, mounted read-only at /root/reference)
is pure Python/NumPy and has no tests or golden vectors of its own, so the oracle and the
HIP kernels are pinned by vectors captured from the reference itself.  This script is the
only place the reference is imported; nothing of its source is written to the repo, only
inputs (parameters, seeds) and outputs (energies, accept bits, states).

Run:  MPLBACKEND=Agg python tools/gen_golden.py  [--reference /root/reference]
Takes ~3 minutes on 8 cores.
"""
import argparse
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

SCHEDULES = [
    {"type": "constant", "beta_const": 5.0},
    {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0},
    {"type": "exponential_annealing", "beta_start": 1.0, "beta_end": 3.0},
    {"type": "logarithmic_annealing", "beta_start": 0.5, "beta_end": 3.0},
    {"type": "sinusoidal_annealing", "beta_start": 0.1, "beta_end": 5.0},
]


def _ref(path):
    os.environ.setdefault("MPLBACKEND", "Agg")
    if path not in sys.path:
        sys.path.insert(0, path)
    import experiments  # noqa: the reference module

    return experiments


def _schedule(ex, sp, n_steps):
    return ex.build_schedule_from_params(
        sp["type"], n_steps, beta_const=sp.get("beta_const"), beta_start=sp.get("beta_start"), beta_end=sp.get("beta_end")
    )


def _state_bytes(state, mode):
    if mode == "board":
        return np.asarray(state.heights, dtype=np.uint8).reshape(-1)
    return np.asarray(state.queens, dtype=np.uint8).reshape(-1)


def run_chain(job):
    """One reference chain -> dict of plain arrays."""
    import contextlib
    import io

    ref_path, case = job
    ex = _ref(ref_path)
    sched = _schedule(ex, case["schedule"], case["n_steps"])
    fn = ex.metropolis_mcmc_board if case["mode"] == "board" else ex.metropolis_mcmc
    with contextlib.redirect_stdout(io.StringIO()):  # full_3d klarner fallback prints a warning
        res = fn(
            N=case["N"],
            n_steps=case["n_steps"],
            init_mode=case["init"],
            beta_schedule=sched,
            verbose=False,
            seed=case["seed"],
            early_stop_patience=case.get("patience"),
            **({"Q": case["Q"]} if "Q" in case else {}),  # metropolis_mcmc(..., Q=...): experiments.py:199-203
        )
    hist = np.asarray(res["energy_history"], dtype=np.int32)
    acc = np.zeros(case["n_steps"], dtype=np.uint8)
    acc[np.asarray(res["accepted_steps"], dtype=np.int64)] = 1
    rej = np.asarray(res["rejected_steps"], dtype=np.int64)
    assert not acc[rej].any()
    return {
        "hist": hist,
        "accept": np.packbits(acc, bitorder="little"),
        "n_executed": np.int64(len(res["accepted_steps"]) + len(res["rejected_steps"])),
        "best_energy": np.int32(res["best_energy"]),
        "final_energy": np.int32(res["final_energy"]),
        "steps_to_best": np.int64(res["steps_to_best"]),
        "best_state": _state_bytes(res["best_state"], case["mode"]),
        "final_state": _state_bytes(res["final_state"], case["mode"]),
    }


def gen_rng(out):
    """F1: draws of NumPy's legacy global RandomState exactly as the reference calls it."""
    arrays, meta = {}, []
    for seed in (0, 1, 42, 2**32 - 1):
        for N in (2, 3, 6, 12, 16, 17, 20, 24):
            np.random.seed(seed)
            arrays[f"randint_s{seed}_N{N}"] = np.array([np.random.randint(0, N) for _ in range(2000)], dtype=np.uint8)
        np.random.seed(seed)
        arrays[f"random_s{seed}"] = np.array([np.random.random() for _ in range(2000)], dtype=np.float64)
        for N in (3, 6, 12):
            np.random.seed(seed)
            arrays[f"grid_s{seed}_N{N}"] = np.random.randint(0, N, size=(N, N)).astype(np.uint8)
            np.random.seed(seed)
            arrays[f"choice_s{seed}_N{N}"] = np.random.choice(N**3, size=N * N, replace=False).astype(np.int32)
        # interleaved use, as one chain step does: bounded, bounded, bounded, double
        np.random.seed(seed)
        mix = []
        for _ in range(500):
            mix += [np.random.randint(0, 12), np.random.randint(0, 12), np.random.randint(0, 144)]
            mix.append(np.random.random())
        arrays[f"mixed_s{seed}"] = np.array(mix, dtype=np.float64)
        meta.append(seed)
    np.savez_compressed(os.path.join(out, "rng.npz"), **arrays)
    return {"seeds": meta, "numpy": np.__version__}


def gen_init(ref_path, out):
    """F2: initial states + E0 for every (mode, init, N, seed)."""
    import contextlib
    import io

    _ref(ref_path)
    from mcmc import State3DQueens
    from mcmc_board import State3DQueensBoard

    arrays, cases = {}, []
    for mode in ("board", "full_3d"):
        for init in ("random", "latin", "klarner"):
            for N in (2, 3, 6, 7, 11, 12, 13, 16, 24):
                for seed in (42, 43):
                    np.random.seed(seed)
                    with contextlib.redirect_stdout(io.StringIO()):
                        st = State3DQueensBoard(N, init_mode=init) if mode == "board" else State3DQueens(N, init_mode=init)
                    e0 = st.energy(recompute=True)
                    nxt = np.random.randint(0, 2**31 - 1)  # pins how many words the init consumed
                    key = f"{mode}_{init}_N{N}_s{seed}"
                    arrays[key] = _state_bytes(st, mode)
                    cases.append({"key": key, "mode": mode, "init": init, "N": N, "seed": seed, "E0": int(e0), "next_draw": int(nxt)})
    np.savez_compressed(os.path.join(out, "init.npz"), **arrays)
    return cases


def gen_analytic(ref_path):
    """F7: RNG-free known answers computed by the reference's energy()."""
    _ref(ref_path)
    from mcmc import State3DQueens
    from mcmc_board import State3DQueensBoard

    latin_b = {N: int(State3DQueensBoard(N, init_mode="latin").energy()) for N in range(2, 25)}
    latin_f = {N: int(State3DQueens(N, init_mode="latin").energy()) for N in range(2, 25)}
    klar = {N: int(State3DQueensBoard(N, init_mode="klarner").energy()) for N in (11, 13, 17, 19, 23)}
    return {"latin_board": latin_b, "latin_full_3d": latin_f, "klarner_exact_board": klar}


def chain_cases():
    cases = []
    for mode in ("board", "full_3d"):
        for init in ("random", "latin", "klarner"):
            for sp in SCHEDULES:
                for N in (3, 6, 12):
                    for seed in (42, 1042):
                        cases.append({"mode": mode, "init": init, "schedule": sp, "N": N, "seed": seed, "n_steps": 2000})
    # shapes of BASELINE configs 3 and 5, and odd sizes
    cases.append({"mode": "board", "init": "random", "schedule": SCHEDULES[4], "N": 24, "seed": 42, "n_steps": 1500})
    cases.append({"mode": "full_3d", "init": "random", "schedule": SCHEDULES[2], "N": 24, "seed": 42, "n_steps": 600})
    cases.append({"mode": "board", "init": "klarner", "schedule": SCHEDULES[1], "N": 17, "seed": 7, "n_steps": 1500})
    cases.append({"mode": "board", "init": "klarner", "schedule": SCHEDULES[1], "N": 20, "seed": 7, "n_steps": 1500})
    cases.append({"mode": "full_3d", "init": "klarner", "schedule": SCHEDULES[1], "N": 16, "seed": 7, "n_steps": 1000})
    cases.append({"mode": "board", "init": "random", "schedule": SCHEDULES[1], "N": 2, "seed": 5, "n_steps": 300})
    cases.append({"mode": "full_3d", "init": "random", "schedule": SCHEDULES[1], "N": 2, "seed": 5, "n_steps": 300})
    # degenerate schedule lengths (n_steps <= 1 -> beta_end)
    cases.append({"mode": "board", "init": "random", "schedule": SCHEDULES[1], "N": 6, "seed": 3, "n_steps": 1})
    cases.append({"mode": "board", "init": "random", "schedule": SCHEDULES[2], "N": 6, "seed": 3, "n_steps": 0})
    # F4 early stop (board only); full_3d ignores patience
    cases.append({"mode": "board", "init": "random", "schedule": SCHEDULES[0], "N": 6, "seed": 7, "n_steps": 10000, "patience": 300})
    cases.append({"mode": "board", "init": "random", "schedule": SCHEDULES[1], "N": 12, "seed": 11, "n_steps": 4000, "patience": 150})
    cases.append({"mode": "board", "init": "latin", "schedule": SCHEDULES[0], "N": 6, "seed": 9, "n_steps": 500, "patience": 0})
    cases.append({"mode": "full_3d", "init": "random", "schedule": SCHEDULES[0], "N": 6, "seed": 7, "n_steps": 1000, "patience": 50})
    # a longer run: N=12 board, the headline shape
    cases.append({"mode": "board", "init": "random", "schedule": SCHEDULES[1], "N": 12, "seed": 42, "n_steps": 20000})
    for idx, c in enumerate(cases):
        c["key"] = f"c{idx:03d}"
    return cases


def q_cases():
    """full_3d chains with Q != N^2 queens (State3DQueens(N, Q=...), mcmc.py:6-18; random init only): a separate file
    (chains_q.npz / manifest["chains_q"]) so that the other fixtures stay byte for byte what they were.
    `python tools/gen_golden.py --only q` writes just these."""
    cases = []
    for N, Q, n_steps in ((3, 2, 400), (3, 20, 400), (4, 5, 600), (6, 20, 800), (6, 100, 800), (12, 60, 1000), (12, 300, 600), (17, 100, 500), (5, 124, 300)):
        for seed in (42, 1042):
            for sp in (SCHEDULES[1], SCHEDULES[0]):
                cases.append({"mode": "full_3d", "init": "random", "schedule": sp, "N": N, "Q": Q, "seed": seed, "n_steps": n_steps})
    for idx, c in enumerate(cases):
        c["key"] = f"q{idx:03d}"
    return cases


def big_cases():
    """board chains beyond N = 32 (State3DQueensBoard is unbounded, mcmc_board.py:12; this build runs boards up to N = 128):
    chains_big.npz / manifest["chains_big"], `python tools/gen_golden.py --only big`."""
    cases = []
    for N, init, n_steps in ((33, "random", 300), (33, "klarner", 200), (40, "latin", 200), (47, "klarner", 150), (64, "random", 150), (100, "random", 60)):
        for seed in (42, 1042):
            cases.append({"mode": "board", "init": init, "schedule": SCHEDULES[1], "N": N, "seed": seed, "n_steps": n_steps})
    cases.append({"mode": "board", "init": "random", "schedule": SCHEDULES[0], "N": 36, "seed": 7, "n_steps": 400, "patience": 60})
    for idx, c in enumerate(cases):
        c["key"] = f"big{idx:03d}"
    return cases


def wide_cases():
    """full_3d chains beyond N = 32 (State3DQueens is unbounded, mcmc.py:6-18; this build runs full_3d up to N = 64 with 64-bit
    column words): chains_wide.npz / manifest["chains_wide"], `python tools/gen_golden.py --only wide`."""
    cases = []
    for N, init, n_steps in ((33, "random", 300), (33, "klarner", 200), (33, "latin", 200), (41, "random", 200), (48, "random", 200), (48, "klarner", 150),
                             (64, "random", 120), (64, "latin", 120)):
        for seed in (42, 1042):
            cases.append({"mode": "full_3d", "init": init, "schedule": SCHEDULES[1], "N": N, "seed": seed, "n_steps": n_steps})
    cases.append({"mode": "full_3d", "init": "random", "schedule": SCHEDULES[0], "N": 40, "Q": 3000, "seed": 7, "n_steps": 300})
    for idx, c in enumerate(cases):
        c["key"] = f"wide{idx:03d}"
    return cases


def stream_cases():
    """Chains that continue NumPy's global stream instead of seeding it (metropolis_mcmc[_board](..., seed=None), experiments.py:200-201,
    287-288): the state the chain starts from (key words + position, every position class: 0, inside a generation, 624), the chain's
    results, and the four 32-bit words the global stream yields AFTER the chain.  chains_stream.npz / manifest["chains_stream"],
    `python tools/gen_golden.py --only stream`."""
    cases = []
    shapes = [("board", "random", 6, 300), ("board", "latin", 12, 400), ("board", "klarner", 9, 300), ("full_3d", "random", 6, 300),
              ("full_3d", "latin", 12, 300), ("full_3d", "klarner", 10, 200), ("board", "random", 33, 120), ("full_3d", "random", 33, 100)]
    positions = [0, 1, 15, 16, 63, 64, 65, 300, 576, 608, 609, 623, 624]
    for idx, (mode, init, N, n_steps) in enumerate(shapes):
        for pos in (positions[idx::3] if idx else positions):
            cases.append({"mode": mode, "init": init, "schedule": SCHEDULES[1], "N": N, "seed": 1000 + idx, "n_steps": n_steps, "stream_pos": pos,
                          "warm_words": 624 * (idx % 3) + 7})
    for idx, c in enumerate(cases):
        c["key"] = f"stream{idx:03d}"
    return cases


def run_stream_chain(job):
    """np.random.seed(case seed); draw warm_words words; force the position; run the chain with seed=None."""
    ref_path, case = job
    np.random.seed(case["seed"])
    np.random.randint(0, 2**32, size=case["warm_words"], dtype=np.uint32)
    st = np.random.get_state()
    key = np.array(st[1], dtype=np.uint32)
    np.random.set_state(("MT19937", key, case["stream_pos"]))
    res = run_chain((ref_path, dict(case, seed=None)))
    res["state"] = np.concatenate([key, np.array([case["stream_pos"]], dtype=np.uint32)])
    res["after"] = np.random.randint(0, 2**32, size=4, dtype=np.uint32)
    return res


def gen_beta(ref_path, out):
    """F5: float64 beta(step) tables of the five schedule closures."""
    ex = _ref(ref_path)
    arrays, cases = {}, []
    rng = np.random.RandomState(12345)
    for si, sp in enumerate(SCHEDULES):
        for n in (1, 2, 1000, 1000000):
            sched = _schedule(ex, sp, n)
            steps = np.arange(n, dtype=np.int64) if n <= 1000 else np.unique(np.concatenate([[0, 1, n - 2, n - 1], rng.randint(0, n, 4000)]))
            vals = np.array([float(sched(int(s))) for s in steps], dtype=np.float64)
            key = f"b{si}_n{n}"
            arrays[key + "_steps"] = steps
            arrays[key + "_beta"] = vals
            cases.append({"key": key, "schedule": sp, "n_steps": n})
    np.savez_compressed(os.path.join(out, "beta.npz"), **arrays)
    return cases


def gen_plumbing(ref_path):
    """F6: BASELINE config 1 through the reference's run_experiment (process pool)."""
    ex = _ref(ref_path)
    sp = {"type": "constant", "beta_const": 5.0}
    hist, best, _t, acc, rej, stb = ex.run_experiment(
        N=6, n_steps=10000, init_mode="random", beta_schedule=_schedule(ex, sp, 10000), n_runs=4, base_seed=42,
        verbose=False, n_workers=4, schedule_params=sp, mcmc_type="board", early_stop_patience=None,
    )
    return {
        "N": 6, "n_steps": 10000, "init": "random", "schedule": sp, "n_runs": 4, "base_seed": 42, "mode": "board",
        "E0": [int(h[0]) for h in hist], "best": [int(b) for b in best], "steps_to_best": [int(s) for s in stb],
        "n_accepted": [len(a) for a in acc], "final": [int(h[-1]) for h in hist],
        "hist_crc": [int(np.bitwise_xor.reduce((np.asarray(h, dtype=np.int64) * (np.arange(len(h)) + 1)) & 0x7FFFFFFF)) for h in hist],
    }


def gen_drivers(ref_path):
    """Small runs of the reference's experiment drivers (plot=False): seed derivations, labels, result dicts."""
    import contextlib
    import io

    ex = _ref(ref_path)
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        r = ex.run_beta_start_end_pairs(N=6, n_steps=400, beta_start_ends=[[0.5, 3.0], [1.0, 5.0], [0.1, 2.0]],
                                        annealing_type="linear_annealing", init_mode="random", n_runs=3, base_seed=42,
                                        verbose=False, plot=False, mcmc_type="board", early_stop_patience=None)
        out["pairs"] = {"args": {"N": 6, "n_steps": 400, "beta_start_ends": [[0.5, 3.0], [1.0, 5.0], [0.1, 2.0]],
                                 "annealing_type": "linear_annealing", "init_mode": "random", "n_runs": 3, "base_seed": 42,
                                 "mcmc_type": "board", "early_stop_patience": None},
                        "best": {k: [int(b) for b in v] for k, v in r["all_best_energies"].items()},
                        "final": {k: [int(h[-1]) for h in v] for k, v in r["all_histories"].items()},
                        "hist_sum": {k: [int(np.sum(h)) for h in v] for k, v in r["all_histories"].items()}}
        sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
        m = ex.measure_min_energy_vs_N(Ns=[3, 4, 6], n_steps=300, beta_schedule=_schedule(ex, sp, 300), schedule_params=sp,
                                       init_modes=["random", "latin", "klarner"], n_runs=3, base_seed=42, verbose=False,
                                       plot=False, mcmc_type="board", early_stop_patience=100)
        out["min_vs_N"] = {"args": {"Ns": [3, 4, 6], "n_steps": 300, "schedule_params": sp, "init_modes": ["random", "latin", "klarner"],
                                    "n_runs": 3, "base_seed": 42, "mcmc_type": "board", "early_stop_patience": 100},
                           "results": {im: {"all_min": [[int(x) for x in a] for a in v["all_min_energies"]],
                                            "all_stb": [[int(x) for x in a] for a in v["all_steps_to_best"]],
                                            "mean_min": [float(x) for x in v["mean_min_energies"]],
                                            "std_min": [float(x) for x in v["std_min_energies"]],
                                            "mean_stb": [float(x) for x in v["mean_steps_to_best"]],
                                            "std_stb": [float(x) for x in v["std_steps_to_best"]]}
                                       for im, v in m["results"].items()}}
        c = ex.run_compare_beta_end(Ns=[4, 5], n_steps=300, beta_start_ends=[[1.0, 3.0], [1.0, 5.0]],
                                    annealing_type="exponential_annealing", init_mode="latin", n_runs=2, base_seed=7,
                                    verbose=False, plot=False, mcmc_type="full_3d", early_stop_patience=None)
        out["compare"] = {"args": {"Ns": [4, 5], "n_steps": 300, "beta_start_ends": [[1.0, 3.0], [1.0, 5.0]],
                                   "annealing_type": "exponential_annealing", "init_mode": "latin", "n_runs": 2, "base_seed": 7,
                                   "mcmc_type": "full_3d", "early_stop_patience": None},
                          "N1": c["N1"], "N2": c["N2"],
                          "best_N1": {k: [int(b) for b in v] for k, v in c["result_N1"]["all_best_energies"].items()},
                          "best_N2": {k: [int(b) for b in v] for k, v in c["result_N2"]["all_best_energies"].items()}}
        # acceptance binning of the reference's plot helper, re-derived from its inputs
        sp2 = {"type": "constant", "beta_const": 1.0}
        h, b, _t, acc, rej, _s = ex.run_experiment(N=5, n_steps=1000, init_mode="random", beta_schedule=_schedule(ex, sp2, 1000),
                                                  n_runs=2, base_seed=3, verbose=False, n_workers=2, schedule_params=sp2,
                                                  mcmc_type="board", early_stop_patience=None)
        edges = np.linspace(0, 1000, 101)
        edges[-1] = 1000
        a_all, r_all = np.concatenate(acc), np.concatenate(rej)
        rates = []
        for i in range(100):
            if i == 99:
                a = np.sum((a_all >= edges[i]) & (a_all <= edges[i + 1])); rr = np.sum((r_all >= edges[i]) & (r_all <= edges[i + 1]))
            else:
                a = np.sum((a_all >= edges[i]) & (a_all < edges[i + 1])); rr = np.sum((r_all >= edges[i]) & (r_all < edges[i + 1]))
            rates.append(float(a / (a + rr)) if a + rr else None)
        out["acceptance"] = {"args": {"N": 5, "n_steps": 1000, "schedule_params": sp2, "n_runs": 2, "base_seed": 3}, "rates": rates}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--only", default="", help="'q': only the Q != N^2 chains (chains_q.npz); 'big': only the boards beyond N = 32 (chains_big.npz); 'wide': only the full_3d chains beyond N = 32 (chains_wide.npz); 'stream': only the seed=None chains (chains_stream.npz); merged into the existing manifest")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    if args.only in ("q", "big", "wide", "stream"):
        name = "chains_" + args.only
        cases = {"q": q_cases, "big": big_cases, "wide": wide_cases, "stream": stream_cases}[args.only]()
        with ProcessPoolExecutor(max_workers=args.workers) as pool:
            results = list(pool.map(run_stream_chain if args.only == "stream" else run_chain, [(args.reference, c) for c in cases], chunksize=2))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **{f"{c['key']}_{k}": v for c, r in zip(cases, results) for k, v in r.items()})
        with open(os.path.join(OUT, "manifest.json")) as f:
            manifest = json.load(f)
        manifest[name] = cases
        with open(os.path.join(OUT, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        print(f"wrote {len(cases)} chains to {name}.npz")
        return
    manifest = {"generator": "tools/gen_golden.py", "reference": "galgantar/monte-carlo-collective @ 2026-01-09"}

    manifest["rng"] = gen_rng(OUT)
    manifest["init"] = gen_init(args.reference, OUT)
    manifest["analytic"] = gen_analytic(args.reference)
    manifest["beta"] = gen_beta(args.reference, OUT)
    manifest["plumbing"] = gen_plumbing(args.reference)
    manifest["drivers"] = gen_drivers(args.reference)

    cases = chain_cases()
    with ProcessPoolExecutor(max_workers=args.workers) as pool:
        results = list(pool.map(run_chain, [(args.reference, c) for c in cases], chunksize=4))
    arrays = {}
    for c, r in zip(cases, results):
        for name, val in r.items():
            arrays[f"{c['key']}_{name}"] = val
    np.savez_compressed(os.path.join(OUT, "chains.npz"), **arrays)
    manifest["chains"] = cases

    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    total = sum(os.path.getsize(os.path.join(OUT, n)) for n in os.listdir(OUT))
    print(f"wrote {len(cases)} chains, {len(manifest['init'])} init states; {total / 1e6:.2f} MB in {OUT}")


if __name__ == "__main__":
    main()

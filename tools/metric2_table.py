#!/usr/bin/env python3
"""Metric 2 (min-energy-reached) at the shape the reference's report publishes (report p.4 section IV-C, figures/energy_history_N3to15*.png):
measure_min_energy_vs_N with Ns = 3..15, init modes random / latin / klarner, linear annealing 1 -> 3, 20 runs x 5 000 000 steps,
base_seed 42, board chains, no early stop (config.yaml: early_stop_patience None), no trace (experiments.py:1061 discards histories).
Run on the GPU box:   python tools/metric2_table.py gpurun_out/r04_metric2.json [--n-steps 5000000] [--n-runs 20]
Writes the per-cell min / mean / std of the best energy and the mean / std of steps-to-best, the wall time and moves/s."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--n-steps", type=int, default=5_000_000)
    ap.add_argument("--n-runs", type=int, default=20)
    ap.add_argument("--n-max", type=int, default=15)
    ap.add_argument("--patience", type=int, default=None, help="early_stop_patience (default None = disabled, config.yaml:9; the reference's function default is 100000)")
    a = ap.parse_args()
    import mcq_amd
    import numpy as np

    Ns = list(range(3, a.n_max + 1))
    modes = ["random", "latin", "klarner"]
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    sched = mcq_amd.build_schedule_from_params("linear_annealing", a.n_steps, beta_start=1.0, beta_end=3.0)
    t0 = time.perf_counter()
    r = mcq_amd.measure_min_energy_vs_N(Ns, a.n_steps, sched, schedule_params=sp, init_modes=modes, n_runs=a.n_runs, base_seed=42, verbose=False,
                                        plot=False, mcmc_type="board", early_stop_patience=a.patience)
    wall = time.perf_counter() - t0
    cells = {}
    for m in modes:
        res = r["results"][m]
        for k, N in enumerate(Ns):
            best = np.asarray(res["all_min_energies"][k])
            stb = np.asarray(res["all_steps_to_best"][k])
            cells[f"{m}_N{N}"] = {"min": int(best.min()), "mean": float(best.mean()), "std": float(best.std()), "runs_at_zero": int((best == 0).sum()),
                                  "mean_steps_to_best": float(stb.mean()), "std_steps_to_best": float(stb.std()), "best": best.tolist()}
    moves = len(Ns) * len(modes) * a.n_runs * a.n_steps
    out = {"workload": f"measure_min_energy_vs_N Ns=3..{a.n_max} x {modes} board linear 1->3, {a.n_runs} runs x {a.n_steps} steps, base_seed 42, early_stop_patience {a.patience}",
           "wall_seconds": wall, "moves": moves, "moves_per_second": moves / wall, "cells": cells}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(f"{moves} moves in {wall:.2f} s = {moves / wall:.3e} moves/s (one-shot driver call: allocation, {len(Ns) * len(modes)} cells as {len(Ns)} launches, reduce)")
    print("N    " + "".join(f"{m:>28s}" for m in modes) + "      (min | mean +- std | mean steps to best)")
    for N in Ns:
        print(f"{N:<5d}" + "".join(f"{cells[f'{m}_N{N}']['min']:>6d} |{cells[f'{m}_N{N}']['mean']:>7.1f} +-{cells[f'{m}_N{N}']['std']:>5.1f} |{cells[f'{m}_N{N}']['mean_steps_to_best']:>9.0f}" for m in modes))


if __name__ == "__main__":
    main()

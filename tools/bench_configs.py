#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on ONE MI355X (configs 3, 4, 5; bench.py measures config 1/2's headline).

Each config is a set of launches (one per beta pair / per (init, N) cell) enqueued on separate HIP streams so that
they overlap on the GPU; moves/s = sum of executed steps / wall time from first enqueue to last completion.
Prints one JSON line per config.  Usage: python tools/bench_configs.py [--n-steps 100000] [--chains 8192]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_set(name, jobs, trace):
    import torch

    import mcq_amd

    abi = mcq_amd.abi
    runs = []
    for j in jobs:
        p = abi.make_params(j["N"], j["n_steps"], j["init"], j["sp"], j["n_chains"], mcmc_type=j["mode"], early_stop_patience=None, trace=trace)
        runs.append(mcq_amd._lib.DeviceRun(p, abi.seeds_for(j["seed"], j["n_chains"]), trace=trace, states=False))
    streams = [torch.cuda.Stream() for _ in runs]
    for r, st in zip(runs, streams):  # warm-up (first-touch of the buffers, code load)
        r.launch(stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    order = sorted(range(len(runs)), key=lambda i: -(runs[i].p.N * runs[i].p.n_steps * runs[i].p.n_chains))  # longest first, as _lib.run_many
    for i in order:
        runs[i].launch(stream=streams[i])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    moves = sum(int(r.t["steps_executed"].sum().item()) for r in runs)
    best = [int(r.t["best_energy"].min().item()) for r in runs]
    line = {"config": name, "launches": len(runs), "chains": sum(j["n_chains"] for j in jobs), "moves": moves, "seconds": dt,
            "moves_per_s": moves / dt, "min_energy_per_launch": best, "trace": trace if isinstance(trace, str) else ("i32" if trace else "none")}
    if trace == "reduced":  # the statistics the reference plots, straight from the accumulators
        t = runs[0].t
        n = t["step_count"].double()
        mean = t["step_sum"].double() / n
        line["mean_energy_at"] = {str(e): float(mean[e]) for e in (0, len(mean) // 4, len(mean) // 2, len(mean) - 1)}
        line["accepted_total"] = int(t["step_accepted"].sum().item())
    print(json.dumps(line), flush=True)
    del runs
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-steps", type=int, default=100000)
    ap.add_argument("--chains", type=int, default=8192)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    n = a.n_steps
    if a.only in ("c2r",):
        run_set("C2 single_N N=12 board linear 1->3, 65536 chains, 10^6 steps, trace=reduced",
                [dict(N=12, n_steps=1000000, init="random", sp={"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0},
                      n_chains=65536, mode="board", seed=42)], trace="reduced")
    if a.only in ("", "c3"):
        run_set("C3 single_N N=12 full_3d exponential 1->3, 65536 chains",
                [dict(N=12, n_steps=n, init="random", sp={"type": "exponential_annealing", "beta_start": 1.0, "beta_end": 3.0},
                      n_chains=65536, mode="full_3d", seed=42)], trace=True)
    if a.only in ("", "c5"):
        pairs = [(s, e) for s in (0.1, 0.5, 1.0, 2.0) for e in (2.0, 3.0, 5.0, 8.0)]
        run_set(f"C5 beta_start_end_pairs N=24 board sinusoidal, 16 pairs x {a.chains} chains",
                [dict(N=24, n_steps=n, init="random", sp={"type": "sinusoidal_annealing", "beta_start": s, "beta_end": e},
                      n_chains=a.chains, mode="board", seed=42 + 1000 * i) for i, (s, e) in enumerate(pairs)], trace=True)
    if a.only in ("", "c5", "c5b"):
        # the same 16 pairs as ONE launch with 16 schedule sets (mcq_params.sets)
        import numpy as np
        import torch

        import mcq_amd

        abi = mcq_amd.abi
        pairs = [(s, e) for s in (0.1, 0.5, 1.0, 2.0) for e in (2.0, 3.0, 5.0, 8.0)]
        sets = [{"type": "sinusoidal_annealing", "beta_start": s, "beta_end": e} for s, e in pairs]
        p = abi.make_params_sets(24, n, "random", sets, a.chains, mcmc_type="board", trace=True)
        seeds = np.concatenate([abi.seeds_for(42 + 1000 * i, a.chains) for i in range(len(pairs))])
        run = mcq_amd._lib.DeviceRun(p, seeds, trace=True, states=False)
        run.launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run.launch()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        moves = int(run.t["steps_executed"].sum().item())
        best = run.t["best_energy"].reshape(len(pairs), a.chains).min(dim=1).values.tolist()
        print(json.dumps({"config": f"C5 batched: the 16 pairs x {a.chains} chains as one launch with 16 schedule sets", "launches": 1,
                          "chains": len(seeds), "moves": moves, "seconds": dt, "moves_per_s": moves / dt, "min_energy_per_launch": best,
                          "trace": "i32"}), flush=True)
        del run
        torch.cuda.empty_cache()
    if a.only in ("", "c4"):
        jobs = []
        for init in ("random", "latin", "klarner"):
            off = sum(ord(c) for c in init) % 1000
            for idx, N in enumerate(range(3, 21)):
                jobs.append(dict(N=N, n_steps=n, init=init, sp={"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0},
                                 n_chains=a.chains, mode="board", seed=42 + 10 * idx + off))
        run_set(f"C4 measure_min_energy_vs_N Ns=3..20 x 3 inits, {a.chains} chains per cell", jobs, trace=False)


if __name__ == "__main__":
    main()

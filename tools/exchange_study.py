#!/usr/bin/env python3
"""Replica exchange (mcq_params.exchange_every; NOT a mode of the reference) against plain annealing at equal moves, judged the way a
ladder is meant to be judged: by the best energy any of its 16 chains reaches.  N = 12 board, 65 536 chains x 100 000 steps, seeds 42 + r.
For every variant: global minimum, mean over all chains of the best energy, mean over the 4 096 groups of 16 consecutive chains of the
group's minimum (for the plain runs the same groups of 16 independent chains), accepted swaps per chain, sweep ms.
usage (GPU box): python tools/exchange_study.py > gpurun_out/r03_exchange_study.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")


def main():
    import numpy as np
    import torch

    import mcq_amd

    abi = mcq_amd.abi
    n, steps, R = 65536, 100000, 16
    seeds = abi.seeds_for(42, n)
    st = torch.cuda.current_stream()
    rows = []

    def run(name, sp, every=0, lo=1.0, hi=1.0):
        p = abi.make_params(12, steps, "random", sp, n, mcmc_type="board", early_stop_patience=None, trace=False)
        if every:
            abi.set_exchange(p, every, lo * (hi / lo) ** (np.arange(R) / (R - 1)))
        r = mcq_amd._lib.DeviceRun(p, seeds, trace=False, states=False)
        r.launch(st)
        ms = min(r.launch_timed(st)[1] for _ in range(2))
        best = r.t["best_energy"].cpu().numpy().astype(np.int64)
        row = {"variant": name, "schedule": sp, "exchange_every": every, "ladder": [lo, hi] if every else None, "sweep_ms": ms,
               "moves_per_s": n * steps / (ms * 1e-3), "min_energy": int(best.min()), "mean_best_energy": float(best.mean()),
               "mean_of_group_minima": float(best.reshape(-1, R).min(axis=1).mean()),
               "groups_at_or_below_30": int((best.reshape(-1, R).min(axis=1) <= 30).sum()),
               "accepted_swaps_per_chain": float(r.t["n_exchanges"].double().mean().item()) if every else None}
        rows.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
        del r
        torch.cuda.empty_cache()

    lin = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    run("plain linear annealing 1 -> 3 (BASELINE configs[1])", lin)
    run("plain constant beta = 3", {"type": "constant", "beta_const": 3.0})
    run("plain constant beta = 2", {"type": "constant", "beta_const": 2.0})
    for every in (16, 64, 1000):
        run("parallel tempering: constant beta = 1, rungs x 1 ... 3", {"type": "constant", "beta_const": 1.0}, every, 1.0, 3.0)
    for every in (64,):
        run("parallel tempering: constant beta = 1, rungs x 1.5 ... 4", {"type": "constant", "beta_const": 1.0}, every, 1.5, 4.0)
        run("parallel tempering: constant beta = 1, rungs x 2 ... 3.5", {"type": "constant", "beta_const": 1.0}, every, 2.0, 3.5)
        run("annealed ladder: linear 1 -> 3, rungs x 0.7 ... 1.4", lin, every, 0.7, 1.4)
        run("annealed ladder: linear 1 -> 3, rungs x 1 ... 1.5", lin, every, 1.0, 1.5)
    json.dump({"what": __doc__.split("usage")[0].strip(), "rows": rows}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

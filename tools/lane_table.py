#!/usr/bin/env python3
"""Sweep time of one board launch per (N, lanes per chain): what the occupancy-aware lane choice is tuned on.
usage (GPU box): python tools/lane_table.py [--chains 3072] [--n-steps 20000] [--Ns 3-24] [--trace none|reduced] [--mode board] [--json OUT]

--json writes the table jobs.plan_lanes reads (monte-carlo-collective_amd/lane_table.json is a copy of it): the milliseconds per
(N, lanes) together with the sha256 of the kernel source they were measured on; tests/test_host_logic.py fails when
csrc/mcq_hip.hip changes without a new table."""
import argparse
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=3072)
    ap.add_argument("--n-steps", type=int, default=20000)
    ap.add_argument("--Ns", default="3-24")
    ap.add_argument("--trace", default="none")
    ap.add_argument("--mode", default="board")
    ap.add_argument("--lanes", default="4,8,16")
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    table = {}
    import torch

    import mcq_amd

    lo, hi = (int(x) for x in args.Ns.split("-"))
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    trace = {"none": False, "reduced": "reduced", "i32": True}[args.trace]
    st = torch.cuda.current_stream()
    for N in range(lo, hi + 1):
        row = []
        for G in (int(g) for g in args.lanes.split(",")):
            p = mcq_amd.abi.make_params(N, args.n_steps, "random", sp, args.chains, mcmc_type=args.mode, trace=trace, lanes_per_chain=G)
            run = mcq_amd._lib.DeviceRun(p, mcq_amd.abi.seeds_for(42, args.chains), trace=trace, states=False)
            run.launch(st)
            best = min(run.launch_timed(st)[1] for _ in range(2))
            row.append(f"G={G}: {best:8.3f} ms")
            table.setdefault(str(N), []).append(round(best, 3))
            del run
        print(f"N={N:2d} chains={args.chains} steps={args.n_steps}  " + "   ".join(row), flush=True)
    if args.json:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        with open(os.path.join(root, "monte-carlo-collective_amd", "csrc", "mcq_hip.hip"), "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()
        with open(args.json, "w") as f:
            json.dump({"kernel_sha256": sha, "what": f"sweep ms of one {args.mode} launch of {args.chains} chains x {args.n_steps} steps, trace {args.trace}, "
                                                    "per lanes per chain: a wavefront with the SIMD (nearly) to itself",
                       "chains": args.chains, "n_steps": args.n_steps, "lanes": [int(g) for g in args.lanes.split(",")], "ms": table}, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()

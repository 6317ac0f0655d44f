#!/usr/bin/env python3
"""full_3d beyond N = 32 (64-bit column words, the queens and the init kernel's N^3 permutation in global memory): init and sweep time of
one launch per N, with N = 24 / 32 (32-bit column words, 16 lanes) beside them.
usage (GPU box): python tools/wide_timing.py [--chains 1024] [--n-steps 20000] > profiles/rNN_full3d_wide.txt"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=1024)
    ap.add_argument("--n-steps", type=int, default=20000)
    args = ap.parse_args()
    import torch

    import mcq_amd

    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    st = torch.cuda.current_stream()
    print(f"# full_3d, {args.chains} chains x {args.n_steps} steps, trace none, 16 lanes per chain (4 chains per wavefront): init kernel | sweep | moves/s | workspace")
    for N, init in ((24, "random"), (32, "random"), (33, "random"), (33, "latin"), (40, "random"), (48, "random"), (48, "klarner"), (64, "random"), (64, "latin")):
        p = mcq_amd.abi.make_params(N, args.n_steps, init, sp, args.chains, mcmc_type="full_3d", trace=False, lanes_per_chain=16)
        run = mcq_amd._lib.DeviceRun(p, mcq_amd.abi.seeds_for(42, args.chains), trace=False, states=False)
        run.launch(st)
        t = min((run.launch_timed(st) for _ in range(2)), key=lambda x: x[1])
        ws = run.ws_bytes
        print(f"N={N:2d} {init:8s} init {t[0]:9.3f} ms   sweep {t[1]:9.3f} ms   {args.chains * args.n_steps / t[1] * 1e3:9.3e} moves/s   workspace {ws / 2**20:8.1f} MiB", flush=True)
        del run


if __name__ == "__main__":
    main()

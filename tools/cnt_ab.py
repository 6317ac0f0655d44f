#!/usr/bin/env python3
"""Round 4, review item 7: dE from LDS line counters (MCQ_FLAG_LINE_COUNTERS) against the bit-mask probes on the small boards of
BASELINE configs[3] (N = 3..8), in the two regimes that config meets: a launch far below the device's capacity (3 072 chains: the
per-GPU shape of an 8-GPU node, wavefronts with a SIMD to themselves) and a full device (65 536 chains).
usage (GPU box): python tools/cnt_ab.py > profiles/r04_line_counters.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import mcq_amd

    abi = mcq_amd.abi
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    st = torch.cuda.current_stream()
    print("# sweep ms of one launch, board, no trace, 4 lanes per chain, 20 000 steps: probes | line counters (LDS bytes per wavefront)")
    for chains in (3072, 65536):
        for N in range(3, 9):
            row = []
            for flags in (0, abi.FLAG_LINE_COUNTERS):
                p = abi.make_params(N, 20000, "random", sp, chains, mcmc_type="board", trace=False, lanes_per_chain=4, flags=flags)
                run = mcq_amd._lib.DeviceRun(p, abi.seeds_for(42, chains), trace=False, states=False)
                run.launch(st)
                row.append(min(run.launch_timed(st)[1] for _ in range(3)))
                del run
            T = 2 * N * N + 6 * N * (2 * N - 1) + 4 * (2 * N - 1) ** 2
            print(f"chains={chains:6d} N={N}  probes {row[0]:8.3f} ms   counters {row[1]:8.3f} ms   ({(row[0] / row[1] - 1) * 100:+5.1f} % moves/s)   counters: {T} B per chain", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A COLD one-shot driver call next to the steady state bench.py measures (advisor, round 3): run_beta_start_end_pairs-shaped job list
(N = 24 board, sinusoidal, 16 pairs x 1 024 chains, 100 000 steps, on-device statistics) as JobSet(...).run() -- allocation of the device
buffers and of ONE page-locked host buffer, launch, reduce -- and then the same JobSet again (steady state: the second page-locked buffer
is allocated by the second reduce)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import mcq_amd

    jb = mcq_amd.jobs
    pairs = [(s, e) for s in (0.1, 0.5, 1.0, 2.0) for e in (2.0, 3.0, 5.0, 8.0)]
    jobs = [jb.make_job(24, 100000, "random", {"type": "sinusoidal_annealing", "beta_start": s, "beta_end": e}, 1024, 42 + 1000 * i, "board", None)
            for i, (s, e) in enumerate(pairs)]
    torch.cuda.init()
    torch.zeros(1, device="cuda")  # the HIP runtime and the library are up: a driver process has them anyway
    mcq_amd._lib.lib()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    js = jb.JobSet(jobs, want="stats")
    t1 = time.perf_counter()
    res = js.run()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    moves = sum(r["summary"]["proposed"] for r in res)
    print(f"cold one-shot: allocate {1e3 * (t1 - t0):.1f} ms + launch and reduce {1e3 * (t2 - t1):.1f} ms = {1e3 * (t2 - t0):.1f} ms -> {moves / (t2 - t0):.3e} moves/s")
    for k in range(3):
        t3 = time.perf_counter()
        js.run()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print(f"same JobSet again ({k + 2}. run): {1e3 * (t4 - t3):.1f} ms -> {moves / (t4 - t3):.3e} moves/s")


if __name__ == "__main__":
    main()

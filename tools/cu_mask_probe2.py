#!/usr/bin/env python3
"""How many compute units a CU mask of a HIP stream really enables on this device: one sweep launch of 16 384 chains (1 024 wavefronts: time ~ 1 / CUs once they are
oversubscribed) on streams whose masks are prefixes / strided subsets of the 256 bits."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import mcq_amd

    abi, _lib = mcq_amd.abi, mcq_amd._lib
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    run = _lib.DeviceRun(abi.make_params(12, 20000, "random", sp, 16384, mcmc_type="board", trace=False, lanes_per_chain=4), abi.seeds_for(42, 16384), trace=False, states=False)

    def t(st):
        run.launch(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run.launch(st)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0)

    print("plain stream: %.2f ms" % t(torch.cuda.Stream()))
    for name, ids in [(f"first {n} bits", range(n)) for n in (8, 16, 32, 48, 64, 96, 128, 192, 256)] + \
                     [("even bits of the first 64", range(0, 64, 2)), ("odd bits of the first 64", range(1, 64, 2)), ("bits 64..127", range(64, 128)),
                      ("bits 0..7 + 64..71", list(range(8)) + list(range(64, 72))), ("every 4th bit (64)", range(0, 256, 4)), ("every 2nd bit (128)", range(0, 256, 2)),
                      ("bits 8k..8k+3 for all k (128)", [8 * k + j for k in range(32) for j in range(4)])]:
        h = _lib.cu_masked_stream(list(ids), 256)
        print(f"{name:34s}: %.2f ms" % t(torch.cuda.ExternalStream(h)))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""What a CU mask on a HIP stream does on this device: one small sweep launch (N = 12 board, 3 072 chains x 20 000 steps) on streams with different masks,
alone and two at a time."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    import mcq_amd

    abi, _lib = mcq_amd.abi, mcq_amd._lib
    sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    hip = _lib.hip_runtime()

    def run_on(streams, chains=3072):
        runs = [_lib.DeviceRun(abi.make_params(12, 20000, "random", sp, chains, mcmc_type="board", trace=False, lanes_per_chain=4), abi.seeds_for(42 + 7 * i, chains), trace=False, states=False)
                for i in range(len(streams))]
        for r, st in zip(runs, streams):
            r.launch(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r, st in zip(runs, streams):
            r.launch(st)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0)

    def masked(ids):
        h = _lib.cu_masked_stream(ids, n_cus)
        words = (n_cus + 31) // 32
        back = (C.c_uint32 * words)()
        rc = hip.hipExtStreamGetCUMask(C.c_void_p(h), words, back)
        return torch.cuda.ExternalStream(h), [hex(x) for x in back], rc

    print("CUs", n_cus)
    print("plain stream, one launch: %.2f ms" % run_on([torch.cuda.Stream()]))
    print("two plain streams: %.2f ms" % run_on([torch.cuda.Stream(), torch.cuda.Stream()]))
    for name, ids in (("bits 0..31", range(0, 32)), ("bits 0..63", range(0, 64)), ("every 8th bit (32 CUs)", range(0, 256, 8)), ("bits 0..255", range(0, 256)),
                      ("bits 0..127", range(0, 128)), ("even bits (128)", range(0, 256, 2))):
        st, back, rc = masked(list(ids))
        print(f"masked {name}: one launch %.2f ms   (mask read back rc={rc}: {back})" % run_on([st]))
    a, _, _ = masked(list(range(0, 128)))
    b, _, _ = masked(list(range(128, 256)))
    print("two masked streams (0..127 | 128..255): %.2f ms" % run_on([a, b]))
    a, _, _ = masked(list(range(0, 256, 2)))
    b, _, _ = masked(list(range(1, 256, 2)))
    print("two masked streams (even | odd bits): %.2f ms" % run_on([a, b]))
    sts = [masked(list(range(16 * i, 16 * i + 16)))[0] for i in range(8)]
    print("eight masked streams of 16 contiguous bits each, eight launches: %.2f ms" % run_on(sts))
    print("eight plain streams, eight launches: %.2f ms" % run_on([torch.cuda.Stream() for _ in range(8)]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise the per-wavefront times written by a -DMCQ_WAVE_TIMES build (tools/wave_times.sh): one line per wavefront,
`block start end xcc se cu simd slot`, times in 10 ns ticks from the first start."""
import sys

import numpy as np


def main(path):
    d = np.loadtxt(path, dtype=np.int64)
    blk, t0, t1, xcc, se, cu, simd, slot = d.T
    dur = (t1 - t0) / 100.0  # microseconds
    q = lambda x, p: np.percentile(x, p)
    print(f"{len(blk)} wavefronts, start spread {t0.max() / 100.0:.1f} us; duration us: min {dur.min():.0f} p5 {q(dur, 5):.0f} "
          f"median {np.median(dur):.0f} p95 {q(dur, 95):.0f} max {dur.max():.0f}")
    simds = np.unique(((xcc * 4 + se) * 16 + cu) * 4 + simd, return_counts=True)[1]
    print(f"SIMDs used {len(simds)}, wavefronts per SIMD: {dict(zip(*np.unique(simds, return_counts=True)))}")
    for s in np.unique(slot):
        m = slot == s
        print(f"  wave slot {s}: n {m.sum():5d}  duration median {np.median(dur[m]):.0f} us  (p5 {q(dur[m], 5):.0f}, p95 {q(dur[m], 95):.0f})")
    for x in np.unique(xcc):
        m = xcc == x
        print(f"  XCC {x}: n {m.sum():5d}  end median {np.median(t1[m]) / 100.0:.0f} us  max {t1[m].max() / 100.0:.0f} us")


if __name__ == "__main__":
    main(sys.argv[1])

"""Runs the ctypes stub printed in INTEGRATION.md section 2 -- extracted from the document itself, only the library path made
absolute -- against the package: what a maintainer of the reference would paste must give mcq_amd.run_experiment's results."""
import os
import re
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def doc_block():
    s = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.search(r"```python\n# experiments.py \(reference side\)\n(.*?)```", s, re.S).group(1)


def stub_namespace():
    """exec the documented stub; `build_schedule_from_params` is the reference's own function there (experiments.py:79-105),
    here the package's drop-in of the same name."""
    sys.path.insert(0, ROOT)
    import mcq_amd

    src = doc_block().replace('C.CDLL("libmcq_hip.so")', "C.CDLL(%r)" % os.path.join(ROOT, "monte-carlo-collective_amd/csrc/libmcq_hip.so"))
    ns = {"build_schedule_from_params": mcq_amd.build_schedule_from_params}
    exec(compile(src, "INTEGRATION.md", "exec"), ns)
    return ns, mcq_amd


if __name__ == "__main__":
    import numpy as np

    ns, mcq_amd = stub_namespace()
    for sp, mode in (({"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}, "board"),
                     ({"type": "exponential_annealing", "beta_start": 0.5, "beta_end": 3.0}, "full_3d")):
        a = ns["run_experiment"](12, 3000, "random", None, 24, base_seed=42, schedule_params=sp, mcmc_type=mode, early_stop_patience=None)
        b = mcq_amd.run_experiment(12, 3000, "random", None, 24, base_seed=42, schedule_params=sp, mcmc_type=mode, early_stop_patience=None)
        assert all((np.asarray(x) == np.asarray(y)).all() for x, y in zip(a[0], b[0])), "histories"
        assert list(a[1]) == list(b[1]) and list(a[5]) == list(b[5])
        assert all((np.asarray(x) == np.asarray(y)).all() for x, y in zip(a[3], b[3]))
    print("INTEGRATION.md stub == mcq_amd.run_experiment:", a[1][:6])

import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import mcq_amd
from oracle import oracle
abi = mcq_amd.abi
sets = [{"type": "linear_annealing", "beta_start": s, "beta_end": e} for s, e in ((0.5, 3.0), (1.0, 3.0), (0.1, 5.0))]
p = abi.make_params_sets(6, 700, "random", sets, 32, mcmc_type="board", early_stop_patience=120)
seeds = np.concatenate([abi.seeds_for(42 + 1000 * i, 32) for i in range(3)])
want = oracle.run(p, seeds)
red, _ = mcq_amd._lib.run_host(p, seeds, trace="reduced", states=False)
print("hist_len equal", np.array_equal(red["hist_len"], want["hist_len"]))
for t in range(3):
    sl = slice(32 * t, 32 * t + 32)
    st = mcq_amd.jobs.stats_from_trace({k: v[sl] for k, v in want.items()}, 700)
    for k in ("step_sum", "step_sumsq", "step_accepted", "step_count"):
        d = np.flatnonzero(red[k][t] != st[k])
        print("set", t, k, "mismatches", len(d), d[:10], (red[k][t][d[:10]] - st[k][d[:10]]))
# the JobSet path
jobs = [mcq_amd.jobs.make_job(6, 700, "random", sp, 32, 42 + 1000 * i, "board", 120) for i, sp in enumerate(sets)]
out = mcq_amd.jobs.JobSet(jobs, want="stats").run()
for t in range(3):
    sl = slice(32 * t, 32 * t + 32)
    st = mcq_amd.jobs.stats_from_trace({k: v[sl] for k, v in want.items()}, 700)
    for k in ("step_sum", "step_accepted", "step_count", "step_stopped"):
        d = np.flatnonzero(out[t][k] != st[k])
        print("jobset", t, k, "mismatches", len(d), d[:10], (out[t][k][d[:10]] - st[k][d[:10]]))

"""Sweep time of the headline board problem (20 000 steps) without early stopping, with a patience equal to the run length (it
cannot stop a chain that ever improves, but selects the early-stop kernel variants: what the reference's default
early_stop_patience = 100000 does for runs of 100 000 steps and more) and with a patience that does stop chains.
usage (GPU box): python tools/bench_patience.py"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, mcq_amd
abi = mcq_amd.abi
sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
for pat in (None, 20000, 2000):
    p = abi.make_params(12, 20000, "random", sp, 65536, mcmc_type="board", early_stop_patience=pat, trace=True)
    run = mcq_amd._lib.DeviceRun(p, abi.seeds_for(42, 65536), trace=True, states=False)
    run.launch(); torch.cuda.synchronize()
    ms = [run.launch_timed()[1] for _ in range(5)]  # (single launches scatter by ~1 %)
    print("patience", pat, "sweep ms min %.2f mean %.2f of 5" % (min(ms), sum(ms) / len(ms)), "executed", int(run.t["steps_executed"].sum().item()))
    del run; torch.cuda.empty_cache()

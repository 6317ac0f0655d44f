"""Sharding of chains over ranks and the node-level summary reduce.

Chains never interact during a sweep (experiments.py:507-517), so rank g simply runs the
contiguous block [lo, hi) of the chain indices with the seeds base_seed + global index: the
results do not depend on the number of GPUs and there is no data-path collective.  The only
exchange is the summary at the end: MIN over best energies, SUM over counters -- one small
all-reduce each (RCCL over xGMI on a GPU node: backend "nccl"; gloo on CPU for the tests).
"""
import numpy as np

from . import abi


def shard_bounds(n_runs, rank, world):
    """Contiguous block of chain indices for `rank`: ceil(n/world) per rank, the last ones may be short or empty."""
    per = -(-int(n_runs) // int(world))
    lo = min(n_runs, rank * per)
    return lo, min(n_runs, lo + per)


def shard_seeds(base_seed, n_runs, rank, world):
    """Seeds of this rank's chains: base_seed + global chain index (experiments.py:508)."""
    lo, hi = shard_bounds(n_runs, rank, world)
    return abi.seeds_for(int(base_seed) + lo, hi - lo), lo, hi


SUMMARY_FIELDS = ("n_chains", "accepted", "proposed", "sum_best", "sumsq_best", "sum_steps_to_best")


def local_summary(res):
    """Per-rank summary of a result dict (NumPy arrays as described in include/mcq.h)."""
    best = np.asarray(res["best_energy"], dtype=np.int64)
    sums = np.array([len(best), int(np.sum(res["n_accepted"])), int(np.sum(res["steps_executed"])), int(best.sum()),
                     int((best * best).sum()), int(np.sum(res["steps_to_best"]))], dtype=np.int64)
    mn = np.array([int(best.min()) if len(best) else np.iinfo(np.int64).max], dtype=np.int64)
    return mn, sums


def reduce_summary(mn, sums, dist=None, device="cpu"):
    """All-reduce (MIN, SUM) over the default process group; returns the node-level summary dict.

    `dist` is torch.distributed (already initialised) or None for a single process."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        import torch

        t_mn = torch.as_tensor(np.asarray(mn), dtype=torch.int64, device=device)
        t_sm = torch.as_tensor(np.asarray(sums), dtype=torch.int64, device=device)
        dist.all_reduce(t_mn, op=dist.ReduceOp.MIN)
        dist.all_reduce(t_sm, op=dist.ReduceOp.SUM)
        mn, sums = t_mn.cpu().numpy(), t_sm.cpu().numpy()
    out = dict(zip(SUMMARY_FIELDS, (int(v) for v in sums)))
    n = max(1, out["n_chains"])
    out["min_best"] = int(np.asarray(mn).reshape(-1)[0])
    out["mean_best"] = out["sum_best"] / n
    out["std_best"] = float(np.sqrt(max(0.0, out["sumsq_best"] / n - out["mean_best"] ** 2)))  # population std, as np.std (experiments.py:1080)
    out["mean_steps_to_best"] = out["sum_steps_to_best"] / n
    out["acceptance_rate"] = out["accepted"] / max(1, out["proposed"])
    return out


def run_experiment_sharded(N, n_steps, init_mode, schedule_params, n_runs, base_seed=0, mcmc_type="full_3d",
                           early_stop_patience=None, dist=None, runner=None, device="cpu", trace=False):
    """This rank's shard of run_experiment plus the reduced node-level summary.

    `runner(N, n_steps, init_mode, schedule_params, seeds, mcmc_type=..., early_stop_patience=..., trace=...)`
    returns (result dict, seconds); the default is the GPU path (experiments.run_chains)."""
    if runner is None:
        from . import experiments

        runner = experiments.run_chains
    rank = dist.get_rank() if dist is not None and dist.is_initialized() else 0
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    seeds, lo, hi = shard_seeds(base_seed, n_runs, rank, world)
    res, secs = runner(N, n_steps, init_mode, schedule_params, seeds, mcmc_type=mcmc_type,
                       early_stop_patience=early_stop_patience, trace=trace)
    mn, sums = local_summary(res)
    return res, reduce_summary(mn, sums, dist=dist, device=device), (lo, hi), secs

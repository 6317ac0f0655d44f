"""Sharding of chains over ranks and the node-level reduce.

Chains never interact during a sweep (experiments.py:507-517), so rank g simply runs the contiguous block
[lo, hi) of the chain indices with the seeds base_seed + global index: the results do not depend on the number
of GPUs and there is no data-path collective.  The only exchange is at the end of a job list, and it is ONE
all-reduce (SUM, int64) of one packed tensor over the default process group (RCCL over xGMI on a GPU node:
backend "nccl"; gloo on CPU for the tests):

    per job   6 counters                      n_chains, accepted, proposed, sum / sum of squares of best_energy,
                                              sum of steps_to_best
              `world` slots                   rank r writes its local minimum of best_energy (+1) into slot r and zeros
                                              elsewhere: the SUM then holds every rank's minimum and the node-level MIN is a
                                              local min over the slots -- no second collective with another reduce op
              2 x n_runs per-chain slots      (optional) best_energy / steps_to_best of chain i in slot i, written by the rank
                                              that owns chain i: the SUM is a gather (what measure_min_energy_vs_N and
                                              run_beta_start_end_pairs return per run, experiments.py:1074-1096, 843-846)
              5 x (n_steps + 1) per-step sums (optional) sum E, sum E^2, accepted, count, stopped-here -- what the mean / std
                                              and the binned acceptance CSVs are computed from (experiments.py:593-608, 660-711)

The tensor lives where the results live (HBM for the GPU path), so nothing is staged through the host before the
collective.
"""
import math

import numpy as np

from . import abi

N_COUNTERS = 6
COUNTER_FIELDS = ("n_chains", "accepted", "proposed", "sum_best", "sumsq_best", "sum_steps_to_best")
STAT_FIELDS = ("step_sum", "step_sumsq", "step_accepted", "step_count", "step_stopped")


def shard_bounds(n_runs, rank, world):
    """Contiguous block of chain indices for `rank`: ceil(n/world) per rank, the last ones may be short or empty."""
    per = -(-int(n_runs) // int(world))
    lo = min(n_runs, rank * per)
    return lo, min(n_runs, lo + per)


def shard_seeds(base_seed, n_runs, rank, world):
    """Seeds of this rank's chains: base_seed + global chain index (experiments.py:508)."""
    lo, hi = shard_bounds(n_runs, rank, world)
    return abi.seeds_for(int(base_seed) + lo, hi - lo), lo, hi


def rank_world(dist):
    if dist is not None and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class JobLayout:
    """Where one job's fields sit inside the packed tensor."""

    def __init__(self, offset, n_runs, n_steps, world, per_chain, stats):
        self.n_runs, self.n_steps, self.world, self.per_chain, self.stats = int(n_runs), int(n_steps), int(world), per_chain, stats
        self.counters = offset
        self.mins = self.counters + N_COUNTERS
        self.best = self.mins + world
        self.stb = self.best + (self.n_runs if per_chain else 0)
        self.stat0 = self.stb + (self.n_runs if per_chain else 0)
        self.end = self.stat0 + (len(STAT_FIELDS) * (self.n_steps + 1) if stats else 0)


def layout_for(jobs, world, per_chain=True, stats=False):
    """jobs: iterable of (n_runs, n_steps); returns ([JobLayout], total words)."""
    out, off = [], 0
    for n_runs, n_steps in jobs:
        lay = JobLayout(off, n_runs, n_steps, world, per_chain, stats)
        out.append(lay)
        off = lay.end
    return out, off


def pack_job(buf, lay, rank, lo, res, torch):
    """Write this rank's results of one job into its slots of `buf` (an int64 tensor, zero-initialised).

    `res` maps field names to torch tensors (any integer dtype) on buf's device: best_energy, steps_to_best, n_accepted,
    steps_executed of the local chains [lo, lo + n_local), and with lay.stats the five STAT_FIELDS arrays."""
    best = res["best_energy"].to(torch.int64)
    stb = res["steps_to_best"].to(torch.int64)
    n = best.numel()
    c = buf[lay.counters:lay.counters + N_COUNTERS]
    c[0] = n
    if n:
        c[1] = res["n_accepted"].sum()
        c[2] = res["steps_executed"].sum()
        c[3] = best.sum()
        c[4] = (best * best).sum()
        c[5] = stb.sum()
        buf[lay.mins + rank] = best.min() + 1  # 0 = "this rank has no chain of the job"
        if lay.per_chain:
            buf[lay.best + lo:lay.best + lo + n] = best
            buf[lay.stb + lo:lay.stb + lo + n] = stb
    if lay.stats:
        L = lay.n_steps + 1
        for k, name in enumerate(STAT_FIELDS):
            buf[lay.stat0 + k * L:lay.stat0 + (k + 1) * L] = res[name].to(torch.int64)


def all_reduce_packed(buf, dist):
    """THE collective: one SUM all-reduce of the packed tensor (a no-op for a single process)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def summary_from_counters(counters, mins):
    """Node-level summary dict of one job from its reduced counters and per-rank minimum slots (NumPy int64 arrays)."""
    out = dict(zip(COUNTER_FIELDS, (int(v) for v in counters)))
    n = max(1, out["n_chains"])
    present = [int(m) - 1 for m in mins if int(m) > 0]
    out["min_best"] = min(present) if present else None
    out["mean_best"] = out["sum_best"] / n
    # population std, as np.std (experiments.py:1080), from exact integers
    out["std_best"] = math.sqrt(max(0, n * out["sumsq_best"] - out["sum_best"] ** 2)) / n  # exact Python integers under the root
    out["mean_steps_to_best"] = out["sum_steps_to_best"] / n
    out["acceptance_rate"] = out["accepted"] / max(1, out["proposed"])
    return out


def unpack_job(host, lay):
    """Reduced fields of one job from the packed tensor copied to the host (NumPy int64).  The arrays are VIEWS of `host` (66 MB
    for BASELINE configs[4]: copying them out again cost 9 ms of a 25 ms reduce, profiles/r03_reduce_path.txt); the caller hands
    over a host array of its own per reduce, which the views keep alive."""
    out = {"summary": summary_from_counters(host[lay.counters:lay.counters + N_COUNTERS], host[lay.mins:lay.mins + lay.world])}
    if lay.per_chain:
        out["best_energy"] = host[lay.best:lay.best + lay.n_runs]
        out["steps_to_best"] = host[lay.stb:lay.stb + lay.n_runs]
    if lay.stats:
        L = lay.n_steps + 1
        for k, name in enumerate(STAT_FIELDS):
            out[name] = host[lay.stat0 + k * L:lay.stat0 + (k + 1) * L]
    return out


def reduce_summary(res, dist=None, device=None):
    """Node-level summary of ONE job whose local results are `res` (torch tensors on `device`, or NumPy arrays): one
    packed SUM all-reduce (bench.py's per-launch summary).  Returns the summary dict, identical on every rank."""
    import torch

    rank, world = rank_world(dist)
    if device is None:
        v = res["best_energy"]
        device = v.device if hasattr(v, "device") and not isinstance(v, np.ndarray) else "cpu"
    t = {k: (torch.as_tensor(np.asarray(v).astype(np.int64) if isinstance(v, np.ndarray) else v, device=device))
         for k, v in res.items() if k in ("best_energy", "steps_to_best", "n_accepted", "steps_executed")}
    (lay,), total = layout_for([(0, 0)], world, per_chain=False, stats=False)
    buf = torch.zeros(total, dtype=torch.int64, device=device)
    pack_job(buf, lay, rank, 0, t, torch)
    all_reduce_packed(buf, dist)
    return unpack_job(buf.cpu().numpy(), lay)["summary"]


def run_experiment_sharded(N, n_steps, init_mode, schedule_params, n_runs, base_seed=0, mcmc_type="full_3d",
                           early_stop_patience=None, dist=None, runner=None, device="cpu", trace=False):
    """This rank's shard of run_experiment plus the reduced node-level summary.

    `runner(N, n_steps, init_mode, schedule_params, seeds, mcmc_type=..., early_stop_patience=..., trace=...)`
    returns (result dict, seconds); the default is the GPU path (experiments.run_chains)."""
    if runner is None:
        from . import experiments

        runner = experiments.run_chains
    rank, world = rank_world(dist)
    seeds, lo, hi = shard_seeds(base_seed, n_runs, rank, world)
    res, secs = runner(N, n_steps, init_mode, schedule_params, seeds, mcmc_type=mcmc_type,
                       early_stop_patience=early_stop_patience, trace=trace)
    return res, reduce_summary(res, dist=dist, device=device), (lo, hi), secs

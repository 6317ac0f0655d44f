"""Device-resident, sharded execution of an experiment driver's job list.

One *job* is one run_experiment call of the reference (experiments.py:475-573): n_runs chains of one (N, init_mode,
schedule, mcmc_type) seeded base_seed + r.  The drivers loop over such jobs -- one per beta pair in
run_beta_start_end_pairs (experiments.py:777-806), one per (init_mode, N) cell in measure_min_energy_vs_N
(experiments.py:1050-1096).  Here the whole list runs at once:

* every rank takes the contiguous block [lo, hi) of each job's chains (distributed.shard_bounds), seeds base_seed + global
  index, so results do not depend on the number of GPUs;
* jobs that differ only in schedule, init mode and seeds become ONE launch with one set per job (mcq_params.sets): the beta
  pairs of a pair experiment, the init modes of one N in measure_min_energy_vs_N; the others get a launch and a HIP stream each,
  all enqueued back to back;
* nothing but per-chain scalars and, on request, per-step integer sums ever exists: `want="stats"` runs with
  trace = REDUCED, so a job of 8 192 chains x 10^5 steps never materialises its 3.3 GB trace, on the device or on the host;
* the node-level result is ONE packed SUM all-reduce (distributed.py) -- after it every rank holds the per-job minima,
  counters, per-run best energies / steps-to-best and the per-step sums the CSV writers need.

`want="histories"` is the small-run path that returns full energy histories to the host like the reference does.
"""
import time

import numpy as np

from . import abi, distributed as dm


def make_job(N, n_steps, init_mode, schedule_params, n_runs, base_seed, mcmc_type="full_3d", early_stop_patience=None):
    """One run_experiment call.  The reference's n_runs == 1 branch does not forward early_stop_patience
    (experiments.py:550-558) and n_runs > 1 needs schedule_params (experiments.py:505-506)."""
    if n_runs > 1 and schedule_params is None:
        raise ValueError("schedule_params is required for parallel execution when n_runs > 1")
    return dict(N=int(N), n_steps=int(n_steps), init_mode=init_mode, schedule_params=schedule_params, n_runs=int(n_runs),
                base_seed=int(base_seed), mcmc_type=mcmc_type, early_stop_patience=early_stop_patience if n_runs > 1 else None)


def _stopped_hist(hist_len, n_steps, xp, **kw):
    """step_stopped[e] = chains that stopped early with e as the entry they did not append (hist_len == e <= n_steps)."""
    hl = hist_len[hist_len <= n_steps]
    return xp.bincount(hl, minlength=n_steps + 1, **kw)[: n_steps + 1]


class _Launch:
    """One DeviceRun and the jobs (schedule sets) it carries."""

    def __init__(self, job_ids, run, n_local, cps):
        self.job_ids, self.run, self.n_local, self.cps = job_ids, run, n_local, cps
        self.stream = None


class JobSet:
    """Allocate once, launch any number of times (bench.py), reduce after each launch."""

    def __init__(self, jobs, want="summary", dist=None, runner=None, lanes_per_chain=0, rng="mt19937"):
        if want not in ("summary", "stats", "histories"):
            raise ValueError(f"unknown want: {want}")
        self.jobs, self.want, self.dist, self.runner = list(jobs), want, dist, runner
        self.rank, self.world = dm.rank_world(dist)
        if want == "histories" and self.world > 1:
            raise ValueError("full histories are a single-process product; use want='stats' when the chains are sharded")
        self.shards = []
        for j in self.jobs:
            seeds, lo, hi = dm.shard_seeds(j["base_seed"], j["n_runs"], self.rank, self.world)
            self.shards.append((seeds, lo, hi))
        self.layouts, self.total_words = dm.layout_for([(j["n_runs"], j["n_steps"]) for j in self.jobs], self.world,
                                                       per_chain=True, stats=want == "stats")
        self.trace = {"summary": False, "stats": "reduced", "histories": True}[want]
        self.launches, self.local = [], [None] * len(self.jobs)
        self.launch_seconds = 0.0
        if runner is None:
            self._allocate(lanes_per_chain, rng)

    # ---- GPU path --------------------------------------------------------------------------------------------------
    def _allocate(self, lanes_per_chain, rng):
        import torch

        from . import _lib

        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device())
        groups = {}
        for i, j in enumerate(self.jobs):
            n = len(self.shards[i][0])
            key = (j["N"], j["n_steps"], j["mcmc_type"], j["early_stop_patience"], n)  # schedule, init mode and seeds may differ inside a launch
            batchable = n > 0 and n % 16 == 0 and j["schedule_params"] is not None
            groups.setdefault(key if batchable else ("single", i), []).append(i)
        for key, ids in groups.items():
            j0 = self.jobs[ids[0]]
            n = len(self.shards[ids[0]][0])
            if n == 0:
                continue
            if len(ids) > 1:
                p = abi.make_params_sets(j0["N"], j0["n_steps"], j0["init_mode"], [self.jobs[i]["schedule_params"] for i in ids], n,
                                         mcmc_type=j0["mcmc_type"], early_stop_patience=j0["early_stop_patience"], trace=self.trace,
                                         lanes_per_chain=lanes_per_chain, rng=rng, init_modes=[self.jobs[i]["init_mode"] for i in ids])
                seeds = np.concatenate([self.shards[i][0] for i in ids])
            else:
                p = abi.make_params(j0["N"], j0["n_steps"], j0["init_mode"], j0["schedule_params"], n, mcmc_type=j0["mcmc_type"],
                                    early_stop_patience=j0["early_stop_patience"], trace=self.trace, lanes_per_chain=lanes_per_chain, rng=rng)
                seeds = self.shards[ids[0]][0]
            run = _lib.DeviceRun(p, seeds, trace=self.trace, states=False)
            la = _Launch(ids, run, n, n)
            la.stream = torch.cuda.Stream()
            self.launches.append(la)
        # longest first (a step costs roughly N lane-operations per chain): the short launches then fill the tail of the long ones
        self.launches.sort(key=lambda la: -(la.run.p.N * la.run.p.n_steps * la.run.p.n_chains))
        self.buf = torch.zeros(self.total_words, dtype=torch.int64, device=self.device)

    def launch(self):
        """Enqueue every launch on its stream; returns immediately (GPU path) or after the injected runner has run."""
        t0 = time.perf_counter()
        if self.runner is not None:
            self._run_injected()
        else:
            cur = self.torch.cuda.current_stream()
            for la in self.launches:
                la.stream.wait_stream(cur)
                la.run.launch(stream=la.stream)
        self.launch_seconds = time.perf_counter() - t0

    def synchronize(self):
        if self.runner is None:
            for la in self.launches:
                la.stream.synchronize()

    def _local_views(self):
        """Per job: this rank's result tensors (views into the launches' output buffers)."""
        torch = self.torch
        out = [None] * len(self.jobs)
        for la in self.launches:
            t = la.run.t
            n_steps = int(la.run.p.n_steps)
            for s, i in enumerate(la.job_ids):
                sl = slice(s * la.cps, s * la.cps + la.n_local)
                r = {k: t[k][sl] for k in ("best_energy", "steps_to_best", "n_accepted", "steps_executed", "hist_len")}
                if self.want == "stats":
                    for k in ("step_sum", "step_sumsq", "step_accepted", "step_count"):
                        r[k] = t[k][s] if len(la.job_ids) > 1 else t[k]
                    r["step_stopped"] = _stopped_hist(r["hist_len"], n_steps, torch)
                out[i] = r
        return out

    # ---- injected runner (tests: the CPU oracle) --------------------------------------------------------------------
    def _run_injected(self):
        for i, j in enumerate(self.jobs):
            seeds = self.shards[i][0]
            if len(seeds) == 0:
                self.local[i] = None
                continue
            res, _ = self.runner(j["N"], j["n_steps"], j["init_mode"], j["schedule_params"], seeds, mcmc_type=j["mcmc_type"],
                                 early_stop_patience=j["early_stop_patience"], trace=self.want != "summary")
            self.local[i] = res

    def _injected_views(self):
        import torch

        self.torch = torch
        out = [None] * len(self.jobs)
        for i, j in enumerate(self.jobs):
            res = self.local[i]
            if res is None:
                continue
            r = {k: torch.from_numpy(np.ascontiguousarray(res[k]).astype(np.int64))
                 for k in ("best_energy", "steps_to_best", "n_accepted", "steps_executed", "hist_len")}
            if self.want == "stats":
                r.update({k: torch.from_numpy(v) for k, v in stats_from_trace(res, j["n_steps"]).items()})
            out[i] = r
        return out

    # ---- the reduce ---------------------------------------------------------------------------------------------------
    def reduce(self):
        """Pack this rank's results, all-reduce once, and return one dict per job (identical on every rank):
        summary, best_energy[n_runs], steps_to_best[n_runs], and with want="stats" the five per-step arrays."""
        if self.runner is None:
            torch = self.torch
            cur = torch.cuda.current_stream()
            for la in self.launches:  # everything below runs on the current stream, behind every launch
                cur.wait_stream(la.stream)
            torch_views = self._local_views()
            buf = self.buf
            buf.zero_()
        else:
            torch_views = self._injected_views()
            torch = self.torch
            dev = "cpu"
            if self.dist is not None and self.dist.is_initialized() and self.dist.get_backend() == "nccl":
                dev = torch.device("cuda", torch.cuda.current_device())
            buf = torch.zeros(self.total_words, dtype=torch.int64, device=dev)
        empty = torch.zeros(0, dtype=torch.int64, device=buf.device)
        for i, lay in enumerate(self.layouts):
            r = torch_views[i]
            if r is None:
                zeros = {k: torch.zeros(lay.n_steps + 1, dtype=torch.int64, device=buf.device) for k in dm.STAT_FIELDS} if lay.stats else {}
                r = dict(best_energy=empty, steps_to_best=empty, n_accepted=empty, steps_executed=empty, **zeros)
            else:
                r = {k: v.to(buf.device) for k, v in r.items()}
            dm.pack_job(buf, lay, self.rank, self.shards[i][1], r, torch)
        if buf.device.type == "cuda" and self.world > 1 and self.dist.get_backend() != "nccl":
            buf = buf.cpu()  # a host-side process group (gloo: tests with several ranks on one GPU) reduces host tensors
        dm.all_reduce_packed(buf, self.dist)
        host = buf.cpu().numpy()
        out = [dm.unpack_job(host, lay) for lay in self.layouts]
        if self.want == "histories":
            self._attach_histories(out)
        return out

    def _attach_histories(self, out):
        if self.runner is not None:
            for i, res in enumerate(self.local):
                if res is not None:
                    out[i].update({k: res[k] for k in ("energy_hist", "accept_bits", "hist_len", "steps_executed")})
            return
        for la in self.launches:
            res = la.run.results()
            for s, i in enumerate(la.job_ids):
                sl = slice(s * la.cps, s * la.cps + la.n_local)
                out[i].update({k: res[k][sl] for k in ("energy_hist", "accept_bits", "hist_len", "steps_executed")})

    def run(self):
        """launch + reduce; returns the per-job results."""
        self.launch()
        return self.reduce()


def stats_from_trace(res, n_steps):
    """The five per-step integer arrays of distributed.STAT_FIELDS from a host result dict with a full trace (what the
    REDUCED trace accumulates on the device)."""
    L = np.asarray(res["hist_len"], dtype=np.int64)
    ex = np.asarray(res["steps_executed"], dtype=np.int64)
    n = len(L)
    h = np.asarray(res["energy_hist"])[:, : n_steps + 1].astype(np.int64)
    valid = np.arange(n_steps + 1)[None, :] < L[:, None]
    hv = np.where(valid, h, 0)
    bits = np.unpackbits(np.ascontiguousarray(res["accept_bits"]).view(np.uint8), axis=1, bitorder="little")[:, :max(n_steps, 0)]
    executed = np.arange(n_steps)[None, :] < ex[:, None]
    acc = np.zeros(n_steps + 1, dtype=np.int64)
    if n_steps > 0 and n > 0:
        acc[1:] = (bits.astype(bool) & executed).sum(axis=0)
    return {"step_sum": hv.sum(axis=0), "step_sumsq": (hv * hv).sum(axis=0), "step_accepted": acc,
            "step_count": valid.sum(axis=0).astype(np.int64), "step_stopped": _stopped_hist(L, n_steps, np).astype(np.int64)}


def run_jobs(jobs, want="summary", dist=None, runner=None, lanes_per_chain=0, rng="mt19937"):
    """One-shot form of JobSet: allocate, launch, reduce."""
    return JobSet(jobs, want=want, dist=dist, runner=runner, lanes_per_chain=lanes_per_chain, rng=rng).run()


# ---- what the CSV artefacts are computed from ----------------------------------------------------------------------------
def mean_std_from_sums(step_sum, step_sumsq, step_count):
    """Per-step mean and population std from integer sums.  The mean equals np.mean of the int64 matrix bit for bit
    (every partial sum is exact in float64 below 2^53); the variance comes from the exact integer numerator
    n * sum(x^2) - sum(x)^2, so the std agrees with NumPy's two-pass formula to ~1e-15 relative."""
    s = np.asarray(step_sum, dtype=np.int64)
    q = np.asarray(step_sumsq, dtype=np.int64)
    n = np.asarray(step_count, dtype=np.int64)
    with np.errstate(invalid="ignore", divide="ignore"):
        mean = s / n
    no, qo, so = n.astype(object), q.astype(object), s.astype(object)  # Python integers: n * sum(x^2) passes 2^63 at BASELINE sizes
    num, den = no * qo - so * so, no * no
    var = np.array([v / d if d else np.nan for v, d in zip(num, den)], dtype=np.float64)  # int / int: correctly rounded
    return mean, np.sqrt(np.maximum(var, 0.0))


def acceptance_bins_from_steps(step_accepted, step_count, step_stopped, n_steps, n_bins=100):
    """Binned acceptance of plot_acceptance_rates_binned (experiments.py:660-695) from the per-step sums: step s was
    proposed by count[s + 1] + stopped[s + 1] chains and accepted by accepted[s + 1] of them.  Bins: np.linspace(0, n_steps,
    n_bins + 1), left-closed (the last one closed on both sides, which adds nothing: steps end at n_steps - 1); empty bins NaN.
    Returns (bin centers, rates, accepted per bin, proposed per bin)."""
    acc = np.asarray(step_accepted, dtype=np.int64)[1:]
    pro = (np.asarray(step_count, dtype=np.int64) + np.asarray(step_stopped, dtype=np.int64))[1:]
    edges = np.linspace(0, n_steps, n_bins + 1)
    centers = (edges[:-1] + edges[1:]) / 2
    lo = np.minimum(np.ceil(edges).astype(np.int64), n_steps)  # step s is in bin b iff edges[b] <= s < edges[b + 1]
    lo[-1] = n_steps
    ca, cp = np.concatenate([[0], np.cumsum(acc)]), np.concatenate([[0], np.cumsum(pro)])
    a, p = ca[lo[1:]] - ca[lo[:-1]], cp[lo[1:]] - cp[lo[:-1]]
    with np.errstate(invalid="ignore", divide="ignore"):
        rates = np.where(p > 0, a / p, np.nan)
    return centers, rates, a, p

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
import json
import os
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
    if xp is np:
        hl = hist_len[hist_len <= n_steps]
        return xp.bincount(hl, minlength=n_steps + 1, **kw)[: n_steps + 1]
    # torch, on the device: a scatter-add with the full-length chains parked in a spare slot -- no boolean indexing and no bincount,
    # each of which waits for the device to learn a size
    idx = hist_len.to(xp.int64).clamp(max=n_steps + 1)
    return xp.zeros(n_steps + 2, dtype=xp.int64, device=hist_len.device).scatter_add_(0, idx, xp.ones_like(idx))[: n_steps + 1]


# Step time of ONE board wavefront with the SIMD to itself, ms per 20 000 steps, by N and lanes per chain (4, 8, 16): the measure a
# job list's lane plan and launch order are built on.  Lone-wave latency, not throughput: it ranks launches that run side by side
# below the device's capacity; above it the library default stands.  The numbers are DATA, not code: lane_table.json next to this
# file is written by tools/lane_table.py --json on one MI355X together with the sha256 of the kernel source it was measured on, and
# tests/test_host_logic.py fails when csrc/mcq_hip.hip changes without a new table.
LANE_TABLE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lane_table.json")


def _load_lane_table(path=LANE_TABLE_PATH):
    with open(path) as f:
        d = json.load(f)
    col = {int(g): k for k, g in enumerate(d["lanes"])}
    ms = {int(N): tuple(float(x) for x in row) for N, row in d["ms"].items()}
    if not {4, 8, 16} <= set(col) or not ms:
        raise ValueError(f"{path}: a lane table needs the columns 4, 8 and 16 lanes and at least one board size")
    return ms, col, d.get("kernel_sha256")


_LONE_MS, _LANE_COL, LANE_TABLE_SHA = _load_lane_table()
WAVES_PER_SIMD = 4  # resident wavefronts per SIMD of the sweep kernels (their register budget)


def lone_ms(N, lanes):
    """Estimated lone-wavefront sweep time of a board launch per 20 000 steps (sizes outside the table: the nearest measured one,
    scaled with N beyond the largest)."""
    N = int(N)
    top = max(_LONE_MS)
    t = _LONE_MS.get(N) or _LONE_MS[min(_LONE_MS, key=lambda k: abs(k - N))]
    lanes = int(lanes)
    if lanes not in _LANE_COL:  # (2 lanes per chain: 32 chains per wavefront, measured 1.3-1.45 x the step of 4 lanes for N = 3..12, profiles/r04_shapes.txt)
        return 1.4 * t[_LANE_COL[4]] * (max(N, top) / float(top))
    return t[_LANE_COL[lanes]] * (max(N, top) / float(top))


def _lds_bytes_per_wave(N, lanes):
    """LDS of one board wavefront (csrc: chain_lds_words_for): staging 16 + scalars 4 + ring 96 words, the heights, a pad; a slice is a
    multiple of 4 words and never 0 mod 8."""
    w = 116 + (N * N + 3) // 4 + (N + 2) // 4
    w = (w + 3) & ~3
    if w % 8 == 0:
        w += 4
    return (64 // lanes) * w * 4


def plan_lanes(shapes, simds, default_lanes):
    """Lanes per chain for launches that run side by side on one device.  `shapes`: one (N, n_chains, mode) per launch;
    `default_lanes(mode, N)`: the library's choice for a device full of that launch.  The library picks lanes for a launch on
    its own; a job list knows how many wavefronts its launches put on the device TOGETHER (capacity: WAVES_PER_SIMD per SIMD):
      * start from the library defaults; if together they overflow the capacity but 4 lanes everywhere would fit, everything
        runs at 4 -- one resident round beats a second one;
      * while the device is less than half full (under two wavefronts per SIMD a wavefront is bound by its own latency and
        further wavefronts cost the others next to nothing), the launch with the longest estimated step gets twice the lanes if
        that shortens its step -- the job list ends when its slowest launch does.  Closer to the capacity more wavefronts slow
        everybody down: measured on measure_min_energy_vs_N at 1 024 chains per cell, 4 lanes everywhere 199 ms, N = 3 at 16
        lanes 215 ms (profiles/r03_lane_plan.txt).
    full_3d launches keep the library default (returned as 0; counted with `default_lanes`).  Returns the list of lane counts."""
    cap = WAVES_PER_SIMD * int(simds)
    waves = lambda n, g: (n + 64 // g - 1) // (64 // g)
    total = lambda pl: sum(waves(n, g) for (_, n, _), g in zip(shapes, pl))
    board = [m == abi.MODE_BOARD for _, _, m in shapes]
    plan = [int(default_lanes(m, N)) for N, _, m in shapes]
    # Boards of N = 13..20 default to 8 lanes because at 4 their LDS slices (12-14 KB per wavefront) leave a CU 11-13 wavefronts
    # when such a launch has the device to itself.  In a MIX with smaller boards the light wavefronts fill what the heavy ones
    # leave, and 4 lanes -- 16 chains per wavefront, two thirds of the instructions per move -- win: measure_min_energy_vs_N at
    # 8 192 chains per cell 1 156 -> 1 097 ms (profiles/r03_lane_plan.txt).  The test is the average LDS per wavefront of the list.
    if sum(board) >= 2:
        four = [4 if b and N <= 20 else g for (N, _, _), b, g in zip(shapes, board, plan)]
        w4 = [waves(n, g) for (_, n, _), g in zip(shapes, four)]
        lds = sum(w * _lds_bytes_per_wave(N, g) for (N, _, _), g, w, b in zip(shapes, four, w4, board) if b)
        if lds <= 10.6 * 1024 * sum(w for w, b in zip(w4, board) if b):
            plan = four
    if total(plan) > cap:
        # (only where the 4-lane kernels are measured, N <= 24 -- beyond, 16 chains of N x N heights soon exceed the LDS of a CU:
        # N = 100 at 4 lanes is 169 KB per wavefront and the launch would be refused)
        four = [4 if b and N <= 24 else g for (N, _, _), b, g in zip(shapes, board, plan)]
        if total(four) <= cap:
            plan = four
    while 2 * total(plan) <= cap:
        est = [lone_ms(N, g) if b else 0.0 for (N, _, _), g, b in zip(shapes, plan, board)]
        i = max(range(len(plan)), key=lambda k: est[k], default=None)
        if i is None or not board[i] or plan[i] == 16 or lone_ms(shapes[i][0], 2 * plan[i]) >= est[i]:
            break
        trial = plan[:i] + [2 * plan[i]] + plan[i + 1:]
        if 2 * total(trial) > cap:
            break
        plan = trial
    # full_3d launches are left to the library (0): which kernel family it takes depends on more than N (trace, stream, exchange)
    return [g if b else 0 for g, b in zip(plan, board)]


class _Launch:
    """One DeviceRun and the jobs (schedule sets) it carries."""

    def __init__(self, job_ids, run, n_local, cps):
        self.job_ids, self.run, self.n_local, self.cps = job_ids, run, n_local, cps
        self.stream = None
        self.pack_slots = None
        self.priority = 0
        self.no_stops = None  # a zero step_stopped array, shared by the launch's jobs when none of its chains can stop early


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
        self._host, self._flip = [None, None], 0
        self.launch_seconds = 0.0
        if runner is None:
            self._allocate(lanes_per_chain, rng)

    # ---- GPU path --------------------------------------------------------------------------------------------------
    def _allocate(self, lanes_per_chain, rng):
        from . import _lib

        groups = {}
        for i, j in enumerate(self.jobs):
            n = len(self.shards[i][0])
            key = (j["N"], j["n_steps"], j["mcmc_type"], j["early_stop_patience"], n)  # schedule, init mode and seeds may differ inside a launch
            batchable = n > 0 and n % 16 == 0 and j["schedule_params"] is not None
            groups.setdefault(key if batchable else ("single", i), []).append(i)
        groups = {k: ids for k, ids in groups.items() if len(self.shards[ids[0]][0]) > 0}
        self.hw_queues = _lib.ensure_hw_queues(len(groups))  # one stream per launch: before the first GPU call of the process, if this is it
        import torch

        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device())
        L = _lib.lib()
        lanes = {k: lanes_per_chain for k in groups}
        if not lanes_per_chain and len(groups) > 0:  # the launches run side by side: choose their lane counts together
            keys = list(groups)
            shapes = [(self.jobs[groups[k][0]]["N"], len(self.shards[groups[k][0]][0]) * len(groups[k]),
                       abi.mode_of(self.jobs[groups[k][0]]["mcmc_type"])) for k in keys]
            lanes = dict(zip(keys, plan_lanes(shapes, L.mcq_device_simds(), L.mcq_default_lanes_n)))
            # (experiments only, tools/r04_shapes.sh: MCQ_LANES_PLAN="17:8,18:8" overrides the plan for boards of those N)
            for item in filter(None, os.environ.get("MCQ_LANES_PLAN", "").split(",")):
                n_, g_ = (int(x) for x in item.split(":"))
                for k, (N, _, m) in zip(keys, shapes):
                    if N == n_ and m == abi.MODE_BOARD:
                        lanes[k] = g_
        for key, ids in groups.items():
            j0 = self.jobs[ids[0]]
            n = len(self.shards[ids[0]][0])
            lanes_per_chain = lanes[key]
            if len(ids) > 1:
                p = abi.make_params_sets(j0["N"], j0["n_steps"], j0["init_mode"], [self.jobs[i]["schedule_params"] for i in ids], n,
                                         mcmc_type=j0["mcmc_type"], early_stop_patience=j0["early_stop_patience"], trace=self.trace,
                                         lanes_per_chain=lanes_per_chain, rng=rng, init_modes=[self.jobs[i]["init_mode"] for i in ids])
                seeds = np.concatenate([self.shards[i][0] for i in ids])
            else:
                p = abi.make_params(j0["N"], j0["n_steps"], j0["init_mode"], j0["schedule_params"], n, mcmc_type=j0["mcmc_type"],
                                    early_stop_patience=j0["early_stop_patience"], trace=self.trace, lanes_per_chain=lanes_per_chain, rng=rng)
                seeds = self.shards[ids[0]][0]
            run = _lib.DeviceRun(p, seeds, trace=self.trace, states=False, stream_words=os.environ.get("MCQ_JOB_STREAM_WORDS") == "1")  # (experiment hook: the count costs a job list one small kernel per launch)
            la = _Launch(ids, run, n, n)
            self.launches.append(la)  # (its stream: below, once it is known whether the launches get CUs of their own)
        # longest first, by the estimated time of one of its wavefronts: the launches that follow fill in behind it
        def est(la):
            p = la.run.p
            g = _lib.effective_lanes(p)
            return (lone_ms(p.N, g) if p.mode == abi.MODE_BOARD else 2.0 * lone_ms(p.N, 8)) * p.n_steps

        self.launches.sort(key=lambda la: -est(la))
        # Launches that run side by side far below two wavefronts per SIMD each (the (init, N) cells of measure_min_energy_vs_N on one GPU
        # of a node) do not pace themselves; left alone they share every SIMD alike, the short ones finish early and the long ones end
        # on a half-empty device.  The longest estimated launches get hardware priority (include/mcq.h: MCQ_FLAG_PRIORITY): they run
        # close to the pace of a lone wavefront while the others fill the gaps (profiles/r04_priority.txt).  Never changes a result.
        # Better still (and the default): the launches pace EACH OTHER.  A sweep kernel paces its wavefronts through a progress table -- every 64 steps
        # a wavefront takes the priority "number of co-resident wavefronts ahead of me" (+13 % on a full device) -- but only against
        # wavefronts of its own launch and only when it puts two or more on a SIMD; the launches of a job list put a fraction of a wavefront on a SIMD
        # each, so nobody paced anybody and the SIMD arbiter served the oldest wavefront first.  MCQ_FLAG_SHARED_PACING gives all launches of the
        # list ONE table (rows and slots are the hardware's SIMD / wave-slot ids, unique across kernels; progress = the step, equal-length launches):
        # measure_min_energy_vs_N at 1 024 chains per cell 200 -> 181 ms, at 512 175 -> 152 ms (profiles/r04_shapes.txt).  MCQ_JOB_PACING=0 falls back
        # to the static priorities.
        self.pacing = "none"
        total_waves = sum((int(la.run.p.n_chains) * _lib.effective_lanes(la.run.p) + 63) // 64 for la in self.launches)
        one_round = total_waves <= WAVES_PER_SIMD * int(L.mcq_device_simds())  # (beyond one resident round it measures nothing, or -2 % at 8 192 chains per cell)
        if len(self.launches) > 1 and one_round and os.environ.get("MCQ_JOB_PACING", "1") != "0":
            for la in self.launches:
                la.run.p.flags = int(la.run.p.flags) | abi.FLAG_SHARED_PACING
            self.pacing = "shared"
        elif len(self.launches) > 1 and os.environ.get("MCQ_JOB_PRIORITY", "1") != "0":
            self.pacing = "static priorities"
            ests = [est(la) for la in self.launches]
            top = ests[0]
            for la, e in zip(self.launches, ests):
                prio = 3 if e >= 0.85 * top else 2 if e >= 0.7 * top else 1 if e >= 0.55 * top else 0
                la.run.p.flags = int(la.run.p.flags) | abi.flag_priority(prio)
                la.priority = prio
        self._partition_cus(L)
        for la in self.launches:  # (a stream is a hardware queue: the plain ones are created only where no masked one was -- 18 + 18 queues oversubscribe the device's
            if la.stream is None:  # queue slots, and the scheduler then runs the launches one after the other: 3.4 s instead of 0.2 s, profiles/r04_cu_partition.txt)
                la.stream = torch.cuda.Stream()
        self.buf = torch.zeros(self.total_words, dtype=torch.int64, device=self.device)
        # page-locked host copies of the packed tensor, used in turn by reduce(): the first one here, outside anybody's timed region; the
        # second when a second reduce() comes (a one-shot JobSet(...).run() never pays for it)
        self._host = [None, torch.empty(self.total_words, dtype=torch.int64, pin_memory=True)]

    def _partition_cus(self, L):
        """Compute units of their own for GROUPS of the launches of a list that fills the device to less than 60 % (MCQ_CU_PARTITION=0 never, =1 whenever the list
        fits one resident round).  What a CU mask means on this device
        (tools/cu_mask_probe2.py, profiles/r04_cu_partition.txt): bit i belongs to XCD i % 8, an XCD's bits go round its 4 shader engines, workgroups are
        split evenly over XCDs and engines (so a launch runs at the pace of its worst-served engine), and an XCD without any bit is unrestricted.  The
        only clean units are therefore the 8 "layers" of 32 CUs (one CU in every engine of every XCD) = the 8 words of the mask: the launches are put
        into up to 8 groups of equal estimated work (longest first, each into the lightest group) and a group gets one layer."""
        self.cu_partition = None
        mode = os.environ.get("MCQ_CU_PARTITION", "auto")
        if len(self.launches) < 2 or mode == "0":
            return
        from . import _lib

        torch = self.torch
        n_cus = int(torch.cuda.get_device_properties(self.device).multi_processor_count)
        layers = n_cus // 32
        if layers < 2:
            return
        work = []
        for la in self.launches:
            p = la.run.p
            g = _lib.effective_lanes(p)
            waves = (int(p.n_chains) * g + 63) // 64
            lone = (lone_ms(p.N, g) if p.mode == abi.MODE_BOARD else 2.0 * lone_ms(p.N, 8)) * max(1, int(p.n_steps))
            work.append(waves * lone)
        total = sum((int(la.run.p.n_chains) * _lib.effective_lanes(la.run.p) + 63) // 64 for la in self.launches)
        # Where it pays (tools/r04_cu_partition.sh): the dispatcher does not spread small launches over the device -- two launches of 192 wavefronts side by side
        # take 14.6 ms where one takes 10.7 -- so a list that fills the device to 42 % runs 145 -> 101 ms on layers; at 84 % it measures nothing (180 / 186 ms).
        if total > WAVES_PER_SIMD * 4 * n_cus or (mode != "1" and (len(self.launches) < 4 or total > 0.6 * WAVES_PER_SIMD * 4 * n_cus)):
            return
        n_groups = min(layers, len(self.launches))
        load, member = [0.0] * n_groups, [0] * len(self.launches)
        for i in sorted(range(len(work)), key=lambda k: -work[k]):
            gmin = min(range(n_groups), key=lambda k: load[k])
            member[i], load[gmin] = gmin, load[gmin] + work[i]
        # spare layers (fewer groups than layers) go to the heaviest groups
        own = [[k] for k in range(n_groups)]
        for extra in range(n_groups, layers):
            own[max(range(n_groups), key=lambda k: load[k] / len(own[k]))].append(extra)
        self._masked = []
        for la, gidx in zip(self.launches, member):
            ids = [32 * layer + b for layer in own[gidx] for b in range(32)]
            h = _lib.cu_masked_stream(ids, n_cus)
            if h is None:  # the runtime refuses: plain streams
                for lb in self.launches:
                    lb.stream = None
                self.close()
                return
            self._masked.append(h)
            la.stream = torch.cuda.ExternalStream(h, device=self.device)
        self.cu_partition = member

    def close(self):
        """Release the CU-masked HIP streams of this JobSet (torch does not own external streams).  Called on garbage collection too."""
        masked, self._masked = getattr(self, "_masked", []), []
        if not masked:
            return
        try:
            from . import _lib

            for h in masked:
                _lib.destroy_stream(h)
        except Exception:  # interpreter shutdown: the import machinery or the runtime may be gone already
            pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def launch(self):
        """Enqueue every launch on its stream; returns immediately (GPU path) or after the injected runner has run."""
        t0 = time.perf_counter()
        if self.runner is not None:
            self._run_injected()
        else:
            cur = self.torch.cuda.current_stream()
            # ONE event on the current stream for all launches to wait on.  (An event per launch is an operation on the current stream between two
            # launches; CU-masked streams are "blocking" streams in HIP's sense -- hipExtStreamCreateWithCUMask takes no flags -- and torch's default
            # current stream is the legacy default stream, with which a blocking stream synchronises on every operation: the launches then ran one
            # after the other, 3.4 s instead of 0.2 s.)
            ready = self.torch.cuda.Event()
            ready.record(cur)
            for la in self.launches:
                la.stream.wait_event(ready)
                la.run.launch(stream=la.stream)
        self.launch_seconds = time.perf_counter() - t0

    def synchronize(self):
        if self.runner is None:
            for la in self.launches:
                la.stream.synchronize()

    def _pack_slots(self, la):
        """include/mcq.h: mcq_pack_slot per schedule set of the launch -- where its job's fields sit in the packed tensor."""
        arr = (abi.PackSlot * len(la.job_ids))()
        for s, i in enumerate(la.job_ids):
            lay, lo = self.layouts[i], self.shards[i][1]
            arr[s].counters, arr[s].min_slot = lay.counters, lay.mins + self.rank
            arr[s].best = lay.best + lo if lay.per_chain else -1
            arr[s].stb = lay.stb + lo if lay.per_chain else -1
            arr[s].stats = lay.stat0 if lay.stats else -1
        return arr

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
                    if int(la.run.p.patience) < 0 or int(la.run.p.patience) > n_steps:  # no chain of this launch can stop early
                        if la.no_stops is None:
                            la.no_stops = torch.zeros(n_steps + 1, dtype=torch.int64, device=self.device)
                        r["step_stopped"] = la.no_stops
                    else:
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
        summary, best_energy[n_runs], steps_to_best[n_runs], and with want="stats" the five per-step arrays.

        Lifetime of the returned arrays (GPU path): they are NumPy VIEWS of one of two page-locked host buffers this JobSet owns and
        uses in turn -- valid through the NEXT reduce() of this JobSet, overwritten by the one after, and pinned memory stays
        allocated while the JobSet (or any result array) lives.  Copy what must outlive two reduces."""
        native = self.runner is None and os.environ.get("MCQ_PACK", "native") != "torch"
        if self.runner is None:
            torch = self.torch
            cur = torch.cuda.current_stream()
            for la in self.launches:  # everything below runs on the current stream, behind every launch
                cur.wait_stream(la.stream)
            buf = self.buf
            buf.zero_()
            if native:
                # One or two small kernels per launch (mcq_pack_summary_device) instead of ~10 tensor operations per job: 54 jobs of
                # measure_min_energy_vs_N were ~400 tiny kernels, 6.5 ms behind a 180 ms sweep (profiles/r04_shapes.txt).
                import ctypes as C

                from . import _lib

                L = _lib.lib()
                for la in self.launches:
                    if la.pack_slots is None:
                        la.pack_slots = self._pack_slots(la)
                    _lib._check(L.mcq_pack_summary_device(C.byref(la.run.p), C.byref(la.run.out), la.n_local, la.pack_slots, buf.data_ptr(), C.c_void_p(cur.cuda_stream)))
                torch_views = None
            else:
                torch_views = self._local_views()
        else:
            torch_views = self._injected_views()
            torch = self.torch
            dev = "cpu"
            if self.dist is not None and self.dist.is_initialized() and self.dist.get_backend() == "nccl":
                dev = torch.device("cuda", torch.cuda.current_device())
            buf = torch.zeros(self.total_words, dtype=torch.int64, device=dev)
        empty = torch.zeros(0, dtype=torch.int64, device=buf.device)
        for i, lay in enumerate(self.layouts if torch_views is not None else ()):
            r = torch_views[i]
            if r is None:
                zeros = {k: torch.zeros(lay.n_steps + 1, dtype=torch.int64, device=buf.device) for k in dm.STAT_FIELDS} if lay.stats else {}
                r = dict(best_energy=empty, steps_to_best=empty, n_accepted=empty, steps_executed=empty, **zeros)
            else:
                r = {k: v.to(buf.device) for k, v in r.items()}
            dm.pack_job(buf, lay, self.rank, self.shards[i][1], r, torch)
        if buf.device.type == "cuda":
            # The results come back through one of two page-locked host buffers of the JobSet, used in turn (so the arrays of a
            # reduce stay intact through the next one and are overwritten by the one after: copy what must live longer).  A fresh
            # pageable array per reduce -- 66 MB for BASELINE configs[4] -- cost 10-20 ms of page faults and a slower copy on top of
            # the 88 ms sweep of that config's per-GPU shape (profiles/r03_reduce_path.txt).
            self._flip ^= 1
            if self._host[self._flip] is None:
                self._host[self._flip] = torch.empty(self.total_words, dtype=torch.int64, pin_memory=True)
            h = self._host[self._flip]
            if self.world > 1 and self.dist.get_backend() != "nccl":
                h.copy_(buf)  # a host-side process group (gloo: tests with several ranks on one GPU) reduces host tensors
                dm.all_reduce_packed(h, self.dist)
            else:
                dm.all_reduce_packed(buf, self.dist)
                h.copy_(buf, non_blocking=True)
                torch.cuda.current_stream().synchronize()
            host = h.numpy()
        else:
            dm.all_reduce_packed(buf, self.dist)
            host = buf.numpy()
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

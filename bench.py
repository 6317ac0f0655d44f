#!/usr/bin/env python3
"""bench.py -- Metropolis moves/sec (whole job) + min-energy-reached.

Default (the headline, BASELINE.json configs[1], `--config c2`):

    single_N, N=12, mcmc_type=board, init=random, linear_annealing 1.0 -> 3.0,
    65 536 chains per GPU x 100 000 steps, seeds 42 + global chain index, full int32 energy trace.

A bench "step" is one pass of the hot path over the whole batch: one launch of the init kernel and one launch of the
sweep kernel that runs every chain of this rank for n_steps proposals and writes the full energy_history / accept-bit
trace (and the best / final states) to HBM, followed by the node-level summary: ONE packed all-reduce
(distributed.reduce_summary; RCCL when N > 1).  Seeds are resident in HBM before the timed region; all outputs stay in HBM.

Other BASELINE configs, same contract, selected with --config:
    c3  configs[2]  single_N N=12 full_3d exponential 1 -> 3, 65 536 chains per GPU, full trace
    c4  configs[3]  measure_min_energy_vs_N Ns=3..20 x {random, latin, klarner}, linear 1 -> 3, board, n_runs = --chains per GPU
                    per cell (1 024: 8 192 on 8 GPUs), no trace (that driver discards histories, experiments.py:1061)
    c5  configs[4]  beta_start_end_pairs N=24 board sinusoidal, 16 pairs x (--chains per GPU: 1 024 = 8 192 on 8 GPUs) chains,
                    on-device per-step sums instead of histories (trace = REDUCED)
c4 / c5 run through the drivers' own engine (jobs.JobSet): every cell / pair sharded over the ranks, one packed all-reduce.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 --steps 3 --warmup 1        (starts its own ranks: the line below, as child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  The CPU oracle is used only for the `cpu_baseline` leg (rank 0, N=1, a bounded sample of
the same workload) and is never on the measured path.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_MOVE = 4.125  # one int32 energy_history entry + one accept bit
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2  # 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction
METRIC = "Metropolis moves/sec (whole node) + min-energy-reached, N=12 board MCMC"
KERNEL_SOURCE = os.path.join(ROOT, "monte-carlo-collective_amd", "csrc", "mcq_hip.hip")
C5_PAIRS = [(s, e) for s in (0.1, 0.5, 1.0, 2.0) for e in (2.0, 3.0, 5.0, 8.0)]  # BASELINE fixes only the count


def kernel_sha():
    with open(KERNEL_SOURCE, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def measured_traffic(key):
    """PMC figures of this exact workload and THIS kernel source (separate rocprofv3 --pmc passes, tools/pmc_refresh.sh).
    Entries carry the sha256 of csrc/mcq_hip.hip they were measured on; a stale entry is not evidence and yields None."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            e = json.load(f).get(key)
    except (OSError, ValueError):
        return None, "profiles/hbm_traffic.json unreadable"
    if not e:
        return None, f"no PMC entry for {key}"
    if e.get("kernel_sha256") != kernel_sha():
        return None, "PMC entry was measured on another version of csrc/mcq_hip.hip"
    return e, None


def usable_cpus():
    """Host threads this process may use: the affinity mask, capped by a cgroup CPU quota where one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def self_launch(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as fresh child processes (one per GPU, the same command
    line under torch.distributed.run), let rank 0's JSON line through on stdout and return the children's status.  This process
    has not imported torch or touched the GPU, and it never execs: the ranks are children."""
    import socket
    import subprocess

    with socket.socket() as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def init_dist(args, torch):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU")
    # one rank per GPU; MCQ_BENCH_BACKEND=gloo (testing only) lets several ranks share a GPU and reduces on the host
    backend = os.environ.get("MCQ_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > max(1, torch.cuda.device_count()):
        raise SystemExit(f"--gpus {world} but {torch.cuda.device_count()} GPU(s) are visible: RCCL runs one rank per GPU "
                         "(MCQ_BENCH_BACKEND=gloo lets ranks share a GPU and reduces on the host: testing only)")
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("MCQ_BENCH_FORCE_DIST") == "1":  # the override exercises the RCCL path on a one-GPU box
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend)
    red_dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    if dist is not None:  # communicator set-up (lazy in RCCL) must never fall into the timed region, even with --warmup 0
        w = torch.zeros(8, dtype=torch.int64, device=red_dev)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
    return dist, rank, world, red_dev


def timed_region(args, torch, dist, red_dev, step_fn):
    """W untimed + K timed steps bracketed by barrier + synchronize on both sides; returns max-over-ranks seconds."""
    for _ in range(args.warmup):
        step_fn(False)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_fn(True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    return elapsed


def per_rank_table(torch, dist, rank, world, red_dev, values):
    """What every rank measured, side by side (for reading an N > 1 run without a second lease): `values` is this rank's
    {name: milliseconds}; one SUM all-reduce of a [world x len(values)] tensor in which a rank fills its own row, AFTER the timed
    region.  Returns {name: [value of rank 0, 1, ...]} on every rank."""
    names = sorted(values)
    t = torch.zeros(world, len(names), dtype=torch.float64, device=red_dev)
    t[rank] = torch.tensor([float(values[k]) for k in names], dtype=torch.float64, device=red_dev)
    if dist is not None and world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = t.cpu().tolist()
    return {k: [round(row[i], 4) for row in t] for i, k in enumerate(names)}


def roofline_from_pmc(line, pmc, why, kernel_ms, algo_bytes, local_moves):
    """Fill roofline.traffic / traffic_rate / valu_issue from a PMC entry of this workload and kernel source (or say why not)."""
    rl = line["roofline"]
    if not pmc:
        rl["traffic_note"] = why
        return
    rl["traffic"] = pmc["bytes_per_launch"]
    if algo_bytes:
        rl["traffic_ratio"] = pmc["bytes_per_launch"] / algo_bytes  # measured fabric bytes over algorithmic bytes
    # the same kernel against the same peak on the bytes it really moves (PMC count per launch over the live kernel time)
    rl["traffic_rate"] = pmc["bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
    rl["traffic_frac"] = rl["traffic_rate"] / HBM_PEAK_GBS
    rl["traffic_source"] = {k: pmc[k] for k in ("kernel_sha256", "commit", "read_bytes", "write_bytes", "l2_hit_rate", "sweep_launches") if k in pmc}
    if pmc.get("valu_insts_per_launch"):
        # the on-chip view of the same kernel: wave64 VALU instructions (PMC count of this workload) over the live kernel
        # time, against one instruction per 2 cycles per SIMD (MI355X_MICROARCH.md: SIMD-32) on 1024 SIMDs at 2.4 GHz
        ach = pmc["valu_insts_per_launch"] / (kernel_ms * 1e-3) / 1e9
        rl["valu_issue"] = {"achieved": ach, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
                            "frac": ach / VALU_PEAK_GINST, "per_move": pmc["valu_insts_per_launch"] / max(1, local_moves)}


def bench_single(args, torch, mcq_amd, dist, rank, world, red_dev):
    """configs[1] / configs[2]: one DeviceRun per rank."""
    import numpy as np

    abi, dm = mcq_amd.abi, mcq_amd.distributed
    if args.config == "c3":
        args.mcmc_type, args.schedule = "full_3d", "exponential_annealing"
    if args.schedule == "constant":
        sp = {"type": "constant", "beta_const": 5.0}
    else:
        sp = {"type": args.schedule, "beta_start": 1.0, "beta_end": 3.0}
    trace = {"i32": True, "none": False, "reduced": "reduced"}["none" if args.no_trace else args.trace]
    base_seed = 42
    p = abi.make_params(args.N, args.n_steps, "random", sp, args.chains, mcmc_type=args.mcmc_type,
                        early_stop_patience=args.patience, trace=trace, lanes_per_chain=args.lanes, rng=args.rng)
    def with_exchange(q):  # replica exchange between the chains of a ladder: NOT a mode of the reference, evidence only
        if args.exchange:
            if args.replicas not in (2, 4, 8, 16):
                raise SystemExit("--replicas must be 2, 4, 8 or 16")
            lo, hi = (float(x) for x in args.ladder.split(","))
            if not (0.0 < lo <= hi < float("inf")):
                raise SystemExit("--ladder LO,HI: finite positive multipliers, LO <= HI")
            abi.set_exchange(q, args.exchange, lo * (hi / lo) ** (np.arange(args.replicas) / (args.replicas - 1)))
        return q

    with_exchange(p)
    # chains are sharded by contiguous global index; the seed of a chain does not depend on the GPU count
    seeds = abi.seeds_for(base_seed + rank * args.chains, args.chains)
    run = mcq_amd._lib.DeviceRun(p, seeds, trace=trace, states=not args.no_states)
    stream = torch.cuda.current_stream()
    init_ms, sweep_ms, reduce_ms, last = [], [], [], {}

    def step(timed):
        if timed:
            i_ms, s_ms = run.launch_timed(stream)  # HIP events on the launch stream around each kernel
            init_ms.append(i_ms)
            sweep_ms.append(s_ms)
        else:
            run.launch(stream)
        t_r = time.perf_counter()
        src = run.t if red_dev.type == "cuda" else {k: run.t[k].cpu() for k in ("best_energy", "steps_to_best", "n_accepted", "steps_executed")}
        last["summary"] = dm.reduce_summary(src, dist=dist, device=red_dev)  # ONE packed all-reduce
        if timed:
            reduce_ms.append((time.perf_counter() - t_r) * 1e3)  # pack kernels + the collective + the copy of 14 words to the host (it waits for the sweep when that was not timed)

    elapsed = timed_region(args, torch, dist, red_dev, step)
    sm = last["summary"]
    total_chains = args.chains * world
    value = sm["proposed"] * args.steps / elapsed
    sweep_avg_ms = sum(sweep_ms) / len(sweep_ms)
    local_moves = int(run.t["steps_executed"].sum().item())
    algo_bytes = ALGO_BYTES_PER_MOVE * local_moves if trace is True else 0.0
    achieved = algo_bytes / (sweep_avg_ms * 1e-3) / 1e9

    key = f"{args.mcmc_type}_N{args.N}_c{args.chains}_s{args.n_steps}" + ("" if args.rng == "mt19937" else f"_{args.rng}") + ("" if trace is True else "_notrace" if trace is False else "_reduced") \
        + ("" if not args.exchange else f"_x{args.exchange}")
    pmc, why = measured_traffic(key)
    line = {
        "metric": METRIC,
        "value": value,
        "unit": "moves/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32+f64",
        "data": "synthetic (seeded random initial boards, seeds 42 + chain index; "
                + ("MT19937 NumPy-legacy stream)" if args.rng == "mt19937" else "Philox-4x32-10 stream: NOT the reference's)"),
        "config": {
            "workload": f"single_N N={args.N} mcmc_type={args.mcmc_type} init=random {args.schedule} "
                        f"{sp.get('beta_start', sp.get('beta_const'))}->{sp.get('beta_end', '')} "
                        + ("" if args.patience is None else f"early_stop_patience={args.patience} ")
                        + f"n_runs={args.chains}/GPU n_steps={args.n_steps} trace={'i32' if trace is True else 'none' if trace is False else 'reduced'}"
                        + ("" if args.rng == "mt19937" else f" rng={args.rng} (NOT the reference's stream)")
                        + ("" if not args.exchange else f" replica exchange every {args.exchange} steps over {args.replicas} rungs, beta x {args.ladder} (NOT a mode of the reference)"),
            "chains_total": total_chains,
            "lanes_per_chain": mcq_amd._lib.effective_lanes(run.p),
            "parallelism": f"chains sharded over {world} GPU(s), no data-path collective; one packed SUM all-reduce for the summary",
            "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
        },
        "min_energy": sm["min_best"],
        "mean_best_energy": sm["mean_best"],
        "acceptance_rate": sm["acceptance_rate"],
        "kernel_ms": {"init": sum(init_ms) / len(init_ms), "sweep": sweep_avg_ms},
        "per_rank": per_rank_table(torch, dist, rank, world, red_dev, {"init_ms": sum(init_ms) / len(init_ms), "sweep_ms": sweep_avg_ms,
                                                                       "reduce_ms": sum(reduce_ms) / len(reduce_ms), "step_ms": elapsed / args.steps * 1e3}),
        "allreduce": {"per_step": 1, "payload_bytes": 8 * dm.layout_for([(0, 0)], world, per_chain=False, stats=False)[1],
                      "backend": "none (one process)" if dist is None else dist.get_backend()},
        **({"exchanges_per_chain": float(run.t["n_exchanges"].double().mean().item())} if args.exchange else {}),
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "kernel": "mcq_sweep_kernel",
            "note": "algorithmic bytes = 4.125 B/move (int32 trace entry + accept bit); on the chip the sweep is bound by instruction issue, "
                    "and the MT19937 state stream it carries moves ~25x those bytes (traffic_frac): DESIGN.md 4.3",
        },
    }
    roofline_from_pmc(line, pmc, why, sweep_avg_ms, algo_bytes, local_moves)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle  # checker / baseline only; never on the measured path

        threads = args.cpu_threads or usable_cpus()
        n_cpu = min(args.cpu_chains, args.chains)
        ctrace = trace is True
        pc = abi.make_params(args.N, args.n_steps, "random", sp, n_cpu, mcmc_type=args.mcmc_type, trace=ctrace, rng=args.rng,
                             early_stop_patience=args.patience)
        with_exchange(pc)
        t1 = time.perf_counter()
        cres = oracle.run(pc, seeds[:n_cpu], trace=ctrace, states=True, n_threads=threads)
        dt = time.perf_counter() - t1
        # the whole sample against the GPU: every integer output, trace and states included
        same = True
        for k in ("hist_len", "initial_energy", "best_energy", "final_energy", "steps_to_best", "n_accepted", "best_state", "final_state"):
            if k in run.t:
                same = same and bool(np.array_equal(run.t[k][:n_cpu].cpu().numpy(), cres[k]))
        if ctrace:
            same = same and bool(np.array_equal(run.t["energy_hist"][:n_cpu, : args.n_steps + 1].cpu().numpy(), cres["energy_hist"][:, : args.n_steps + 1]))
            same = same and bool(np.array_equal(run.t["accept_bits"][:n_cpu].cpu().numpy().view(np.uint64), cres["accept_bits"]))
        del cres
        n_fast = min(8 * n_cpu, args.chains)
        pf = abi.make_params(args.N, args.n_steps, "random", sp, n_fast, mcmc_type=args.mcmc_type, trace=False, rng=args.rng,
                             early_stop_patience=args.patience)
        with_exchange(pf)
        t2 = time.perf_counter()
        fres = oracle.run(pf, seeds[:n_fast], trace=False, states=False, n_threads=threads, fast=True)
        dt_fast = time.perf_counter() - t2
        same_fast = bool(np.array_equal(run.t["best_energy"][:n_fast].cpu().numpy(), fres["best_energy"])
                         and np.array_equal(run.t["n_accepted"][:n_fast].cpu().numpy(), fres["n_accepted"]))
        line["cpu_baseline"] = {
            "value": float(n_cpu * args.n_steps) / dt,
            "unit": "moves/s",
            "cores": threads,
            "host_cpus": os.cpu_count(),
            "kind": "port",
            "sample": f"first {n_cpu} chains of the same workload (same seeds, {args.n_steps} steps each, trace and states written) on the C "
                      f"restatement of the reference's O(N^2) scan, {threads} threads",
            "sample_matches_gpu": same,
            "compared": "energy_hist, accept_bits, best/final state and every per-chain scalar of the sample, bit for bit",
            "fast_port": {
                "value": float(fres["steps_executed"].sum()) / dt_fast,
                "unit": "moves/s",
                "cores": threads,
                "algorithm": "O(1) dE from per-line occupancy counters (oracle mcq_oracle_run_fast: the best CPU formulation)",
                "sample": f"first {n_fast} chains, no trace",
                "sample_matches_gpu": same_fast,
            },
        }
    return line


def bench_jobs(args, torch, mcq_amd, dist, rank, world, red_dev):
    """configs[3] / configs[4] through jobs.JobSet (what the drivers run on)."""
    jb = mcq_amd.jobs
    chains = args.chains * world  # n_runs of every cell / pair: weak scaling, args.chains per GPU
    n = args.n_steps
    if args.config == "c4":
        jobs = []
        sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
        for init in ("random", "latin", "klarner"):
            off = sum(ord(c) for c in init) % 1000
            for idx, N in enumerate(range(3, 21)):
                jobs.append(jb.make_job(N, n, init, sp, chains, 42 + 10 * idx + off, "board", None))
        want, what = "summary", f"measure_min_energy_vs_N Ns=3..20 x [random,latin,klarner] board linear 1.0->3.0 n_runs={args.chains}/GPU per cell n_steps={n} trace=none"
    else:
        jobs = [jb.make_job(24, n, "random", {"type": "sinusoidal_annealing", "beta_start": s, "beta_end": e}, chains, 42 + 1000 * i, "board", None)
                for i, (s, e) in enumerate(C5_PAIRS)]
        want, what = "stats", f"beta_start_end_pairs N=24 board sinusoidal 16 pairs x {args.chains}/GPU chains n_steps={n} trace=reduced (per-step sums on device)"
    js = jb.JobSet(jobs, want=want, dist=dist, lanes_per_chain=args.lanes, rng=args.rng)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev.append(torch.cuda.Event(enable_timing=True))
    kernel_ms, sweeps_ms, reduce_ms, last = [], [], [], {}

    def step(timed):
        cur = torch.cuda.current_stream()
        if timed:
            ev[0].record(cur)
        js.launch()                # every launch forks from the current stream ...
        if timed:                  # ... (an event where they have all joined it again: the sweeps alone) ...
            for la in js.launches:
                cur.wait_stream(la.stream)
            ev[2].record(cur)
        t_r = time.perf_counter()
        last["res"] = js.reduce()  # ... joins it again, then ONE packed all-reduce
        if timed:
            ev[1].record(cur)
            ev[1].synchronize()
            reduce_ms.append((time.perf_counter() - t_r) * 1e3)  # host time of reduce(): it waits for the sweeps, so device time of the reduce = all - sweeps
            kernel_ms.append(ev[0].elapsed_time(ev[1]))
            sweeps_ms.append(ev[0].elapsed_time(ev[2]))

    elapsed = timed_region(args, torch, dist, red_dev, step)
    res = last["res"]
    proposed = sum(r["summary"]["proposed"] for r in res)
    accepted = sum(r["summary"]["accepted"] for r in res)
    mins = [r["summary"]["min_best"] for r in res]
    all_ms, sw_ms = sum(kernel_ms) / len(kernel_ms), sum(sweeps_ms) / len(sweeps_ms)
    local_moves = sum(int(la.run.t["steps_executed"].sum().item()) for la in js.launches)
    key = f"{args.config}_c{args.chains}_s{n}" + ("" if args.rng == "mt19937" else f"_{args.rng}")
    pmc, why = measured_traffic(key)
    line = {
        "metric": METRIC, "value": proposed * args.steps / elapsed, "unit": "moves/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32+f64", "data": "synthetic (seeded initial boards, the reference's seed derivations; MT19937 NumPy-legacy stream)",
        "config": {"workload": what, "chains_total": chains * len(jobs), "launches_per_rank": len(js.launches),
                   "lanes_per_chain": {f"N={int(la.run.p.N)}": mcq_amd._lib.effective_lanes(la.run.p) for la in js.launches},
                   "parallelism": f"every cell / pair sharded over {world} GPU(s); one packed SUM all-reduce of {js.total_words} int64 words",
                   "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                   "cu_partition": getattr(js, "cu_partition", None), "pacing_across_launches": getattr(js, "pacing", None)},  # compute units per launch (in launch order) when the job list gives its launches CUs of their own
        "min_energy": min(m for m in mins if m is not None),
        "min_energy_per_job": mins,
        "acceptance_rate": accepted / max(1, proposed),
        "kernel_ms": {"all_launches": all_ms, "sweeps": sw_ms, "reduce_on_device": all_ms - sw_ms},
        "per_rank": per_rank_table(torch, dist, rank, world, red_dev, {"sweeps_ms": sw_ms, "launch_and_reduce_ms": all_ms, "reduce_host_ms": sum(reduce_ms) / len(reduce_ms),
                                                                       "step_ms": elapsed / args.steps * 1e3}),
        "allreduce": {"per_step": 1, "payload_bytes": 8 * js.total_words, "backend": "none (one process)" if dist is None else dist.get_backend()},
        "roofline": {"bound": "hbm", "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None,
                     "kernel": "mcq_sweep_kernel (one per launch)",
                     "note": "no per-chain trace leaves the kernel in this config (algorithmic HBM bytes ~ 0 per move, so frac = 0 by definition): the "
                             "bound is instruction issue / latency -- valu_issue (PMC: wave64 VALU instructions of the step's sweep launches over "
                             "their live time) is the figure to read; traffic = the MT19937 state stream, see DESIGN.md"},
    }
    roofline_from_pmc(line, pmc, why, sw_ms, 0.0, local_moves)
    if want == "stats":
        r0 = res[0]
        mean, std = jb.mean_std_from_sums(r0["step_sum"], r0["step_sumsq"], r0["step_count"])
        line["first_pair_mean_energy_at"] = {str(e): float(mean[e]) for e in (0, n // 4, n // 2, n)}
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=int(os.environ.get("WORLD_SIZE", "1")))
    ap.add_argument("--steps", type=int, default=3, help="timed launches (K)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed launches (W)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"], help="BASELINE.json config (c2 = configs[1], the headline)")
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (c2/c3: 65536; c4/c5: per cell / pair, 1024)")
    ap.add_argument("--n-steps", type=int, default=100000, help="Metropolis steps per chain per launch")
    ap.add_argument("--N", type=int, default=12)
    ap.add_argument("--mcmc-type", default="board", choices=["board", "full_3d"])
    ap.add_argument("--schedule", default="linear_annealing")
    ap.add_argument("--lanes", type=int, default=0, help="lanes of a wavefront per chain (4, 8 or 16; 0 = library default)")
    ap.add_argument("--rng", default="mt19937", choices=["mt19937", "philox"],
                    help="mt19937 = NumPy's stream (reference-identical, the bench default); philox = counter-based fast mode (evidence only)")
    ap.add_argument("--patience", type=int, default=None, help="c2 / c3: early_stop_patience (default None = disabled, like config.yaml)")
    ap.add_argument("--no-trace", action="store_true", help="same as --trace none")
    ap.add_argument("--trace", default="i32", choices=["i32", "none", "reduced"],
                    help="c2 / c3: full int32 energy trace (the headline), no trace, or per-step sums accumulated on the device")
    ap.add_argument("--exchange", type=int, default=0, help="c2 / c3: replica exchange every K steps (0 = off; NOT a mode of the reference)")
    ap.add_argument("--replicas", type=int, default=16, help="rungs of a ladder (2, 4, 8, 16)")
    ap.add_argument("--ladder", default="0.7,1.4", help="beta multipliers of the lowest and highest rung (geometric in between)")
    ap.add_argument("--no-states", action="store_true", help="c2 / c3: no best_state / final_state outputs (what the drivers' engine runs with)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-chains", type=int, default=1024)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the cpu_baseline leg (0 = every CPU this process may use)")
    args = ap.parse_args()
    if args.chains is None:
        args.chains = 65536 if args.config in ("c2", "c3") else 1024
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import torch

    import mcq_amd

    if args.config == "c4":  # 18 launches side by side, one HIP stream each: more hardware queues than the runtime's 4, asked for before
        mcq_amd._lib.ensure_hw_queues(18)  # the first GPU call of this process (the variable is read when the HIP runtime initialises)
    dist, rank, world, red_dev = init_dist(args, torch)
    fn = bench_single if args.config in ("c2", "c3") else bench_jobs
    line = fn(args, torch, mcq_amd, dist, rank, world, red_dev)
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

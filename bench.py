#!/usr/bin/env python3
"""bench.py -- Metropolis moves/sec (whole job) + min-energy-reached on BASELINE.json configs[1]:

    single_N, N=12, mcmc_type=board, init=random, linear_annealing 1.0 -> 3.0,
    65 536 chains per GPU x 100 000 steps, seeds 42 + chain index, full int32 energy trace.

A bench "step" is one pass of the hot path over the whole batch: one launch of the init kernel
and one launch of the sweep kernel that runs every chain of this rank for n_steps proposals and
writes the full energy_history / accept-bit trace to HBM, followed by the summary reduce
(RCCL all-reduce MIN / SUM when N > 1).  Seeds are resident in HBM before the timed region; all
outputs stay in HBM.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  The CPU oracle is used only for the `cpu_baseline` leg (rank 0,
N=1, a bounded sample of the same workload) and is never on the measured path.
"""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=int(os.environ.get("WORLD_SIZE", "1")))
    ap.add_argument("--steps", type=int, default=3, help="timed launches (K)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed launches (W)")
    ap.add_argument("--chains", type=int, default=65536, help="chains per GPU")
    ap.add_argument("--n-steps", type=int, default=100000, help="Metropolis steps per chain per launch")
    ap.add_argument("--N", type=int, default=12)
    ap.add_argument("--mcmc-type", default="board", choices=["board", "full_3d"])
    ap.add_argument("--schedule", default="linear_annealing")
    ap.add_argument("--lanes", type=int, default=0, help="lanes of a wavefront per chain (4, 8 or 16; 0 = library default)")
    ap.add_argument("--rng", default="mt19937", choices=["mt19937", "philox"],
                    help="mt19937 = NumPy's stream (reference-identical, the bench default); philox = counter-based fast mode (evidence only)")
    ap.add_argument("--no-trace", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-chains", type=int, default=1024)
    args = ap.parse_args()

    import numpy as np
    import torch

    import mcq_amd

    abi = mcq_amd.abi
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    # one rank per GPU; MCQ_BENCH_BACKEND=gloo (testing only) lets several ranks share a GPU and reduces on the host
    backend = os.environ.get("MCQ_BENCH_BACKEND", "nccl")
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
    red_dev = "cuda" if backend == "nccl" else "cpu"

    if args.schedule == "constant":
        sp = {"type": "constant", "beta_const": 5.0}
    else:
        sp = {"type": args.schedule, "beta_start": 1.0, "beta_end": 3.0}
    trace = not args.no_trace
    base_seed = 42
    p = abi.make_params(args.N, args.n_steps, "random", sp, args.chains, mcmc_type=args.mcmc_type,
                        early_stop_patience=None, trace=trace, lanes_per_chain=args.lanes, rng=args.rng)
    # chains are sharded by contiguous global index; the seed of a chain does not depend on the GPU count
    seeds = abi.seeds_for(base_seed + rank * args.chains, args.chains)
    run = mcq_amd._lib.DeviceRun(p, seeds, trace=trace, states=False)
    stream = torch.cuda.current_stream()

    def summary():
        """node-level summary: min best energy, total accepted, total proposed (one small all-reduce each)."""
        mn = run.t["best_energy"].min().to(torch.int64).reshape(1)
        sm = torch.stack([run.t["n_accepted"].sum(), run.t["steps_executed"].sum(), run.t["best_energy"].to(torch.int64).sum()])
        if dist is not None:
            mn, sm = mn.to(red_dev), sm.to(red_dev)
            dist.all_reduce(mn, op=dist.ReduceOp.MIN)
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        return mn, sm

    if dist is not None:  # communicator set-up (lazy in RCCL) must never fall into the timed region, even with --warmup 0
        w = torch.zeros(1, dtype=torch.int64, device=red_dev)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        dist.all_reduce(w, op=dist.ReduceOp.MIN)
    for _ in range(args.warmup):
        run.launch(stream)
        summary()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()

    init_ms, sweep_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        i_ms, s_ms = run.launch_timed(stream)  # HIP events on the launch stream around each kernel
        init_ms.append(i_ms)
        sweep_ms.append(s_ms)
        mn, sm = summary()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    min_energy = int(mn.item())
    accepted, proposed, best_sum = (int(v) for v in sm.tolist())  # whole job, one launch
    total_chains = args.chains * world
    moves_per_launch = proposed
    value = moves_per_launch * args.steps / elapsed
    sweep_avg_ms = sum(sweep_ms) / len(sweep_ms)
    local_moves = int(run.t["steps_executed"].sum().item())
    algo_bytes = ALGO_BYTES_PER_MOVE * local_moves if trace else 0.0
    achieved = algo_bytes / (sweep_avg_ms * 1e-3) / 1e9

    # PMC figures of this exact workload collected in separate rocprofv3 --pmc passes (tools/pmc_collect.sh), per launch
    traffic, valu_insts = None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                tj = json.load(f)
            key = f"{args.mcmc_type}_N{args.N}_c{args.chains}_s{args.n_steps}" + ("" if args.rng == "mt19937" else f"_{args.rng}")
            traffic = tj.get(key, {}).get("bytes_per_launch")
            valu_insts = tj.get(key, {}).get("valu_insts_per_launch")
        except (OSError, ValueError):
            traffic, valu_insts = None, None

    line = {
        "metric": "Metropolis moves/sec (whole node) + min-energy-reached, N=12 board MCMC",
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
        "data": "synthetic (seeded random initial boards, seeds 42 + chain index; MT19937 NumPy-legacy stream)",
        "config": {
            "workload": f"single_N N={args.N} mcmc_type={args.mcmc_type} init=random {args.schedule} "
                        f"{sp.get('beta_start', sp.get('beta_const'))}->{sp.get('beta_end', '')} "
                        f"n_runs={args.chains}/GPU n_steps={args.n_steps} trace={'i32' if trace else 'none'}"
                        + ("" if args.rng == "mt19937" else f" rng={args.rng} (NOT the reference's stream)"),
            "chains_total": total_chains,
            "lanes_per_chain": int(run.p.lanes_per_chain) or int(mcq_amd._lib.lib().mcq_default_lanes_n(run.p.mode, run.p.N)),
            "parallelism": f"chains sharded over {world} GPU(s), no data-path collective; summary all-reduce (MIN/SUM)",
        },
        "min_energy": min_energy,
        "mean_best_energy": best_sum / total_chains,
        "acceptance_rate": accepted / max(1, proposed),
        "kernel_ms": {"init": sum(init_ms) / len(init_ms), "sweep": sweep_avg_ms},
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "kernel": "mcq_sweep_kernel",
            "note": "algorithmic bytes = 4.125 B/move (int32 trace entry + accept bit); the sweep is issue/latency bound, see DESIGN.md",
        },
    }
    if valu_insts:
        # the on-chip view of the same kernel: wave64 VALU instructions (PMC count of this workload) over the live kernel
        # time, against one instruction per 2 cycles per SIMD (MI355X_MICROARCH.md: SIMD-32) on 1024 SIMDs at 2.4 GHz
        peak = VALU_PEAK_GINST
        ach = valu_insts / (sweep_avg_ms * 1e-3) / 1e9
        line["roofline"]["valu_issue"] = {"achieved": ach, "peak": peak, "unit": "G wave-instructions/s", "frac": ach / peak,
                                          "per_move": valu_insts / max(1, local_moves)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle  # checker / baseline only; never on the measured path

        n_cpu = min(args.cpu_chains, args.chains)
        threads = min(16, os.cpu_count() or 1)
        pc = abi.make_params(args.N, args.n_steps, "random", sp, n_cpu, mcmc_type=args.mcmc_type, trace=False, rng=args.rng)
        t1 = time.perf_counter()
        cres = oracle.run(pc, seeds[:n_cpu], trace=False, states=False, n_threads=threads)
        dt = time.perf_counter() - t1
        gbest = run.t["best_energy"][:n_cpu].cpu().numpy()
        line["cpu_baseline"] = {
            "value": float(cres["steps_executed"].sum()) / dt,
            "unit": "moves/s",
            "cores": threads,
            "kind": "port",
            "sample": f"first {n_cpu} chains of the same workload (same seeds, {args.n_steps} steps each) on the C oracle, {threads} threads",
            "sample_matches_gpu": bool(np.array_equal(gbest, cres["best_energy"])),
        }

    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

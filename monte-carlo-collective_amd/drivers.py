"""Experiment drivers on top of the sweep boundary (the callers of run_experiment in the reference).

Same names, arguments, seed derivations, labels and result dicts as the reference:
    run_beta_start_end_pairs   experiments.py:741-846   pair_seed = base_seed + idx * 1000 (791)
    run_compare_beta_end       experiments.py:943-1029  second N at base_seed + 10000 (1000)
    measure_min_energy_vs_N    experiments.py:1031-1201 seed = base_seed + 10*idx + sum(ord(c)) % 1000 (1060-1067)
    main / load_config         experiments.py:1204-1391 config.yaml dispatch

Every driver turns its loop into a job list and hands it to jobs.JobSet: all jobs run device-resident and
concurrently, the chains of each job sharded over the ranks of `dist` (torch.distributed, one process per GPU), and ONE
packed all-reduce brings the per-pair / per-cell minima, counters, per-run best energies and -- where CSVs are written
-- the per-step integer sums to every rank.  Rank 0 writes the files.

Histories.  The reference returns every energy history to the caller (experiments.py:573, 843-846).  That is kept for
runs whose traces are small (`histories=True`, or "auto" below HISTORY_BYTES_AUTO in a single process); beyond that the
drivers run with trace = REDUCED: `all_histories[label]` is None and the per-step mean / std and binned acceptance come
from on-device integer sums ("energy_stats" / "acceptance" in the result), which is all the reference's plot functions
consume (experiments.py:593-608, 660-711).  BASELINE config 5 (16 pairs x 8 192 chains x 10^5 steps) would otherwise be
52 GB of trace.

Plotting is out of scope; with plot=True the drivers write the reference's CSV data products
(results/*.csv) and no PNGs.  The reference's run_compare_beta_end raises TypeError after all compute when plot=True
(experiments.py:1020-1021 vs 848); that bug is not reproduced.

Every driver takes `runner=`: a callable with run_chains' signature.  The default is the GPU path; tests inject the CPU
oracle to check the host logic without a GPU.
"""
import os

import numpy as np

from . import abi
from . import distributed as dm
from . import experiments as ex
from . import jobs as jb
from .jobs import acceptance_bins_from_steps, mean_std_from_sums  # noqa: F401  (re-exported: the CSV arithmetic)

HISTORY_BYTES_AUTO = 256 << 20  # "auto": full histories come back to the host below this many bytes of trace


def _runner_or_default(runner):
    return ex.run_chains if runner is None else runner


def _want(histories, jobs, dist, need_steps):
    """`histories` in (True, False, "auto") -> jobs.JobSet's `want`."""
    # a full trace row (abi.hist_stride_for(n_steps) entries: n_steps + 1 rounded up to 64) must stay below abi.MAX_HIST_STRIDE = 2^24
    # (include/mcq.h: hist_stride < 2^24); the reference takes any n_steps, so longer runs go through the on-device statistics instead
    # of failing inside the library
    fits = all(abi.hist_stride_for(j["n_steps"]) < abi.MAX_HIST_STRIDE for j in jobs)
    if histories == "auto":
        total = sum(j["n_runs"] * (j["n_steps"] + 1) * 4 for j in jobs)
        histories = dm.rank_world(dist)[1] == 1 and total <= HISTORY_BYTES_AUTO and fits
    if histories:
        if not fits:
            raise ValueError(f"full histories hold at most {abi.MAX_HIST_STRIDE - 65} steps per chain (rows of n_steps + 1 entries, rounded up to 64, below "
                             f"{abi.MAX_HIST_STRIDE}); use histories=False (on-device statistics) for longer runs")
        return "histories"
    return "stats" if need_steps else "summary"


def _is_writer(dist):
    return dm.rank_world(dist)[0] == 0


def _histories_of(res):
    return [res["energy_hist"][r, : int(res["hist_len"][r])] for r in range(len(res["hist_len"]))]


def energy_statistics(all_histories):
    """Per-step mean and (population) std over runs, as plot_energy_histories computes them (experiments.py:593-595).
    Ragged histories (early stop) cannot be stacked -- the reference fails there too; here it is an explicit error."""
    lens = {len(h) for h in all_histories}
    if len(lens) != 1:
        raise ValueError("histories have different lengths (early stop); per-step statistics are undefined")
    e = np.asarray(all_histories)
    return e.mean(axis=0), e.std(axis=0)


def acceptance_rates_from_bins(bin_accepted, bin_proposed):
    """Rates per bin from the on-device counts; empty bins are NaN like the reference (experiments.py:690-693)."""
    a = np.asarray(bin_accepted, dtype=np.float64)
    p = np.asarray(bin_proposed, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(p > 0, a / p, np.nan)


def write_energy_csv(all_histories, label, out_dir="results", stats=None):
    """results/{label}.csv with columns step, mean_energy, std_energy (experiments.py:600-608); `stats` = (mean, std) from
    the on-device sums when the histories never left the GPU."""
    mean, std = energy_statistics(all_histories) if stats is None else stats
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"{label}.csv")
    np.savetxt(path, np.column_stack([np.arange(len(mean)), mean, std]), delimiter=",", header="step,mean_energy,std_energy",
               comments="", fmt=["%d", "%.18g", "%.18g"])
    return path


def acceptance_rates_binned(accepted_steps_runs, rejected_steps_runs, n_steps, n_bins=100):
    """Binned acceptance rate over all runs, exactly as plot_acceptance_rates_binned (experiments.py:660-695):
    n_bins equal bins over [0, n_steps], last bin closed on the right, NaN for empty bins."""
    edges = np.linspace(0, n_steps, n_bins + 1)
    centers = (edges[:-1] + edges[1:]) / 2
    edges[-1] = n_steps
    acc = np.concatenate([np.asarray(a) for a in accepted_steps_runs]) if len(accepted_steps_runs) else np.array([])
    rej = np.concatenate([np.asarray(r) for r in rejected_steps_runs]) if len(rejected_steps_runs) else np.array([])
    rates = []
    for b in range(n_bins):
        lo, hi = edges[b], edges[b + 1]
        if b == n_bins - 1:
            a = np.sum((acc >= lo) & (acc <= hi))
            r = np.sum((rej >= lo) & (rej <= hi))
        else:
            a = np.sum((acc >= lo) & (acc < hi))
            r = np.sum((rej >= lo) & (rej < hi))
        rates.append(a / (a + r) if a + r > 0 else np.nan)
    return centers, np.array(rates)


def write_acceptance_csv(centers, rates, label, out_dir="results"):
    """results/acceptance_rates_{label}.csv (experiments.py:704-711)."""
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"acceptance_rates_{label}.csv")
    np.savetxt(path, np.column_stack([centers, rates]), delimiter=",", header="bin_center,acceptance_rate", comments="")
    return path


def run_beta_start_end_pairs(N, n_steps, beta_start_ends, annealing_type="linear_annealing", init_mode="random", n_runs=5,
                             base_seed=0, verbose=True, plot=True, out_path=None, out_path_acceptance=None,
                             mcmc_type="full_3d", early_stop_patience=100000, runner=None, dist=None, histories="auto",
                             csv_prefix=""):
    """experiments.py:741-846.  Returns {"all_histories": {label: [...] or None}, "all_best_energies": {label: [...]}} with
    label f"beta: {beta_start}->{beta_end}" (806), plus "min_energies" {label: min}, and -- when the histories stay on the
    device -- "energy_stats" {label: (mean, std)} and "acceptance" {label: (centers, rates)}.  The pairs run as one launch
    with one schedule set per pair; with `dist` every pair's chains are sharded over the ranks."""
    jobs, labels = [], []
    for idx, (beta_start, beta_end) in enumerate(beta_start_ends):
        sp = {"type": annealing_type, "beta_start": beta_start, "beta_end": beta_end}
        ex.build_schedule_from_params(annealing_type, n_steps, beta_start=beta_start, beta_end=beta_end)  # the reference's ValueErrors
        jobs.append(jb.make_job(N, n_steps, init_mode, sp, n_runs, base_seed + idx * 1000, mcmc_type, early_stop_patience))
        labels.append(f"beta: {beta_start}->{beta_end}")
    want = _want(histories, jobs, dist, need_steps=plot)
    results = jb.JobSet(jobs, want=want, dist=dist, runner=runner).run()
    all_histories, all_best, mins, stats, acceptance = {}, {}, {}, {}, {}
    for label, res in zip(labels, results):
        all_best[label] = [int(b) for b in res["best_energy"]]
        mins[label] = res["summary"]["min_best"]
        if want == "histories":
            all_histories[label] = _histories_of(res)
            if plot and out_path_acceptance is not None:
                steps = [ex.accepted_rejected_steps(res, r) for r in range(n_runs)]
                acceptance[label] = acceptance_rates_binned([s[0] for s in steps], [s[1] for s in steps], n_steps, n_bins=100)
        else:
            all_histories[label] = None
            if want == "stats":
                stats[label] = mean_std_from_sums(res["step_sum"], res["step_sumsq"], res["step_count"])
                acceptance[label] = acceptance_bins_from_steps(res["step_accepted"], res["step_count"], res["step_stopped"], n_steps, n_bins=100)[:2]
        if verbose:
            for e in all_best[label]:
                print(e)
            print(np.mean(all_best[label]))
    if plot and _is_writer(dist):
        for label in labels:
            write_energy_csv(all_histories[label], csv_prefix + label, stats=stats.get(label))
        if out_path_acceptance is not None:
            for label in labels:
                write_acceptance_csv(*acceptance[label], csv_prefix + label)
    out = {"all_histories": all_histories, "all_best_energies": all_best, "min_energies": mins}
    if stats:
        out["energy_stats"] = stats
    if acceptance:
        out["acceptance"] = acceptance
    return out


def run_compare_beta_end(Ns, n_steps, beta_start_ends, annealing_type="linear_annealing", init_mode="random", n_runs=5,
                         base_seed=0, verbose=True, plot=True, out_path=None, mcmc_type="full_3d",
                         early_stop_patience=100000, runner=None, dist=None, histories="auto"):
    """experiments.py:943-1029: the pair experiment for two board sizes, the second at base_seed + 10000."""
    if len(Ns) != 2:
        raise ValueError("Ns must contain exactly 2 values")
    kw = dict(n_steps=n_steps, beta_start_ends=beta_start_ends, annealing_type=annealing_type, init_mode=init_mode,
              n_runs=n_runs, verbose=verbose, plot=plot, out_path=None, out_path_acceptance=None, mcmc_type=mcmc_type,
              early_stop_patience=early_stop_patience, runner=runner, dist=dist, histories=histories)
    r1 = run_beta_start_end_pairs(N=Ns[0], base_seed=base_seed, csv_prefix=f"N{Ns[0]}_", **kw)
    r2 = run_beta_start_end_pairs(N=Ns[1], base_seed=base_seed + 10000, csv_prefix=f"N{Ns[1]}_", **kw)
    return {"N1": Ns[0], "N2": Ns[1], "result_N1": r1, "result_N2": r2}


def measure_min_energy_vs_N(Ns, n_steps, beta_schedule, schedule_params=None, init_modes=["random"], n_runs=5, base_seed=100,
                            verbose=True, plot=True, out_path=None, mcmc_type="full_3d", early_stop_patience=100000,
                            runner=None, dist=None):
    """experiments.py:1031-1201.  Histories are discarded by this experiment (1061), so no trace is produced: every
    (init_mode, N) cell is a launch of its own stream, all cells run concurrently, each cell's chains sharded over `dist`."""
    if isinstance(init_modes, str):
        init_modes = [init_modes]
    if schedule_params is None:
        schedule_params = getattr(beta_schedule, "params", None)
    jobs = []
    for init_mode in init_modes:
        offset = sum(ord(c) for c in init_mode) % 1000
        for idx, N in enumerate(Ns):
            jobs.append(jb.make_job(N, n_steps, init_mode, schedule_params, n_runs, base_seed + 10 * idx + offset, mcmc_type, early_stop_patience))
    all_res = iter(jb.JobSet(jobs, want="summary", dist=dist, runner=runner).run())
    results = {}
    for init_mode in init_modes:
        all_min, all_stb = [], []
        for _N in Ns:
            res = next(all_res)
            all_min.append(np.array([int(b) for b in res["best_energy"]]))
            all_stb.append(np.array([int(s) for s in res["steps_to_best"]]))
            if verbose:
                print(all_min[-1].mean())
        results[init_mode] = {
            "mean_min_energies": np.array([m.mean() for m in all_min]),
            "std_min_energies": np.array([m.std() for m in all_min]),
            "all_min_energies": all_min,
            "mean_steps_to_best": np.array([s.mean() for s in all_stb]),
            "std_steps_to_best": np.array([s.std() for s in all_stb]),
            "all_steps_to_best": all_stb,
        }
    if plot and _is_writer(dist):
        os.makedirs("results", exist_ok=True)
        for init_mode in init_modes:
            r = results[init_mode]
            np.savetxt(f"results/min_energy_vs_N_{init_mode}.csv",
                       np.column_stack([np.asarray(Ns), r["mean_min_energies"], r["std_min_energies"]]), delimiter=",",
                       header=f"N,{init_mode}_mean_min_energy,{init_mode}_std_min_energy", comments="")
            np.savetxt(f"results/steps_to_best_vs_N_{init_mode}.csv",
                       np.column_stack([np.asarray(Ns), r["mean_steps_to_best"], r["std_steps_to_best"]]), delimiter=",",
                       header=f"N,{init_mode}_mean_steps_to_best,{init_mode}_std_steps_to_best", comments="")
    return {"Ns": Ns, "results": results}


def write_best_heights(heights, N, path):
    """competition.py:179-187: the best board as one `i,j,k` line per column (row-major)."""
    h = np.asarray(heights).reshape(N, N)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as f:
        for i in range(N):
            for j in range(N):
                f.write(f"{i},{j},{int(h[i, j])}\n")
    return path


def run_competition(N=15, n_runs=10, n_steps=100000, beta_start=1.0, beta_end=3.0, base_seed=42, init_mode="random",
                    out_dir="competition_results", runner=None, timestamp=None):
    """competition.py:143-187: board chains with linear annealing beta_start -> beta_end, seeds base_seed + r; the board
    of the run with the lowest best energy is written to {out_dir}/best_heights_{N}_{timestamp}.txt.
    Returns (best energy, heights, path)."""
    import time

    runner = _runner_or_default(runner)
    sp = {"type": "linear_annealing", "beta_start": beta_start, "beta_end": beta_end}
    try:
        res, _ = runner(N, n_steps, init_mode, sp, ex.abi.seeds_for(base_seed, n_runs), mcmc_type="board", early_stop_patience=None,
                        trace=False, states=True)
    except TypeError:  # injected runners without a `states` argument return the states anyway
        res, _ = runner(N, n_steps, init_mode, sp, ex.abi.seeds_for(base_seed, n_runs), mcmc_type="board", early_stop_patience=None, trace=False)
    r = int(np.argmin(res["best_energy"]))  # first run with the minimum, like min() over the runs in order
    heights = np.asarray(res["best_state"][r]).reshape(N, N)
    stamp = timestamp if timestamp is not None else time.strftime("%Y%m%d_%H%M%S")
    path = write_best_heights(heights, N, os.path.join(out_dir, f"best_heights_{N}_{stamp}.txt"))
    return int(res["best_energy"][r]), heights, path


def load_config(path="config.yaml"):
    """The reference's config.yaml (config.yaml:1-37); key names verbatim, including `betta_scheduling`."""
    import yaml

    with open(path) as f:
        return yaml.safe_load(f)


def _single_N(N, n_steps, init_mode, sp, n_runs, base_seed, mcmc_type, patience, label, runner, dist, histories):
    """One run_experiment + plot_energy_histories (experiments.py:1237-1256): returns (histories or None, best energies)
    and writes results/{label}.csv from the histories or, for large runs, from the on-device sums."""
    job = jb.make_job(N, n_steps, init_mode, sp, n_runs, base_seed, mcmc_type, patience)
    want = _want(histories, [job], dist, need_steps=True)
    (res,) = jb.JobSet([job], want=want, dist=dist, runner=runner).run()
    hist = _histories_of(res) if want == "histories" else None
    stats = None if hist is not None else mean_std_from_sums(res["step_sum"], res["step_sumsq"], res["step_count"])
    if _is_writer(dist):
        write_energy_csv(hist, label, stats=stats)
    return hist, [int(b) for b in res["best_energy"]]


def main(config="config.yaml", runner=None, dist=None, histories="auto"):
    """The reference's __main__ dispatch (experiments.py:1204-1391) without the plots; returns the result object."""
    cfg = load_config(config) if isinstance(config, str) else config
    et = cfg["experiment_type"]
    common = cfg["common"]
    n_steps, n_runs, verbose, init_mode = common["n_steps"], common["n_runs"], common["verbose"], common["initialization"]
    mcmc_type = common.get("mcmc_type", "board")
    patience = common.get("early_stop_patience", 100000)
    if patience == "None":  # `early_stop_patience: None` parses as a string (experiments.py:1217-1218)
        patience = None
    if et == "single_N":
        N = cfg["single_N"]["N"]
        sched_cfg = common["betta_scheduling"]
        if isinstance(sched_cfg["type"], list):
            out = {}
            for _sched, base_seed, _desc, label, sp in ex.build_schedules_from_types(sched_cfg["type"], sched_cfg, n_steps):
                out[label] = _single_N(N, n_steps, init_mode, sp, n_runs, base_seed, mcmc_type, patience, label, runner, dist, histories)
            return out
        _sched, base_seed, _desc, sp = ex.build_schedule_from_common(common, n_steps)
        return _single_N(N, n_steps, init_mode, sp, n_runs, base_seed, mcmc_type, patience, "Schedule", runner, dist, histories)
    if et == "measure_min_energy_vs_N":
        params = cfg["measure_min_energy_vs_N"]
        sched, base_seed, _desc, sp = ex.build_schedule_from_common(common, n_steps)
        modes = params.get("init_modes", [init_mode])
        if isinstance(modes, str):  # experiments.py:1302-1305
            modes = [modes]
        return measure_min_energy_vs_N(params["Ns"], n_steps, sched, schedule_params=sp, init_modes=modes, n_runs=n_runs,
                                       base_seed=base_seed, verbose=verbose, plot=True, out_path=common["output_path"],
                                       mcmc_type=mcmc_type, early_stop_patience=patience, runner=runner, dist=dist)
    if et == "beta_start_end_pairs":
        params = cfg["beta_start_end_pairs"]
        return run_beta_start_end_pairs(params["N"], n_steps, params["beta_start_ends"],
                                        annealing_type=params.get("annealing_type", "linear_annealing"), init_mode=init_mode,
                                        n_runs=n_runs, base_seed=common["betta_scheduling"].get("base_seed", 0), verbose=verbose,
                                        plot=True, out_path=params.get("output_path", common["output_path"]),
                                        out_path_acceptance=params.get("output_path_acceptance"), mcmc_type=mcmc_type,
                                        early_stop_patience=patience, runner=runner, dist=dist, histories=histories)
    if et == "compare_beta_end":
        params = cfg["compare_beta_end"]
        return run_compare_beta_end(params["Ns"], n_steps, params["beta_start_ends"],
                                    annealing_type=params.get("annealing_type", "linear_annealing"), init_mode=init_mode,
                                    n_runs=n_runs, base_seed=common["betta_scheduling"].get("base_seed", 0), verbose=verbose,
                                    plot=True, out_path=params.get("output_path"), mcmc_type=mcmc_type,
                                    early_stop_patience=patience, runner=runner, dist=dist, histories=histories)
    raise ValueError(f"Unknown experiment_type: {et}")


def _print_like_reference(cfg, res):
    """What the reference's __main__ prints with `verbose: true` (experiments.py:1264-1266, 1286-1288, 1325-1329, 1359-1362,
    1386-1390): bare numbers, one per line."""
    et = cfg["experiment_type"]
    if et == "single_N":
        pairs = list(res.values()) if isinstance(res, dict) else [res]
        for _hist, best in pairs:
            for e in best:
                print(e)
    elif et == "measure_min_energy_vs_N":
        for mode in res["results"]:  # (insertion order = the order of init_modes)
            for m in res["results"][mode]["mean_min_energies"]:
                print(m)
    elif et == "beta_start_end_pairs":
        for best in res["all_best_energies"].values():
            print(np.mean(best))
    elif et == "compare_beta_end":
        for part in (res["result_N1"], res["result_N2"]):
            for best in part["all_best_energies"].values():
                print(np.mean(best))


def cli(argv=None):
    """`python -m mcq_amd [config.yaml]`: the reference's `python experiments.py` (experiments.py:1204-1391) -- reads config.yaml
    from the working directory unless a path is given, dispatches on experiment_type, prints what the reference prints when
    `verbose` is set, and writes the CSV data products (no PNGs: plotting is out of scope).  Under torch.distributed.run
    (WORLD_SIZE > 1) every rank takes its share of each job's chains and rank 0 writes and prints."""
    import argparse

    ap = argparse.ArgumentParser(prog="python -m mcq_amd", description="MI355X-native drop-in for `python experiments.py` of galgantar/monte-carlo-collective")
    ap.add_argument("config", nargs="?", default="config.yaml", help="YAML file with the reference's keys (default: ./config.yaml)")
    ap.add_argument("--histories", choices=("auto", "yes", "no"), default="auto",
                    help="return full energy histories to the host (yes), use on-device statistics (no), or decide by size (auto)")
    args = ap.parse_args(argv)
    cfg = load_config(args.config)
    dist = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:  # one process per GPU, RCCL ("nccl") over xGMI
        import torch
        import torch.distributed as dist

        local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("MCQ_BACKEND", "nccl"), **({"device_id": torch.device("cuda", local)} if os.environ.get("MCQ_BACKEND", "nccl") == "nccl" else {}))
    try:
        res = main(cfg, dist=dist, histories={"auto": "auto", "yes": True, "no": False}[args.histories])
        if cfg["common"].get("verbose") and _is_writer(dist):
            _print_like_reference(cfg, res)
    finally:
        if dist is not None:
            dist.destroy_process_group()
    return res

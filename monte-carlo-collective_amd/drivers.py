"""Experiment drivers on top of the sweep boundary (the callers of run_experiment in the reference).

Same names, arguments, seed derivations, labels and result dicts as the reference:
    run_beta_start_end_pairs   experiments.py:741-846   pair_seed = base_seed + idx * 1000 (791)
    run_compare_beta_end       experiments.py:943-1029  second N at base_seed + 10000 (1000)
    measure_min_energy_vs_N    experiments.py:1031-1201 seed = base_seed + 10*idx + sum(ord(c)) % 1000 (1060-1067)
    main / load_config         experiments.py:1204-1391 config.yaml dispatch

Plotting is out of scope; with plot=True the drivers write the reference's CSV data
products (results/*.csv: per-step mean/std energy, binned acceptance rates, min energy and
steps-to-best per N) and no PNGs.  The reference's run_compare_beta_end raises TypeError after all
compute when plot=True (it passes annealing_type= / init_mode= to a plot function that lacks them,
experiments.py:1020-1021 vs 848); that bug is not reproduced.

Every driver takes `runner=`: a callable with run_chains' signature.  The default is the GPU path;
tests inject the CPU oracle to check the host logic without a GPU.
"""
import os

import numpy as np

from . import experiments as ex


def _runner_or_default(runner):
    return ex.run_chains if runner is None else runner


def _run_all(runner, jobs, trace):
    """Every job of a driver (one per beta pair / per (init_mode, N) cell).  On the GPU path the jobs are launched
    concurrently on separate streams; an injected runner gets them one by one."""
    prepared = []
    for (N, n_steps, init_mode, sp, n_runs, base_seed, mcmc_type, patience) in jobs:
        if n_runs > 1:
            if sp is None:
                raise ValueError("schedule_params is required for parallel execution when n_runs > 1")
        else:
            patience = None  # the n_runs == 1 branch does not forward early_stop_patience (experiments.py:550-558)
        prepared.append(dict(N=N, n_steps=n_steps, init_mode=init_mode, schedule_params=sp, seeds=ex.abi.seeds_for(base_seed, n_runs),
                             mcmc_type=mcmc_type, early_stop_patience=patience))
    if runner is None:
        return ex.run_chains_many(prepared, trace=trace)[0]
    return [runner(j["N"], j["n_steps"], j["init_mode"], j["schedule_params"], j["seeds"], mcmc_type=j["mcmc_type"],
                   early_stop_patience=j["early_stop_patience"], trace=trace)[0] for j in prepared]


def _run(runner, N, n_steps, init_mode, schedule_params, n_runs, base_seed, mcmc_type, early_stop_patience, trace):
    """run_experiment's semantics (experiments.py:475-573) on an arbitrary runner; returns the raw result dict."""
    if n_runs > 1:
        if schedule_params is None:
            raise ValueError("schedule_params is required for parallel execution when n_runs > 1")
        patience = early_stop_patience
    else:
        patience = None  # the n_runs == 1 branch does not forward early_stop_patience (experiments.py:550-558)
    seeds = ex.abi.seeds_for(base_seed, n_runs)
    res, _ = runner(N, n_steps, init_mode, schedule_params, seeds, mcmc_type=mcmc_type, early_stop_patience=patience, trace=trace)
    return res


def energy_statistics(all_histories):
    """Per-step mean and (population) std over runs, as plot_energy_histories computes them (experiments.py:593-595).
    Ragged histories (early stop) cannot be stacked -- the reference fails there too; here it is an explicit error."""
    lens = {len(h) for h in all_histories}
    if len(lens) != 1:
        raise ValueError("histories have different lengths (early stop); per-step statistics are undefined")
    e = np.asarray(all_histories)
    return e.mean(axis=0), e.std(axis=0)


def mean_std_from_sums(step_sum, step_sumsq, step_count):
    """Per-step mean and population std from integer sums (the on-device statistics of a resident trace).
    The mean equals np.mean of the int64 matrix bit for bit (all partial sums are exact in float64); the
    std is computed from the exact integer numerator n*sum(x^2) - sum(x)^2 and agrees with NumPy's two-pass
    formula to ~1e-15 relative."""
    s = np.asarray(step_sum, dtype=np.int64)
    q = np.asarray(step_sumsq, dtype=np.int64)
    n = np.asarray(step_count, dtype=np.int64)
    mean = s / n
    num = np.array([int(ni) * int(qi) - int(si) * int(si) for ni, qi, si in zip(n, q, s)], dtype=object)
    var = np.array([float(v) / (int(ni) * int(ni)) for v, ni in zip(num, n)], dtype=np.float64)
    return mean, np.sqrt(np.maximum(var, 0.0))


def acceptance_rates_from_bins(bin_accepted, bin_proposed):
    """Rates per bin from the on-device counts; empty bins are NaN like the reference (experiments.py:690-693)."""
    a = np.asarray(bin_accepted, dtype=np.float64)
    p = np.asarray(bin_proposed, dtype=np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(p > 0, a / p, np.nan)


def write_energy_csv(all_histories, label, out_dir="results"):
    """results/{label}.csv with columns step, mean_energy, std_energy (experiments.py:600-608)."""
    mean, std = energy_statistics(all_histories)
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
                             mcmc_type="full_3d", early_stop_patience=100000, runner=None):
    """experiments.py:741-846.  Returns {"all_histories": {label: [...]}, "all_best_energies": {label: [...]}}
    with label f"beta: {beta_start}->{beta_end}" (806)."""
    all_histories, all_best, all_acc, all_rej = {}, {}, {}, {}
    jobs, labels = [], []
    for idx, (beta_start, beta_end) in enumerate(beta_start_ends):
        sp = {"type": annealing_type, "beta_start": beta_start, "beta_end": beta_end}
        ex.build_schedule_from_params(annealing_type, n_steps, beta_start=beta_start, beta_end=beta_end)  # the reference's ValueErrors
        jobs.append((N, n_steps, init_mode, sp, n_runs, base_seed + idx * 1000, mcmc_type, early_stop_patience))
        labels.append(f"beta: {beta_start}->{beta_end}")
    for label, res in zip(labels, _run_all(runner, jobs, True)):
        all_histories[label] = [res["energy_hist"][r, : int(res["hist_len"][r])] for r in range(n_runs)]
        all_best[label] = [int(b) for b in res["best_energy"]]
        if plot and out_path_acceptance is not None:
            steps = [ex.accepted_rejected_steps(res, r) for r in range(n_runs)]
            all_acc[label], all_rej[label] = [s[0] for s in steps], [s[1] for s in steps]
        if verbose:
            for e in all_best[label]:
                print(e)
            print(np.mean(all_best[label]))
    if plot:
        for label, hist in all_histories.items():
            write_energy_csv(hist, label)
        if out_path_acceptance is not None:
            for label in all_histories:
                c, r = acceptance_rates_binned(all_acc[label], all_rej[label], n_steps, n_bins=100)
                write_acceptance_csv(c, r, label)
    return {"all_histories": all_histories, "all_best_energies": all_best}


def run_compare_beta_end(Ns, n_steps, beta_start_ends, annealing_type="linear_annealing", init_mode="random", n_runs=5,
                         base_seed=0, verbose=True, plot=True, out_path=None, mcmc_type="full_3d",
                         early_stop_patience=100000, runner=None):
    """experiments.py:943-1029: the pair experiment for two board sizes, the second at base_seed + 10000."""
    if len(Ns) != 2:
        raise ValueError("Ns must contain exactly 2 values")
    kw = dict(n_steps=n_steps, beta_start_ends=beta_start_ends, annealing_type=annealing_type, init_mode=init_mode,
              n_runs=n_runs, verbose=verbose, plot=False, out_path=None, out_path_acceptance=None, mcmc_type=mcmc_type,
              early_stop_patience=early_stop_patience, runner=runner)
    r1 = run_beta_start_end_pairs(N=Ns[0], base_seed=base_seed, **kw)
    r2 = run_beta_start_end_pairs(N=Ns[1], base_seed=base_seed + 10000, **kw)
    if plot:
        for N, r in ((Ns[0], r1), (Ns[1], r2)):
            for label, hist in r["all_histories"].items():
                write_energy_csv(hist, f"N{N}_{label}")
    return {"N1": Ns[0], "N2": Ns[1], "result_N1": r1, "result_N2": r2}


def measure_min_energy_vs_N(Ns, n_steps, beta_schedule, schedule_params=None, init_modes=["random"], n_runs=5, base_seed=100,
                            verbose=True, plot=True, out_path=None, mcmc_type="full_3d", early_stop_patience=100000,
                            runner=None):
    """experiments.py:1031-1201.  Histories are discarded by this experiment (1061), so no trace is produced."""
    if isinstance(init_modes, str):
        init_modes = [init_modes]
    if schedule_params is None:
        schedule_params = getattr(beta_schedule, "params", None)
    jobs = []
    for init_mode in init_modes:
        offset = sum(ord(c) for c in init_mode) % 1000
        for idx, N in enumerate(Ns):
            jobs.append((N, n_steps, init_mode, schedule_params, n_runs, base_seed + 10 * idx + offset, mcmc_type, early_stop_patience))
    all_res = iter(_run_all(runner, jobs, False))
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
    if plot:
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


def main(config="config.yaml", runner=None):
    """The reference's __main__ dispatch (experiments.py:1204-1391) without the plots; returns the result object."""
    cfg = load_config(config) if isinstance(config, str) else config
    et = cfg["experiment_type"]
    common = cfg["common"]
    n_steps, n_runs, verbose, init_mode = common["n_steps"], common["n_runs"], common["verbose"], common["initialization"]
    mcmc_type = common.get("mcmc_type", "board")
    patience = common.get("early_stop_patience", 100000)
    if patience == "None":  # `early_stop_patience: None` parses as a string (experiments.py:1217-1218)
        patience = None
    run = _runner_or_default(runner)
    if et == "single_N":
        N = cfg["single_N"]["N"]
        sched_cfg = common["betta_scheduling"]
        if isinstance(sched_cfg["type"], list):
            out = {}
            for _sched, base_seed, _desc, label, sp in ex.build_schedules_from_types(sched_cfg["type"], sched_cfg, n_steps):
                res = _run(run, N, n_steps, init_mode, sp, n_runs, base_seed, mcmc_type, patience, True)
                out[label] = ([res["energy_hist"][r, : int(res["hist_len"][r])] for r in range(n_runs)], [int(b) for b in res["best_energy"]])
                write_energy_csv(out[label][0], label)
            return out
        _sched, base_seed, _desc, sp = ex.build_schedule_from_common(common, n_steps)
        res = _run(run, N, n_steps, init_mode, sp, n_runs, base_seed, mcmc_type, patience, True)
        hist = [res["energy_hist"][r, : int(res["hist_len"][r])] for r in range(n_runs)]
        write_energy_csv(hist, "Schedule")
        return hist, [int(b) for b in res["best_energy"]]
    if et == "measure_min_energy_vs_N":
        params = cfg["measure_min_energy_vs_N"]
        sched, base_seed, _desc, sp = ex.build_schedule_from_common(common, n_steps)
        modes = params.get("init_modes", [init_mode])
        return measure_min_energy_vs_N(params["Ns"], n_steps, sched, schedule_params=sp, init_modes=modes, n_runs=n_runs,
                                       base_seed=base_seed, verbose=verbose, plot=True, out_path=common["output_path"],
                                       mcmc_type=mcmc_type, early_stop_patience=patience, runner=runner)
    if et == "beta_start_end_pairs":
        params = cfg["beta_start_end_pairs"]
        return run_beta_start_end_pairs(params["N"], n_steps, params["beta_start_ends"],
                                        annealing_type=params.get("annealing_type", "linear_annealing"), init_mode=init_mode,
                                        n_runs=n_runs, base_seed=common["betta_scheduling"].get("base_seed", 0), verbose=verbose,
                                        plot=True, out_path=params.get("output_path", common["output_path"]),
                                        out_path_acceptance=params.get("output_path_acceptance"), mcmc_type=mcmc_type,
                                        early_stop_patience=patience, runner=runner)
    if et == "compare_beta_end":
        params = cfg["compare_beta_end"]
        return run_compare_beta_end(params["Ns"], n_steps, params["beta_start_ends"],
                                    annealing_type=params.get("annealing_type", "linear_annealing"), init_mode=init_mode,
                                    n_runs=n_runs, base_seed=common["betta_scheduling"].get("base_seed", 0), verbose=verbose,
                                    plot=True, out_path=params.get("output_path"), mcmc_type=mcmc_type,
                                    early_stop_patience=patience, runner=runner)
    raise ValueError(f"Unknown experiment_type: {et}")

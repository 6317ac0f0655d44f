"""Experiment drivers (experiments.py:741-1201) against vectors captured from the reference's own drivers.

CPU part: the host logic (seed derivations, labels, result dicts, statistics) with the CPU oracle injected
as the chain runner.  GPU part (-m gpu): the same through the real path."""
import numpy as np
import pytest

import mcq_amd
from oracle import oracle

abi = mcq_amd.abi
dr = mcq_amd.drivers


def oracle_runner(N, n_steps, init_mode, schedule_params, seeds, mcmc_type="full_3d", early_stop_patience=None, trace=True):
    p = abi.make_params(N, n_steps, init_mode, schedule_params, len(seeds), mcmc_type=mcmc_type,
                        early_stop_patience=early_stop_patience, trace=trace)
    return oracle.run(p, np.asarray(seeds, dtype=np.uint32), trace=trace, states=False), 0.0


def _check_pairs(g, runner):
    a = dict(g["args"])
    r = dr.run_beta_start_end_pairs(a["N"], a["n_steps"], a["beta_start_ends"], annealing_type=a["annealing_type"],
                                    init_mode=a["init_mode"], n_runs=a["n_runs"], base_seed=a["base_seed"], verbose=False,
                                    plot=False, mcmc_type=a["mcmc_type"], early_stop_patience=a["early_stop_patience"], runner=runner)
    assert list(r["all_best_energies"].keys()) == [f"beta: {s}->{e}" for s, e in a["beta_start_ends"]]  # labels in pair order (806)
    assert r["all_best_energies"] == g["best"]
    for k, hs in r["all_histories"].items():
        assert [int(h[-1]) for h in hs] == g["final"][k] and [int(np.sum(h)) for h in hs] == g["hist_sum"][k]


def _check_min_vs_N(g, runner):
    a = dict(g["args"])
    sched = mcq_amd.build_schedule_from_params(a["schedule_params"]["type"], a["n_steps"], beta_start=1.0, beta_end=3.0)
    r = dr.measure_min_energy_vs_N(a["Ns"], a["n_steps"], sched, schedule_params=a["schedule_params"], init_modes=a["init_modes"],
                                   n_runs=a["n_runs"], base_seed=a["base_seed"], verbose=False, plot=False, mcmc_type=a["mcmc_type"],
                                   early_stop_patience=a["early_stop_patience"], runner=runner)
    assert r["Ns"] == a["Ns"]
    for im, want in g["results"].items():
        got = r["results"][im]
        assert [x.tolist() for x in got["all_min_energies"]] == want["all_min"]
        assert [x.tolist() for x in got["all_steps_to_best"]] == want["all_stb"]
        np.testing.assert_allclose(got["mean_min_energies"], want["mean_min"], rtol=0, atol=0)
        np.testing.assert_allclose(got["std_min_energies"], want["std_min"], rtol=1e-15)
        np.testing.assert_allclose(got["mean_steps_to_best"], want["mean_stb"], rtol=0, atol=0)
        np.testing.assert_allclose(got["std_steps_to_best"], want["std_stb"], rtol=1e-15)


def _check_compare(g, runner):
    a = dict(g["args"])
    r = dr.run_compare_beta_end(a["Ns"], a["n_steps"], a["beta_start_ends"], annealing_type=a["annealing_type"],
                                init_mode=a["init_mode"], n_runs=a["n_runs"], base_seed=a["base_seed"], verbose=False, plot=False,
                                mcmc_type=a["mcmc_type"], early_stop_patience=a["early_stop_patience"], runner=runner)
    assert (r["N1"], r["N2"]) == (g["N1"], g["N2"])
    assert r["result_N1"]["all_best_energies"] == g["best_N1"] and r["result_N2"]["all_best_energies"] == g["best_N2"]


def test_pairs_host_logic(golden):
    _check_pairs(golden.manifest["drivers"]["pairs"], oracle_runner)


def test_min_energy_vs_N_host_logic(golden):
    _check_min_vs_N(golden.manifest["drivers"]["min_vs_N"], oracle_runner)


def test_compare_beta_end_host_logic(golden):
    _check_compare(golden.manifest["drivers"]["compare"], oracle_runner)


def test_acceptance_binning(golden):
    g = golden.manifest["drivers"]["acceptance"]
    a = g["args"]
    res, _ = oracle_runner(a["N"], a["n_steps"], "random", a["schedule_params"], abi.seeds_for(a["base_seed"], a["n_runs"]), mcmc_type="board")
    steps = [mcq_amd.experiments.accepted_rejected_steps(res, r) for r in range(a["n_runs"])]
    centers, rates = dr.acceptance_rates_binned([s[0] for s in steps], [s[1] for s in steps], a["n_steps"], n_bins=100)
    want = np.array([np.nan if v is None else v for v in g["rates"]])
    np.testing.assert_array_equal(np.isnan(rates), np.isnan(want))
    np.testing.assert_allclose(rates[~np.isnan(want)], want[~np.isnan(want)], rtol=0, atol=0)
    assert centers[0] == 5.0 and centers[-1] == 995.0


def test_config_dispatch_and_csv(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    cfg = {"experiment_type": "beta_start_end_pairs",
           "common": {"n_steps": 200, "n_runs": 2, "verbose": False, "initialization": "random", "mcmc_type": "board",
                      "early_stop_patience": "None", "output_path": "figures/x.png",
                      "betta_scheduling": {"type": "linear_annealing", "base_seed": 42, "beta_const": 5.0, "beta_start": 1.0, "beta_end": 3.0}},
           "beta_start_end_pairs": {"N": 5, "beta_start_ends": [[0.5, 3.0]], "annealing_type": "linear_annealing",
                                    "output_path_acceptance": "figures/a.png"}}
    r = dr.main(cfg, runner=oracle_runner)
    assert list(r["all_best_energies"].keys()) == ["beta: 0.5->3.0"]
    rows = np.loadtxt(tmp_path / "results" / "beta: 0.5->3.0.csv", delimiter=",", skiprows=1)
    assert rows.shape == (201, 3) and rows[0, 0] == 0 and rows[-1, 0] == 200
    assert (tmp_path / "results" / "acceptance_rates_beta: 0.5->3.0.csv").exists()
    with pytest.raises(ValueError):
        dr.main({"experiment_type": "nope", "common": cfg["common"]}, runner=oracle_runner)
    with pytest.raises(ValueError):
        dr.run_compare_beta_end([4], 10, [[1, 2]], runner=oracle_runner)


def test_competition_writer(tmp_path):
    def runner(N, n_steps, init_mode, sp, seeds, mcmc_type="board", early_stop_patience=None, trace=False):
        p = abi.make_params(N, n_steps, init_mode, sp, len(seeds), mcmc_type=mcmc_type, trace=trace)
        return oracle.run(p, np.asarray(seeds, dtype=np.uint32), trace=trace, states=True), 0.0

    best, heights, path = dr.run_competition(N=7, n_runs=5, n_steps=800, base_seed=42, out_dir=str(tmp_path), runner=runner, timestamp="t")
    lines = open(path).read().split()
    assert path.endswith("best_heights_7_t.txt") and len(lines) == 49 and lines[0].startswith("0,0,") and lines[-1].startswith("6,6,")
    # the written board really has the reported energy: recount attacking pairs from the file
    cells = [tuple(int(v) for v in ln.split(",")) for ln in lines]
    e = 0
    for a in range(49):
        for b in range(a + 1, 49):
            d = [abs(cells[a][t] - cells[b][t]) for t in range(3)]
            nz = [x for x in d if x]
            e += len(set(nz)) == 1
    assert e == best


def test_ragged_histories_are_rejected_explicitly():
    with pytest.raises(ValueError):
        dr.energy_statistics([np.arange(5), np.arange(4)])


@pytest.mark.gpu
def test_drivers_on_the_gpu(golden):
    d = golden.manifest["drivers"]
    _check_pairs(d["pairs"], None)
    _check_min_vs_N(d["min_vs_N"], None)
    _check_compare(d["compare"], None)

"""The command-line entry (`python -m mcq_amd [config.yaml]` = the reference's `python experiments.py`, experiments.py:1204-1391)
fed with YAML FILES that use the reference's key set and its quirks (config.yaml:1-37): `early_stop_patience: None` (a string),
a list-valued `betta_scheduling.type` for single_N, `init_modes` as a list or a bare string, the experiment types the reference
dispatches on.  The YAML text here is this repository's own; the expected numbers are the vectors captured from the reference's
drivers (tests/golden/manifest.json["drivers"]).

CPU part: drivers.main on the files with the CPU oracle injected as the chain runner.  GPU part (-m gpu): the real command line
in a child process, its printed numbers and CSV files."""
import os
import subprocess
import sys

import numpy as np
import pytest

import mcq_amd
from tests.test_drivers import oracle_runner

dr = mcq_amd.drivers
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _yaml_min_vs_N(g, verbose, init_modes_text):
    a = g["args"]
    sp = a["schedule_params"]
    return f"""# measure_min_energy_vs_N through the reference's key set
experiment_type: "measure_min_energy_vs_N"

common:
  n_steps: {a['n_steps']}
  n_runs: {a['n_runs']}
  verbose: {'true' if verbose else 'false'}
  initialization: random
  mcmc_type: "{a['mcmc_type']}"
  early_stop_patience: {a['early_stop_patience']}
  betta_scheduling:
    type: "{sp['type']}"
    base_seed: {a['base_seed']}
    beta_const: 5.0
    beta_start: {sp['beta_start']}
    beta_end: {sp['beta_end']}
  output_path: "figures/min_energy.png"

single_N:
  N: 12

measure_min_energy_vs_N:
  Ns: {a['Ns']}
  init_modes: {init_modes_text}
"""


def _yaml_pairs_or_compare(kind, g, verbose=False):
    a = g["args"]
    body = (f"  N: {a['N']}\n" if kind == "beta_start_end_pairs" else f"  Ns: {a['Ns']}\n")
    return f"""experiment_type: {kind}
common:
  n_steps: {a['n_steps']}
  n_runs: {a['n_runs']}
  verbose: {'true' if verbose else 'false'}
  initialization: {a['init_mode']}
  mcmc_type: {a['mcmc_type']}
  early_stop_patience: None        # a string, like the reference's own config.yaml
  betta_scheduling: {{type: "exponential_annealing", base_seed: {a['base_seed']}, beta_const: 5.0, beta_start: 1.0, beta_end: 3.0}}
  output_path: "figures/unused.png"
{kind}:
{body}  beta_start_ends: {a['beta_start_ends']}
  annealing_type: "{a['annealing_type']}"
  output_path: "figures/pairs.png"
  output_path_acceptance: "figures/acc.png"
"""


def _check_min_vs_N_result(r, g):
    for im, want in g["results"].items():
        got = r["results"][im]
        assert [x.tolist() for x in got["all_min_energies"]] == want["all_min"]
        assert [x.tolist() for x in got["all_steps_to_best"]] == want["all_stb"]


def test_yaml_files_through_main(golden, tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    d = golden.manifest["drivers"]
    # measure_min_energy_vs_N: init_modes as a list ...
    (tmp_path / "config.yaml").write_text(_yaml_min_vs_N(d["min_vs_N"], False, '["random", "latin", "klarner"]'))
    cfg = dr.load_config("config.yaml")
    assert cfg["common"]["early_stop_patience"] == 100 and cfg["experiment_type"] == "measure_min_energy_vs_N"
    r = dr.main("config.yaml", runner=oracle_runner)
    _check_min_vs_N_result(r, d["min_vs_N"])
    assert (tmp_path / "results" / "min_energy_vs_N_klarner.csv").exists() and (tmp_path / "results" / "steps_to_best_vs_N_random.csv").exists()
    # ... and as a bare string (experiments.py:1302-1305)
    (tmp_path / "one.yaml").write_text(_yaml_min_vs_N(d["min_vs_N"], False, "latin"))
    r = dr.main("one.yaml", runner=oracle_runner)
    assert list(r["results"]) == ["latin"]
    assert [x.tolist() for x in r["results"]["latin"]["all_min_energies"]] == d["min_vs_N"]["results"]["latin"]["all_min"]
    # beta_start_end_pairs; `early_stop_patience: None` arrives as the string 'None' (experiments.py:1217-1218)
    (tmp_path / "pairs.yaml").write_text(_yaml_pairs_or_compare("beta_start_end_pairs", d["pairs"]))
    assert dr.load_config("pairs.yaml")["common"]["early_stop_patience"] == "None"
    r = dr.main("pairs.yaml", runner=oracle_runner)
    assert r["all_best_energies"] == d["pairs"]["best"]
    assert (tmp_path / "results" / "beta: 0.5->3.0.csv").exists()
    # compare_beta_end: the reference's default experiment_type (config.yaml:1); its plot bug (experiments.py:1020-1021) is not reproduced
    (tmp_path / "cmp.yaml").write_text(_yaml_pairs_or_compare("compare_beta_end", d["compare"], verbose=True))
    r = dr.main("cmp.yaml", runner=oracle_runner)
    assert r["result_N1"]["all_best_energies"] == d["compare"]["best_N1"] and r["result_N2"]["all_best_energies"] == d["compare"]["best_N2"]
    capsys.readouterr()
    dr._print_like_reference(dr.load_config("cmp.yaml"), r)
    want = [float(np.mean(v)) for part in ("best_N1", "best_N2") for v in d["compare"][part].values()]
    assert [float(x) for x in capsys.readouterr().out.split()] == want


def test_single_N_with_a_list_of_schedule_types(tmp_path, monkeypatch, capsys):
    """single_N with betta_scheduling.type as a list (experiments.py:1231-1266): one run_experiment per type, all from the same base_seed,
    labelled like the reference; a single type gives the "Schedule" CSV."""
    monkeypatch.chdir(tmp_path)
    text = """experiment_type: single_N
common:
  n_steps: 250
  n_runs: 3
  verbose: true
  initialization: klarner
  mcmc_type: board
  early_stop_patience: None
  betta_scheduling:
    type: ["constant", "linear_annealing", "sinusoidal_annealing"]
    base_seed: 11
    beta_const: 2.5
    beta_start: 0.5
    beta_end: 4.0
  output_path: figures/e.png
single_N: {N: 6}
"""
    (tmp_path / "config.yaml").write_text(text)
    out = dr.main("config.yaml", runner=oracle_runner)
    assert list(out) == ["Constant beta=2.5", "Linear 0.5->4.0", "Sinusoidal 0.5->4.0"]
    for label, sp in zip(out, ({"type": "constant", "beta_const": 2.5}, {"type": "linear_annealing", "beta_start": 0.5, "beta_end": 4.0},
                               {"type": "sinusoidal_annealing", "beta_start": 0.5, "beta_end": 4.0})):
        res, _ = oracle_runner(6, 250, "klarner", sp, mcq_amd.abi.seeds_for(11, 3), mcmc_type="board")
        hist, best = out[label]
        assert best == [int(b) for b in res["best_energy"]]
        assert [h.tolist() for h in hist] == [res["energy_hist"][r, :251].tolist() for r in range(3)]
        assert (tmp_path / "results" / f"{label}.csv").exists()
    capsys.readouterr()
    dr._print_like_reference(dr.load_config("config.yaml"), out)
    assert [int(x) for x in capsys.readouterr().out.split()] == [b for _h, best in out.values() for b in best]
    (tmp_path / "single.yaml").write_text(text.replace('["constant", "linear_annealing", "sinusoidal_annealing"]', "logarithmic_annealing"))
    hist, best = dr.main("single.yaml", runner=oracle_runner)
    assert len(best) == 3 and (tmp_path / "results" / "Schedule.csv").exists()


def test_module_entry_points_exist():
    import importlib

    assert callable(dr.cli)
    assert importlib.util.find_spec("monte-carlo-collective_amd.__main__") is not None
    with open(os.path.join(ROOT, "mcq_amd.py")) as f:
        assert "cli(" in f.read()


@pytest.mark.gpu
def test_command_line_on_the_gpu(golden, tmp_path):
    """`python -m mcq_amd` in a working directory that holds config.yaml, like `python experiments.py`: what it prints with
    `verbose: true` (the per-cell means while it runs, experiments.py:1083-1087, then every mean again, 1325-1329) and the CSVs."""
    g = golden.manifest["drivers"]["min_vs_N"]
    (tmp_path / "config.yaml").write_text(_yaml_min_vs_N(g, True, '["random", "latin", "klarner"]'))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "mcq_amd"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    means = [m for im in g["args"]["init_modes"] for m in g["results"][im]["mean_min"]]
    printed = [float(x) for x in r.stdout.split()]
    assert printed == means + means, (printed, means)
    rows = np.loadtxt(tmp_path / "results" / "min_energy_vs_N_random.csv", delimiter=",", skiprows=1)
    np.testing.assert_array_equal(rows[:, 0], g["args"]["Ns"])
    np.testing.assert_allclose(rows[:, 1], g["results"]["random"]["mean_min"], rtol=1e-15)
    # an explicit path, the hyphenated package name, and a file that selects another experiment
    (tmp_path / "pairs.yaml").write_text(_yaml_pairs_or_compare("beta_start_end_pairs", golden.manifest["drivers"]["pairs"], verbose=True))
    r = subprocess.run([sys.executable, "-m", "monte-carlo-collective_amd", "pairs.yaml", "--histories", "no"], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    want = [float(np.mean(v)) for v in golden.manifest["drivers"]["pairs"]["best"].values()]
    assert sorted(float(x) for x in r.stdout.split()[-len(want):]) == sorted(want)

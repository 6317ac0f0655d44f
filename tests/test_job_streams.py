"""The machinery a job list puts around its launches never changes a result (GPU): launches pacing each other through one progress table
(MCQ_FLAG_SHARED_PACING), static launch priorities, compute-unit layers of their own per group of launches (CU-masked HIP streams), and none of them."""
import numpy as np
import pytest

import mcq_amd

jb = mcq_amd.jobs
pytestmark = pytest.mark.gpu
SP = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}


def _jobs():
    out = []
    for init in ("random", "klarner"):
        for idx, N in enumerate((3, 5, 8, 12, 13, 17, 24)):
            out.append(jb.make_job(N, 700, init, SP, 48, 42 + 10 * idx + len(init), "board", None))
    out.append(jb.make_job(12, 500, "random", {"type": "exponential_annealing", "beta_start": 1.0, "beta_end": 3.0}, 32, 7, "full_3d", None))
    return out


def _run(monkeypatch, **env):
    for k in ("MCQ_CU_PARTITION", "MCQ_JOB_PACING", "MCQ_JOB_PRIORITY"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    js = jb.JobSet(_jobs(), want="summary")
    res = js.run()
    out = [(r["summary"]["min_best"], r["summary"]["accepted"], np.array(r["best_energy"]).tolist(), np.array(r["steps_to_best"]).tolist()) for r in res]
    state = (js.pacing, js.cu_partition is not None, [int(la.run.p.flags) for la in js.launches])
    js.close()
    return out, state


def test_pacing_priorities_and_cu_layers_change_nothing(monkeypatch):
    ref, st = _run(monkeypatch, MCQ_CU_PARTITION="0", MCQ_JOB_PACING="0", MCQ_JOB_PRIORITY="0")
    assert st[0] == "none" and not st[1] and all(f == 0 for f in st[2])
    got, st = _run(monkeypatch)  # the defaults: a list this small paces its launches against each other and takes CU layers
    assert got == ref and st[0] == "shared" and st[1] and all(f & mcq_amd.abi.FLAG_SHARED_PACING for f in st[2])
    got, st = _run(monkeypatch, MCQ_JOB_PACING="0")
    assert got == ref and st[0] == "static priorities" and any((f >> mcq_amd.abi.FLAG_PRIORITY_SHIFT) & 3 for f in st[2])
    got, st = _run(monkeypatch, MCQ_CU_PARTITION="0")
    assert got == ref and st[0] == "shared" and not st[1]
    got, st = _run(monkeypatch, MCQ_CU_PARTITION="1", MCQ_JOB_PACING="0", MCQ_JOB_PRIORITY="0")
    assert got == ref and st[1]

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


@pytest.mark.parametrize("want", ["summary", "stats", "histories"])
def test_native_pack_equals_the_tensor_operations(monkeypatch, want):
    """mcq_pack_summary_device (one or two small kernels per launch) against the same packing done with tensor operations (distributed.pack_job):
    counters, per-rank minima, per-chain slots, the four per-entry arrays of the reduced trace and the stopped-chain histogram of early stops."""
    jobs = [jb.make_job(8, 400, "random", SP, 48, 11, "board", 60), jb.make_job(8, 400, "latin", {"type": "constant", "beta_const": 2.0}, 48, 99, "board", 60),
            jb.make_job(12, 300, "random", SP, 37, 5, "board", None), jb.make_job(9, 200, "random", SP, 16, 3, "full_3d", None)]
    out = {}
    for mode in ("torch", "native"):
        monkeypatch.setenv("MCQ_PACK", mode)
        js = jb.JobSet(jobs, want=want)
        res = js.run()
        out[mode] = [{k: (np.array(v).copy() if not isinstance(v, dict) else dict(v)) for k, v in r.items()} for r in res]
        js.close()
    for a, b in zip(out["torch"], out["native"]):
        assert a.keys() == b.keys()
        for k in a:
            if isinstance(a[k], dict):
                assert a[k] == b[k], k
            elif k == "energy_hist":  # (rows of chains that stopped early hold nothing defined behind their last entry)
                for r, n in enumerate(a["hist_len"]):
                    np.testing.assert_array_equal(a[k][r, :n], b[k][r, :n], err_msg=f"{k}[{r}]")
            else:
                np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    if want == "stats":
        assert out["native"][0]["step_stopped"].sum() > 0  # chains of the early-stop jobs did stop

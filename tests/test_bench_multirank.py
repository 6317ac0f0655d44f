"""bench.py under torch.distributed.run with two ranks sharing the one GPU (gloo override for the summary
reduce): the multi-rank code path -- process group, barriers, MIN/SUM all-reduce, max-over-ranks timing,
rank-0 JSON line -- and the fact that the job's results do not depend on how the chains are sharded."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, launcher, chains):
    env = dict(os.environ, **extra_env)
    cmd = launcher + [os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--n-steps", "3000", "--chains", str(chains),
                      "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_ranks_equal_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    one = _bench({}, [sys.executable], 8192)
    two = _bench({"MCQ_BENCH_BACKEND": "gloo"},
                 [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                  "--master-port", str(port)], 4096)
    two_gpus = two.pop("n_gpus")
    assert two_gpus == 2 and one["n_gpus"] == 1
    assert two["config"]["chains_total"] == one["config"]["chains_total"] == 8192
    for k in ("min_energy", "mean_best_energy", "acceptance_rate"):  # same 8192 chains, same seeds, however they are split
        assert two[k] == one[k], k
    assert two["scaling"] == "weak" and two["unit"] == "moves/s" and two["value"] > 0
    # the bench contract: every key the driver reads, on both lines
    for line in (one, two):
        for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                  "data", "config", "roofline"):
            assert k in line, k
        assert line["higher_is_better"] is True and line["vs_baseline"] is None and "workload" in line["config"]
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"]) and line["roofline"]["bound"] == "hbm"
        # what makes an N > 1 run readable afterwards: every rank's own times side by side, the collective's payload, the queue setting
        n = line.get("n_gpus", two_gpus)
        assert set(line["per_rank"]) == {"init_ms", "sweep_ms", "reduce_ms", "step_ms"}
        assert all(len(v) == n and all(x > 0 for x in v) for v in line["per_rank"].values())
        assert line["allreduce"]["per_step"] == 1 and line["allreduce"]["payload_bytes"] == 8 * (6 + n)
        assert "gpu_max_hw_queues" in line["config"] and line["config"]["gpu_max_hw_queues"] is None  # c2: one stream, nothing raised
    assert two["allreduce"]["backend"] == "gloo" and one["allreduce"]["backend"].startswith("none")


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_rccl_path_single_rank():
    """The real backend of the multi-GPU bench ("nccl" = RCCL, device-bound process group, device-side all-reduce and
    barrier) on the one GPU of the box: a world of one rank launched exactly as the driver launches N of them."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    one = _bench({}, [sys.executable], 4096)
    rccl = _bench({"MCQ_BENCH_FORCE_DIST": "1"},
                  [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                   "--master-port", str(port)], 4096)
    assert rccl["n_gpus"] == 1
    for k in ("min_energy", "mean_best_energy", "acceptance_rate"):
        assert rccl[k] == one[k], k


@pytest.mark.gpu
@pytest.mark.timeout(1200)
@pytest.mark.parametrize("config", ["c2", "c5"])
def test_bench_starts_its_own_ranks(config):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts two child ranks (torch.distributed.run), relays
    rank 0's line and returns their status; the job's results equal the one-rank run of the same chains."""
    def run(gpus, chains, env):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "0", "--n-steps", "3000",
               "--chains", str(chains), "--no-cpu-baseline", "--config", config]
        e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        out = subprocess.run(cmd, env=dict(e, **env), cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    per_gpu = 4096 if config == "c2" else 512
    one = run(1, 2 * per_gpu, {})
    two = run(2, per_gpu, {"MCQ_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["chains_total"] == one["config"]["chains_total"]
    for k in ("min_energy", "acceptance_rate") + (("mean_best_energy",) if config == "c2" else ("min_energy_per_job",)):
        assert two[k] == one[k], k
    for line, n in ((one, 1), (two, 2)):
        assert all(len(v) == n for v in line["per_rank"].values()) and line["allreduce"]["per_step"] == 1
    if config == "c5":
        assert set(two["per_rank"]) == {"sweeps_ms", "launch_and_reduce_ms", "reduce_host_ms", "step_ms"}
        assert two["kernel_ms"]["sweeps"] > 0 and two["kernel_ms"]["reduce_on_device"] >= 0
        assert two["allreduce"]["payload_bytes"] == 8 * 16 * (6 + 2 + 2 * 1024 + 5 * 3001)


def test_self_launch_command(monkeypatch):
    """The parent of a self-launched multi-GPU bench never imports torch: it only builds the launcher's command line."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]

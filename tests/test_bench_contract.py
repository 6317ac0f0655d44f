"""CPU-only checks of what bench.py quotes: a PMC entry in profiles/hbm_traffic.json is evidence only for the kernel source
it was measured on, so the headline entry must carry the sha256 of the CURRENT csrc/mcq_hip.hip (tools/pmc_refresh.sh /
pmc_refresh.py write it); bench.py itself prints traffic: null for a stale entry."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_headline_pmc_entry_matches_the_kernel_source():
    with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
        e = json.load(f)["board_N12_c65536_s100000"]
    assert e["kernel_sha256"] == bench.kernel_sha(), "profiles/hbm_traffic.json is stale: run tools/pmc_refresh.sh on the GPU box, then tools/pmc_refresh.py"
    assert e["bytes_per_launch"] == e["read_bytes"] + e["write_bytes"] > 0
    pmc, why = bench.measured_traffic("board_N12_c65536_s100000")
    assert pmc is not None and why is None


def test_stale_or_missing_entries_yield_null():
    pmc, why = bench.measured_traffic("no_such_workload")
    assert pmc is None and "no PMC entry" in why


def test_usable_cpus_is_positive():
    assert 1 <= bench.usable_cpus() <= (os.cpu_count() or 1)


def test_hardware_queues_are_raised_on_demand_only(monkeypatch):
    """Importing the package leaves GPU_MAX_HW_QUEUES alone (RCCL-only processes, single launches); an entry point that is about
    to run more than 4 streams raises it before the first GPU call, and never over the user's own value."""
    import mcq_amd

    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    assert mcq_amd._lib.ensure_hw_queues(1) is None and "GPU_MAX_HW_QUEUES" not in os.environ
    assert mcq_amd._lib.ensure_hw_queues(4) is None
    assert mcq_amd._lib.ensure_hw_queues(18) == "24" and os.environ["GPU_MAX_HW_QUEUES"] == "24"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert mcq_amd._lib.ensure_hw_queues(18) == "8"
    with open(os.path.join(ROOT, "monte-carlo-collective_amd", "_lib.py")) as f:
        src = f.read()
    assert src.count('os.environ["GPU_MAX_HW_QUEUES"]') == 1  # only inside ensure_hw_queues


def test_per_rank_table_and_pmc_roofline_helpers():
    import torch

    t = bench.per_rank_table(torch, None, 0, 1, torch.device("cpu"), {"sweep_ms": 1.5, "reduce_ms": 0.25})
    assert t == {"reduce_ms": [0.25], "sweep_ms": [1.5]}
    line = {"roofline": {"traffic": None}}
    bench.roofline_from_pmc(line, None, "no PMC entry for x", 10.0, 0.0, 1)
    assert line["roofline"]["traffic"] is None and line["roofline"]["traffic_note"] == "no PMC entry for x"
    pmc = {"bytes_per_launch": 8e9, "valu_insts_per_launch": 2e9, "kernel_sha256": "x", "read_bytes": 5e9, "write_bytes": 3e9, "sweep_launches": 18}
    bench.roofline_from_pmc(line, pmc, None, 10.0, 0.0, 10**9)
    r = line["roofline"]
    assert r["traffic"] == 8e9 and abs(r["traffic_rate"] - 800.0) < 1e-9 and abs(r["valu_issue"]["achieved"] - 200.0) < 1e-9
    assert r["valu_issue"]["per_move"] == 2.0 and r["traffic_source"]["sweep_launches"] == 18 and "traffic_ratio" not in r

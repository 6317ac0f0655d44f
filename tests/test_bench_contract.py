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

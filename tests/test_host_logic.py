"""Host-side rules that need no GPU: the lane plan of a job list, the trace-length guard of the drivers, and the beta table that
follows the parameter block it is built from."""
import numpy as np
import pytest

import mcq_amd
from mcq_amd import abi, drivers, jobs

SP = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}


def _default(mode, N):
    return 8 if mode == abi.MODE_FULL3D or N > 12 else 4


def _waves(shapes, plan):
    return sum((n * g + 63) // 64 for (_, n, _), g in zip(shapes, plan))


def test_lane_plan_of_side_by_side_launches():
    simds = 1024
    cap = jobs.WAVES_PER_SIMD * simds
    # measure_min_energy_vs_N at BASELINE's per-GPU shape: the defaults (8 lanes beyond N = 12) would overflow the resident capacity,
    # 4 lanes everywhere fit in one round
    c4 = [(N, 3072, abi.MODE_BOARD) for N in range(3, 21)]
    assert _waves(c4, [_default(m, N) for N, _, m in c4]) > cap
    plan = jobs.plan_lanes(c4, simds, _default)
    assert plan == [4] * 18 and _waves(c4, plan) <= cap
    # the same cells with 8 192 chains each: several rounds whatever the lanes, but in this mix of light and heavy wavefronts 4 lanes
    # (16 chains per wavefront) beat the 8 that N = 13..20 would take on their own ...
    big = [(N, 24576, abi.MODE_BOARD) for N in range(3, 21)]
    assert jobs.plan_lanes(big, simds, _default) == [4] * 18
    # ... which a list of heavy launches alone does not get (their LDS slices would leave the CUs 11 wavefronts), nor N > 20
    heavy = [(N, 24576, abi.MODE_BOARD) for N in range(17, 21)]
    assert jobs.plan_lanes(heavy, simds, _default) == [8] * 4
    assert jobs.plan_lanes([(24, 16384, abi.MODE_BOARD), (12, 16384, abi.MODE_BOARD), (6, 16384, abi.MODE_BOARD)], simds, _default) == [8, 4, 4]
    # a lone small launch: more lanes while the device stays under half full, and only where the table says the step gets shorter
    assert jobs.plan_lanes([(17, 512, abi.MODE_BOARD)], simds, _default) == [16]
    assert jobs.plan_lanes([(24, 16384, abi.MODE_BOARD)], simds, _default) == [8]
    assert jobs.plan_lanes([(3, 512, abi.MODE_BOARD)], simds, _default) == [4]
    # boards beyond the measured 4-lane range are never forced to 4 lanes (N = 100: 16 chains would need 169 KB of LDS)
    for N in (33, 64, 99, 100, 128):
        d = _default(abi.MODE_BOARD, N)
        assert jobs.plan_lanes([(N, 40000, abi.MODE_BOARD)], simds, _default) == [d]
        assert jobs.plan_lanes([(N, 40000, abi.MODE_BOARD), (12, 40000, abi.MODE_BOARD)], simds, _default)[0] == d
    # full_3d keeps the library's choice
    assert jobs.plan_lanes([(12, 4096, abi.MODE_FULL3D), (12, 512, abi.MODE_BOARD)], simds, _default)[0] == 0  # (0 = the library's own choice)
    for N in range(2, 33):
        for g in (4, 8, 16):
            assert jobs.lone_ms(N, g) > 0


def test_long_runs_take_the_statistics_path():
    """A full trace row holds < 2^24 entries (include/mcq.h); the reference takes any n_steps, so `auto` must not pick
    histories for a longer run and an explicit request fails up front with a message, not inside the library."""
    short = [jobs.make_job(6, 1000, "random", SP, 2, 0, "board", None)]
    long_ = [jobs.make_job(6, (1 << 24) + 5, "random", SP, 2, 0, "board", None)]
    assert drivers._want("auto", short, None, need_steps=True) == "histories"
    assert drivers._want("auto", long_, None, need_steps=True) == "stats"
    assert drivers._want("auto", long_, None, need_steps=False) == "summary"
    with pytest.raises(ValueError, match="full histories hold at most"):
        drivers._want(True, long_, None, need_steps=True)
    edge = (1 << 24) - 64 - 1  # hist_stride = n_steps + 1 rounded up to 64
    assert abi.hist_stride_for(edge) < abi.MAX_HIST_STRIDE <= abi.hist_stride_for(edge + 64)


def test_beta_table_follows_the_parameter_block():
    """host_beta_table is derived from the struct's own fields: editing a Params after make_params changes the schedule that runs."""
    p = abi.make_params(6, 50, "random", SP, 4, mcmc_type="board")
    np.testing.assert_array_equal(abi.host_beta_table(p)[0], abi.beta_values(SP, 50))
    p.beta_end = 5.0
    np.testing.assert_array_equal(abi.host_beta_table(p)[0], abi.beta_values(dict(SP, beta_end=5.0), 50))
    p.sched = abi.SCHED["constant"]
    p.beta_const = 0.25
    np.testing.assert_array_equal(abi.host_beta_table(p)[0], np.full(50, 0.25))
    q = abi.make_params_sets(6, 50, "random", [SP, {"type": "constant", "beta_const": 2.0}], 16, mcmc_type="board")
    q.sets[1].beta_const = 3.5
    tab = abi.host_beta_table(q)
    assert tab.shape == (2, 50) and (tab[1] == 3.5).all()
    c = abi.copy_params(q)
    np.testing.assert_array_equal(abi.host_beta_table(c), tab)
    p._schedules = None  # a hand-filled block: the device evaluates the schedule
    assert abi.host_beta_table(p) is None


def test_lane_table_is_data_measured_on_the_current_kernel():
    """jobs.plan_lanes ranks launches by measured step times (tools/lane_table.py --json on one MI355X): the table carries the sha256
    of the kernel source it was measured on and must be regenerated when csrc/mcq_hip.hip changes."""
    import hashlib
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "monte-carlo-collective_amd", "csrc", "mcq_hip.hip"), "rb") as f:
        sha = hashlib.sha256(f.read()).hexdigest()
    assert jobs.LANE_TABLE_SHA == sha, "monte-carlo-collective_amd/lane_table.json is stale: run tools/lane_table.py --json on the GPU box (tools/finalize.sh) and copy it in"
    ms, col, _ = jobs._load_lane_table()
    assert set(range(3, 25)) <= set(ms) and all(len(r) == 3 and min(r) > 0 for r in ms.values())
    assert jobs.lone_ms(2, 4) > 0 and jobs.lone_ms(48, 8) == 2 * jobs.lone_ms(24, 8)

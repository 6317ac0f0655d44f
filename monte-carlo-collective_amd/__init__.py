"""MI355X-native Metropolis sweeps for the 3D N^2-queens problem: a drop-in for the sweep
path (run_experiment and below) of galgantar/monte-carlo-collective.  The compute path is
csrc/libmcq_hip.so (hand-written HIP for gfx950) behind the C-ABI of include/mcq.h; there is
no CPU fallback.  See DESIGN.md / INTEGRATION.md."""
from . import abi, build, _lib, experiments, distributed, jobs, drivers  # noqa: F401
from .drivers import measure_min_energy_vs_N, run_beta_start_end_pairs, run_compare_beta_end  # noqa: F401
from .experiments import (  # noqa: F401
    build_schedule_from_common,
    build_schedule_from_params,
    build_schedules_from_types,
    metropolis_mcmc,
    metropolis_mcmc_board,
    run_chains,
    run_experiment,
    run_single_chain,
    run_single_chain_board,
    run_single_chain_board_multithread,
    run_single_chain_multithread,
)

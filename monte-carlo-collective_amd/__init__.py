"""MI355X-native Metropolis sweeps for the 3D N^2-queens problem (drop-in for the sweep
path of galgantar/monte-carlo-collective).  See DESIGN.md / INTEGRATION.md."""
from . import abi  # noqa: F401

"""Importable alias: the package directory is `monte-carlo-collective_amd/` (not a Python
identifier), so `import mcq_amd` loads it through importlib and stands in for it.
`python -m mcq_amd [config.yaml]` runs the package's command line (drivers.cli: the reference's `python experiments.py`)."""
import importlib
import sys

_pkg = importlib.import_module("monte-carlo-collective_amd")
if __name__ == "__main__":
    _pkg.drivers.cli(sys.argv[1:])
else:
    sys.modules[__name__] = _pkg

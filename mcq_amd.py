"""Importable alias: the package directory is `monte-carlo-collective_amd/` (not a Python
identifier), so `import mcq_amd` loads it through importlib and stands in for it."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("monte-carlo-collective_amd")

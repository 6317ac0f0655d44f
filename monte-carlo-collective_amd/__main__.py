"""`python -m mcq_amd [config.yaml]` (or `python -m monte-carlo-collective_amd`): the command-line entry of the reference,
`python experiments.py` (experiments.py:1204-1391), on the GPU path.  See drivers.cli."""
import sys

from .drivers import cli

if __name__ == "__main__":
    cli(sys.argv[1:])

"""INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.  tools/check_integration_stub.py carries
that text verbatim (only the library path differs) and runs it against the package; these tests keep the two in step."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _doc_block():
    s = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.search(r"```python\n# experiments.py \(reference side\)\n(.*?)```", s, re.S).group(1)


def test_script_carries_the_documented_stub():
    script = open(os.path.join(ROOT, "tools", "check_integration_stub.py")).read()
    body = script[script.index("# --- verbatim from INTEGRATION.md"):script.index('\nif __name__ == "__main__":')]
    doc = [l for l in _doc_block().splitlines() if not l.startswith("_L = C.CDLL(")]
    got = [l for l in body.splitlines()[1:] if not l.startswith("_L = C.CDLL(")]
    assert [l.rstrip() for l in got if l.strip()] == [l.rstrip() for l in doc if l.strip()]


@pytest.mark.gpu
def test_documented_stub_runs_and_matches_the_package():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_integration_stub.py")], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, GRAFT_REPO_ROOT=ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "INTEGRATION.md stub == mcq_amd.run_experiment" in out.stdout

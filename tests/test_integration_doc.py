"""INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.  tools/check_integration_stub.py extracts that
text from the document and runs it against the package (GPU); on CPU the stub must at least compile and declare the structs of
include/mcq.h field for field."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_documented_stub_declares_the_header_structs():
    import check_integration_stub as chk

    import mcq_amd

    ns, _ = chk.stub_namespace()  # loads the library and executes the stub's definitions (no compute call)
    abi = mcq_amd.abi
    assert [f[0] for f in ns["McqParams"]._fields_] == [f[0] for f in abi.Params._fields_]
    assert [f[0] for f in ns["McqOutputs"]._fields_] == [f[0] for f in abi.Outputs._fields_]
    assert ctypes.sizeof(ns["McqParams"]) == ctypes.sizeof(abi.Params) and ctypes.sizeof(ns["McqOutputs"]) == ctypes.sizeof(abi.Outputs)
    assert f"abi_version={abi.ABI_VERSION}," in chk.doc_block()


@pytest.mark.gpu
def test_documented_stub_runs_and_matches_the_package():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_integration_stub.py")], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, GRAFT_REPO_ROOT=ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "INTEGRATION.md stub == mcq_amd.run_experiment" in out.stdout

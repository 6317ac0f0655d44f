"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every function include/mcq.h
declares for it (no compute call is made: there is no GPU here)."""
import ctypes
import os
import re

import numpy as np

import mcq_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mcq.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.findall(r"\b(mcq_\w+)\s*\(", src)


def test_header_symbols_are_exported():
    names = _declared()
    hip = [n for n in names if not n.startswith("mcq_oracle")]
    assert {"mcq_run_device", "mcq_run_device_timed", "mcq_default_lanes", "mcq_trace_stats_device", "mcq_run_host", "mcq_workspace_bytes", "mcq_state_bytes", "mcq_abi_version",
            "mcq_last_error", "mcq_device_count"} <= set(hip)
    L = mcq_amd._lib.lib()
    for n in hip:
        assert hasattr(L, n), n
    assert L.mcq_abi_version() == mcq_amd.abi.ABI_VERSION


def test_oracle_symbols_are_exported():
    from oracle import oracle

    L = oracle.lib()
    for n in _declared():
        if n.startswith("mcq_oracle"):
            assert hasattr(L, n), n


def test_struct_layout_matches_header():
    """sizeof / field order of the ctypes mirrors against a C compiler's view of the header."""
    import subprocess
    import tempfile

    abi = mcq_amd.abi
    fields = [f for f, _ in abi.Params._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "mcq.h"\nint main(){printf("%zu %zu", sizeof(mcq_params), sizeof(mcq_outputs));' + \
        "".join(f'printf(" %zu", offsetof(mcq_params, {f}));' for f in fields) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")], check=True)
        out = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == ctypes.sizeof(abi.Params) and int(out[1]) == ctypes.sizeof(abi.Outputs)
    assert [int(x) for x in out[2:]] == [getattr(abi.Params, f).offset for f in fields]


def test_pure_helpers_without_gpu():
    L = mcq_amd._lib.lib()
    assert L.mcq_state_bytes(12, 0) == 144 and L.mcq_state_bytes(12, 1) == 432 and L.mcq_state_bytes(1, 0) == 0
    p = mcq_amd.abi.make_params(12, 1000, "random", {"type": "constant", "beta_const": 1.0}, 10, mcmc_type="board")
    # beta table + c32 table + pacing table (2048 SIMD rows of 16 words) + exchange ladder (16 doubles) + chain records (628 + 36 words, rounded up to 64 bytes)
    assert L.mcq_workspace_bytes(ctypes.byref(p)) == 8192 + 4096 + 2048 * 16 * 4 + 128 + 10 * 672 * 4
    p.N = 200
    assert L.mcq_workspace_bytes(ctypes.byref(p)) == 0
    assert b"N out of range" in L.mcq_last_error()


def test_full_3d_beyond_32_sizing_and_limits():
    """full_3d at N = 33..64 (64-bit column words): the workspace holds the queen table as 32-bit entries and, for a random init, the
    N^3 cells np.random.choice permutes for as many chains at a time as 1 GiB holds; what the variant does not run is an explicit error."""
    L = mcq_amd._lib.lib()
    sp = {"type": "constant", "beta_const": 1.0}
    make = mcq_amd.abi.make_params
    fixed = 8192 + 4096 + 2048 * 16 * 4 + 128  # tables of 1000 steps, pacing rows, exchange ladder
    rec = lambda Q: ((628 + (3 * Q + 3) // 4 + 15) & ~15) * 4  # noqa: E731  (chain record: MT words, cursor, E0, 3 Q state bytes; 64-byte multiple)
    for chains, slots in ((10, 12), (1000, 1000), (5000, 1024)):
        p = make(64, 1000, "random", sp, chains, mcmc_type="full_3d")
        assert L.mcq_workspace_bytes(ctypes.byref(p)) == fixed + chains * (rec(4096) + 4096 * 4) + slots * 64**3 * 4
        p = make(64, 1000, "latin", sp, chains, mcmc_type="full_3d")
        assert L.mcq_workspace_bytes(ctypes.byref(p)) == fixed + chains * (rec(4096) + 4096 * 4)
    p = make(32, 1000, "random", sp, 10, mcmc_type="full_3d")  # up to N = 32: 16-bit entries, the permutation in LDS
    assert L.mcq_workspace_bytes(ctypes.byref(p)) == fixed + 10 * (rec(1024) + 1024 * 2)
    for kw, msg in (({"lanes_per_chain": 8}, b"16 lanes per chain"), ({"rng": "philox"}, b"MT19937 stream only")):
        p = make(33, 1000, "random", sp, 16, mcmc_type="full_3d", **kw)
        assert L.mcq_workspace_bytes(ctypes.byref(p)) == 0 and msg in L.mcq_last_error()
    p = make(48, 1000, "random", sp, 16, mcmc_type="full_3d")
    mcq_amd.abi.set_exchange(p, 10, [1.0, 0.8, 0.6, 0.4])
    assert L.mcq_workspace_bytes(ctypes.byref(p)) == 0 and b"without replica exchange" in L.mcq_last_error()
    assert L.mcq_default_lanes_n(1, 40) == 16 and L.mcq_default_lanes_n(1, 32) == 8


def test_stream_layout_rewinds_the_unconsumed_generation():
    """mcq_params.stream_states: NumPy holds the 624 words of the CURRENT generation, the kernels twist a generation block by block as its words are
    needed.  The library therefore rewinds the words behind the (rounded-up) position to the generation before; twisting them again -- what the
    kernels will do -- must give NumPy's words back, for every position class."""
    L = mcq_amd._lib.lib()
    rs = np.random.RandomState(77)
    UP, LO, A = 0x80000000, 0x7FFFFFFF, 0x9908B0DF
    for pos in (0, 1, 15, 16, 63, 64, 65, 226, 227, 228, 300, 396, 397, 398, 575, 576, 577, 608, 623, 624):
        rs.randint(0, 2**32, size=1000, dtype=np.uint32)
        key = np.array(rs.get_state()[1], dtype=np.uint32)
        st = np.concatenate([key, np.array([pos], dtype=np.uint32)])
        out = np.zeros(626, dtype=np.uint32)
        L.mcq_stream_layout(st.ctypes.data, out.ctypes.data)
        if pos == 624:  # the key IS the generation before the one the next draw starts
            assert out[624] == 0 and out[625] == 0 and (out[:624] == key).all()
            continue
        ge = min(624, (pos + 63) & ~63)
        assert out[624] == pos and out[625] == ge and (out[:ge] == key[:ge]).all()
        w = [int(x) for x in out[:624]]
        for i in range(ge, 624):  # the twist, word by word and in place, as the kernels run it
            y = (w[i] & UP) | (w[(i + 1) % 624] & LO)
            w[i] = w[(i + 397) % 624] ^ (y >> 1) ^ (A if y & 1 else 0)
        assert w == [int(x) for x in key], pos
    p = mcq_amd.abi.make_params(6, 10, "random", {"type": "constant", "beta_const": 1.0}, 2, mcmc_type="board", rng="philox")
    mcq_amd.abi.set_stream_states(p, [np.random.RandomState(1).get_state()] * 2)
    assert L.mcq_workspace_bytes(ctypes.byref(p)) == 0 and b"continues an MT19937 stream" in L.mcq_last_error()
    from oracle import oracle

    import pytest
    with pytest.raises(ValueError, match="continues an MT19937 stream"):  # the oracle refuses the same
        oracle.run(p, np.zeros(2, dtype=np.uint32))


def test_n_chains_bound_and_diag_gate():
    """n_chains >= 2^31 is MCQ_EINVAL (one workgroup per chain in the init kernel); MCQ_DIAG_LIB alone does not swap the library."""
    import subprocess
    import sys

    L = mcq_amd._lib.lib()
    p = mcq_amd.abi.make_params(6, 10, "random", {"type": "constant", "beta_const": 1.0}, 2**31, mcmc_type="board")
    assert L.mcq_workspace_bytes(ctypes.byref(p)) == 0 and b"n_chains out of range" in L.mcq_last_error()
    code = "import mcq_amd\ntry:\n    mcq_amd._lib.lib()\n    print('loaded')\nexcept mcq_amd._lib.McqError as e:\n    print('refused', e)\n"
    env = dict(os.environ, MCQ_DIAG_LIB="/nonexistent/libmcq_hip_x.so")
    env.pop("MCQ_ALLOW_DIAG", None)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True).stdout
    assert out.startswith("refused") and "MCQ_ALLOW_DIAG" in out

"""In-tree build of libmcq_hip.so (hipcc cross-compiles gfx950 without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "libmcq_hip.so")
SOURCES = [os.path.join(CSRC, "mcq_hip.hip")]
HEADER = os.path.join(os.path.dirname(HERE), "include", "mcq.h")
# -ffp-contract=off: the reference's schedule / acceptance expressions are evaluated without
# fused multiply-adds (hipcc's default would contract beta_start + frac * delta into an FMA).
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libmcq_hip.so")
    return exe


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(f) > t for f in SOURCES + [HEADER])


def build(force=False, verbose=False):
    """Compile the HIP kernels + C-ABI for gfx950; returns the path of the shared library."""
    if not force and not stale():
        return SO
    cmd = [hipcc()] + FLAGS + ["-o", SO] + SOURCES
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(" ".join(cmd))
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))

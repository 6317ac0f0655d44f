#!/usr/bin/env python3
"""Register / scratch / LDS table of every kernel in libmcq_hip.so, read from the code object's metadata notes.

    python tools/register_table.py [path/to/lib.so] > profiles/rNN_registers.txt

The library is a fat binary: the gfx950 code object is extracted with clang-offload-bundler and its
amdhsa.kernels note is parsed (vgpr_count, sgpr_count, *_spill_count, private_segment_fixed_size = scratch bytes per lane)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(so):
    with tempfile.TemporaryDirectory() as d:
        co, fb = os.path.join(d, "gfx950.co"), os.path.join(d, "fatbin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fb}", so, os.path.join(d, "unused")], check=True, capture_output=True)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fb}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k, blk=blk: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
        out.append({k: g(k) for k in ("name", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                      "private_segment_fixed_size", "group_segment_fixed_size", "agpr_count")})
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return r.stdout.strip().split("\n")


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "monte-carlo-collective_amd", "csrc", "libmcq_hip.so")
    ks = kernels(so)
    names = demangle([k["name"] for k in ks])
    rows = []
    for k, n in zip(ks, names):
        n = re.sub(r"^void \(anonymous namespace\)::", "", n).replace("((anonymous namespace)::KArgs)", "")
        rows.append((n, k))
    rows.sort(key=lambda r: r[0])
    print("# kernel <MODE(0 board,1 full_3d), G lanes/chain, PATIENCE, NT probe passes (0 = loop), REDUCED, PHILOX>")
    print(f"{'kernel':58s} {'vgpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch_B':>9s} {'waves/SIMD':>10s}")
    bad = 0
    for n, k in rows:
        v = int(k["vgpr_count"])
        alloc = (v + 7) // 8 * 8
        waves = min(8, 512 // max(alloc, 1))
        print(f"{n:58s} {k['vgpr_count']:>5s} {k['sgpr_count']:>5s} {k['vgpr_spill_count']:>6s} {k['sgpr_spill_count']:>6s} "
              f"{k['private_segment_fixed_size']:>9s} {waves:>10d}")
        bad += int(k["private_segment_fixed_size"]) > 0
    print(f"# {len(rows)} kernels, {bad} with scratch")


if __name__ == "__main__":
    main()

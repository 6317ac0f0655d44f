#!/usr/bin/env python3
"""Turn the passes of tools/pmc_refresh.sh into the entry bench.py quotes: profiles/hbm_traffic.json[key].

    python tools/pmc_refresh.py gpurun_out/rNN_pmc board_N12_c65536_s100000 [profiles/rNN_pmc_summary.json]

The entry records the sha256 of csrc/mcq_hip.hip the counters were measured on (written on the GPU box next to the passes)
and refuses to be written if the source has changed since: a stale entry is not evidence.  HBM / fabric bytes per launch:
reads = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B (the L2's memory-side requests by size; FETCH_SIZE tallies the
128-byte requests at 64 bytes, the gfx950 correction of MI355X_MICROARCH.md), writes = 64 x WRREQ_64B + 32 x the rest
(= WRITE_SIZE)."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, key = sys.argv[1], sys.argv[2]
    copy_to = sys.argv[3] if len(sys.argv) > 3 else None
    with open(os.path.join(src, "summary.json")) as f:
        summ = json.load(f)
    d = dict(summ["sweep"])
    n_disp = int(d.pop("_dispatches", 1)) or 1
    if n_disp > 1:  # a job list (bench.py --config c4 / c5 with --steps 1 --warmup 0): the entry holds the SUM over the step's sweep launches
        d = {k: v * n_disp for k, v in d.items()}
    with open(os.path.join(src, "kernel_sha256.txt")) as f:
        sha_box = f.read().strip()
    with open(os.path.join(ROOT, "monte-carlo-collective_amd", "csrc", "mcq_hip.hip"), "rb") as f:
        sha_now = hashlib.sha256(f.read()).hexdigest()
    if sha_box != sha_now:
        raise SystemExit(f"csrc/mcq_hip.hip changed since the counters were collected ({sha_box[:12]} -> {sha_now[:12]}): collect again")
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "monte-carlo-collective_amd/csrc"], capture_output=True, text=True).stdout.strip()
    rd = 32 * d["TCC_EA0_RDREQ_32B_sum"] + 64 * d["TCC_EA0_RDREQ_64B_sum"] + 128 * d["TCC_EA0_RDREQ_128B_sum"]
    wr = 64 * d["TCC_EA0_WRREQ_64B_sum"] + 32 * (d["TCC_EA0_WRREQ_sum"] - d["TCC_EA0_WRREQ_64B_sum"])
    entry = {
        "kernel_sha256": sha_now,
        "commit": commit + ("+uncommitted kernel edits" if dirty else ""),
        "bench_args": open(os.path.join(src, "bench_args.txt")).read().strip(),
        "sweep_launches": n_disp,
        "bytes_per_launch": rd + wr,
        "read_bytes": rd,
        "write_bytes": wr,
        "fetch_size_kb": d["FETCH_SIZE"],
        "write_size_kb": d["WRITE_SIZE"],
        "rdreq_by_size": {"32B": d["TCC_EA0_RDREQ_32B_sum"], "64B": d["TCC_EA0_RDREQ_64B_sum"], "128B": d["TCC_EA0_RDREQ_128B_sum"]},
        "l2_hit_rate": d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]),
        "valu_insts_per_launch": d["SQ_INSTS_VALU"],
        "salu_insts_per_launch": d["SQ_INSTS_SALU"],
        "lds_insts_per_launch": d["SQ_INSTS_LDS"],
        "lds_bank_conflict_share": d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"],
        "note": "rocprofv3 --pmc, one counter group per pass (tools/pmc_refresh.sh), sweep kernel, one launch; bytes = L2 memory-side "
                "requests by size (reads 32/64/128 B, writes 64/32 B); FETCH_SIZE counts a 128-byte request as 64 bytes on gfx950",
    }
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            allj = json.load(f)
    except (OSError, ValueError):
        allj = {}
    allj[key] = entry
    with open(path, "w") as f:
        json.dump(allj, f, indent=1, sort_keys=True)
        f.write("\n")
    if copy_to:
        with open(os.path.join(ROOT, copy_to), "w") as f:
            json.dump(summ, f, indent=1, sort_keys=True)
            f.write("\n")
    print(json.dumps({key: entry}, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Turn one tools/collect_evidence.sh session (gpurun_out/RNN_evidence) into the tracked files under profiles/.

    python tools/evidence_to_profiles.py r02

Refuses (through tools/pmc_refresh.py) if csrc/mcq_hip.hip has changed since the counters were collected."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    r = sys.argv[1] if len(sys.argv) > 1 else "r03"
    ev = os.path.join(ROOT, "gpurun_out", f"{r}_evidence")
    prof = os.path.join(ROOT, "profiles")
    run = lambda *a: subprocess.run([sys.executable] + list(a), check=True, cwd=ROOT, capture_output=True, text=True).stdout
    for src, key, out in (("pmc_headline", "board_N12_c65536_s100000", f"profiles/{r}_pmc_summary.json"),
                          ("pmc_c3", "full_3d_N12_c65536_s100000", f"profiles/{r}_pmc_summary_full3d.json"),
                          ("pmc_philox", "board_N12_c65536_s100000_philox", f"profiles/{r}_pmc_summary_philox.json"),
                          ("pmc_n24", "board_N24_c65536_s100000_reduced", f"profiles/{r}_pmc_summary_n24.json")):
        if os.path.exists(os.path.join(ev, src, "summary.json")):
            run("tools/pmc_refresh.py", os.path.join("gpurun_out", f"{r}_evidence", src), key, out)
            print("refreshed", key)
    copies = {"bench.json": f"{r}_bench.json", "configs.jsonl": f"{r}_configs.jsonl", "patience.txt": f"{r}_patience.txt",
              "occupancy_board.txt": f"{r}_occupancy_board.txt", "occupancy_full3d.txt": f"{r}_occupancy_full3d.txt",
              "stamps.txt": f"{r}_stamp_shares.txt", "occupancy_n24.txt": f"{r}_occupancy_n24.txt", "c4_1e6.json": f"{r}_c4_1e6_steps.json", "small_launches.txt": f"{r}_small_launches.txt", "c5_rss.txt": f"{r}_c5_host_memory.txt",
              "lds_stride_ab.txt": f"{r}_lds_stride_ab.txt", "lds_stride_pmc.txt": f"{r}_lds_stride_pmc.txt",
              "kernel_stats_bench.json": f"{r}_kernel_stats_bench.json"}
    for a, b in copies.items():
        if os.path.exists(os.path.join(ev, a)):
            shutil.copy(os.path.join(ev, a), os.path.join(prof, b))
    ks = glob.glob(os.path.join(ev, "kernel_stats", "**", "*kernel_stats.csv"), recursive=True)
    if ks:  # gpurun merges every session into the same directory: take the newest trace
        shutil.copy(max(ks, key=os.path.getmtime), os.path.join(prof, f"{r}_kernel_stats.csv"))
    with open(os.path.join(prof, f"{r}_registers.txt"), "w") as f:
        f.write(run("tools/register_table.py"))
    # the Philox evidence in one place: bench line + counters
    try:
        with open(os.path.join(prof, "hbm_traffic.json")) as f:
            tr = json.load(f)
        lines = [json.loads(l) for l in open(os.path.join(ev, "configs.jsonl")) if l.strip()]
        ph = [l for l in lines if "rng=philox" in l["config"]["workload"] and "mcmc_type=board" in l["config"]["workload"]]
        mt = json.load(open(os.path.join(ev, "bench.json")))
        e = tr.get("board_N12_c65536_s100000_philox")
        if ph and e:
            moves = 65536 * 100000
            with open(os.path.join(prof, f"{r}_philox.json"), "w") as f:
                json.dump({"workload": ph[0]["config"]["workload"], "moves_per_s": ph[0]["value"], "sweep_ms": ph[0]["kernel_ms"]["sweep"],
                           "mt19937_moves_per_s_same_session": mt["value"], "mt19937_sweep_ms_same_session": mt["kernel_ms"]["sweep"],
                           "read_bytes_per_move": e["read_bytes"] / moves, "write_bytes_per_move": e["write_bytes"] / moves,
                           "algorithmic_bytes_per_move": 4.125, "l2_hit_rate": e["l2_hit_rate"],
                           "valu_insts_per_move": e["valu_insts_per_launch"] / moves, "kernel_sha256": e["kernel_sha256"],
                           "mt19937_read_bytes_per_move": tr["board_N12_c65536_s100000"]["read_bytes"] / moves,
                           "mt19937_write_bytes_per_move": tr["board_N12_c65536_s100000"]["write_bytes"] / moves}, f, indent=1)
                f.write("\n")
    except (OSError, KeyError, ValueError) as ex:
        print("philox summary skipped:", ex)
    print("profiles updated for", r)


if __name__ == "__main__":
    main()

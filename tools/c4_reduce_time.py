"""launch / reduce split of BASELINE configs[3] on one GPU (per-GPU shape 1 024 chains per cell and 8 192)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch, mcq_amd
jb = mcq_amd.jobs
sp = {"type": "linear_annealing", "beta_start": 1.0, "beta_end": 3.0}
for chains in (1024, 8192):
    jobs = []
    for init in ("random", "latin", "klarner"):
        off = sum(ord(c) for c in init) % 1000
        for idx, N in enumerate(range(3, 21)):
            jobs.append(jb.make_job(N, 100000, init, sp, chains, 42 + 10 * idx + off, "board", None))
    js = jb.JobSet(jobs, want="summary")
    js.launch(); js.reduce(); torch.cuda.synchronize()
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); js.launch(); te = time.perf_counter(); js.synchronize(); torch.cuda.synchronize(); t1 = time.perf_counter(); js.reduce(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"c4 {chains}/cell: enqueue {1e3*(te-t0):.1f} ms, launches done {1e3*(t1-t0):.1f} ms, reduce {1e3*(t2-t1):.1f} ms")
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); js.launch(); js.reduce(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"   back to back: {1e3*(t2-t0):.1f} ms")
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); js.launch(); js.synchronize(); js.reduce(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"   wait for the launch streams on the host, then reduce: {1e3*(t2-t0):.1f} ms")
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); js.launch(); js.reduce(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"   back to back again: {1e3*(t2-t0):.1f} ms")

"""Where the time of jobs.JobSet.reduce() goes for BASELINE configs[4] on one GPU (evidence for profiles/r03_reduce_path.txt)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch, mcq_amd
jb, dm = mcq_amd.jobs, mcq_amd.distributed
pairs = [(s, e) for s in (0.1, 0.5, 1.0, 2.0) for e in (2.0, 3.0, 5.0, 8.0)]
for chains in (1024, 8192):
    jobs = [jb.make_job(24, 100000, "random", {"type": "sinusoidal_annealing", "beta_start": s, "beta_end": e}, chains, 42 + 1000 * i, "board", None) for i, (s, e) in enumerate(pairs)]
    js = jb.JobSet(jobs, want="stats")
    js.launch(); js.reduce(); torch.cuda.synchronize()
    for _ in range(2):
        t0 = time.perf_counter(); js.launch(); js.synchronize(); torch.cuda.synchronize(); t1 = time.perf_counter(); js.reduce(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"c5 {chains}/pair: launch+sweep {1e3*(t1-t0):.1f} ms, reduce {1e3*(t2-t1):.1f} ms")
    if hasattr(js, "_local_views"):
        sync = torch.cuda.synchronize
        t0 = time.perf_counter(); v = js._local_views(); sync(); t1 = time.perf_counter()
        js.buf.zero_()
        for i, lay in enumerate(js.layouts):
            dm.pack_job(js.buf, lay, js.rank, js.shards[i][1], {k: x.to(js.buf.device) for k, x in v[i].items()}, torch)
        sync(); t2 = time.perf_counter()
        host = js.buf.cpu().numpy(); t3 = time.perf_counter()
        out = [dm.unpack_job(host, lay) for lay in js.layouts]; t4 = time.perf_counter()
        print(f"   views {1e3*(t1-t0):.1f} ms, pack {1e3*(t2-t1):.1f} ms, D2H {1e3*(t3-t2):.1f} ms ({js.buf.numel()*8/1e6:.0f} MB), unpack {1e3*(t4-t3):.1f} ms")
    # the bench's own sequence: launch and reduce back to back, no synchronisation in between
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); js.launch(); t1 = time.perf_counter(); js.reduce(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"   back to back: enqueue {1e3*(t1-t0):.1f} ms, launch + reduce {1e3*(t2-t0):.1f} ms")
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); js.launch(); js.synchronize(); js.reduce(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"   launch, wait for the launch streams on the host, reduce: {1e3*(t2-t0):.1f} ms")

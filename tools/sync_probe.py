"""How long the host takes to notice that a launch has finished, by the way it waits (evidence for profiles/r03_reduce_path.txt)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch, mcq_amd
jb = mcq_amd.jobs
pairs = [(s, e) for s in (0.1, 0.5, 1.0, 2.0) for e in (2.0, 3.0, 5.0, 8.0)]
jobs = [jb.make_job(24, 100000, "random", {"type": "sinusoidal_annealing", "beta_start": s, "beta_end": e}, 1024, 42 + 1000 * i, "board", None) for i, (s, e) in enumerate(pairs)]
js = jb.JobSet(jobs, want="stats")
js.launch(); js.reduce(); torch.cuda.synchronize()
def t(label, fn, n=4):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); js.launch(); fn(); out.append(1e3 * (time.perf_counter() - t0))
    print(label, " ".join(f"{x:.1f}" for x in out))
ev = torch.cuda.Event()
t("stream.synchronize()      ", lambda: js.synchronize())
t("torch.cuda.synchronize()  ", lambda: torch.cuda.synchronize())
t("event.synchronize()       ", lambda: (ev.record(js.launches[0].stream), ev.synchronize()))
t("event.query() spin        ", lambda: (ev.record(js.launches[0].stream), [None for _ in iter(lambda: ev.query(), True)]))
t("reduce() (ends in .cpu()) ", lambda: js.reduce())
t("spin on event, then reduce", lambda: (ev.record(js.launches[0].stream), [None for _ in iter(lambda: ev.query(), True)], js.reduce()))
t("device sync, then reduce  ", lambda: (torch.cuda.synchronize(), js.reduce()))

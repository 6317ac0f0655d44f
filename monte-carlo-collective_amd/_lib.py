"""ctypes binding of libmcq_hip.so -- the only compute path of this package.

There is deliberately no CPU fallback: if the library is missing or no GPU is present the
calls raise.  (The CPU oracle under oracle/ is test infrastructure and is never imported here.)
"""
import ctypes as C
import importlib.util
import os
import sys
import warnings

import numpy as np

from . import abi, build as _build

_lib = None

HW_QUEUES = 24


def ensure_hw_queues(n_streams):
    """A job list of many small launches runs one HIP stream per launch.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES
    hardware queues (default 4) and launches that share a queue run one after the other: with the default, the 18 launches of
    measure_min_energy_vs_N at 3 x 1 024 chains each run at a third of the rate they reach with 16 queues
    (profiles/r02_small_launches.txt).  The variable is read when the HIP runtime initialises, so an entry point that is about to
    launch on more than 4 streams (jobs.JobSet, bench.py --config c4) calls this BEFORE the first GPU call: it sets
    GPU_MAX_HW_QUEUES=24 unless the user chose a value, and warns when the runtime is already up.  Nothing is changed at import,
    and nothing at all for runs of up to 4 streams (single launches, bench.py's default, RCCL-only processes).
    Returns the value in force (None: the runtime's default)."""
    if n_streams > 4 and "GPU_MAX_HW_QUEUES" not in os.environ:
        _torch = sys.modules.get("torch")
        if _torch is not None and _torch.cuda.is_initialized():
            warnings.warn(f"mcq_amd: {n_streams} concurrent launches, but the GPU runtime was initialised before GPU_MAX_HW_QUEUES could be "
                          "raised: they will share 4 hardware queues (about a third of the rate).  Export GPU_MAX_HW_QUEUES=24 before the "
                          "first GPU call.", RuntimeWarning, stacklevel=3)
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = str(HW_QUEUES)
    return os.environ.get("GPU_MAX_HW_QUEUES")


class McqError(RuntimeError):
    pass


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP
    runtimes in one process cannot both open the GPU, and streams / events of one are meaningless
    to the other, so when torch is installed its runtime is loaded first (RTLD_GLOBAL): the dynamic
    linker then binds libmcq_hip.so's NEEDED libamdhip64.so.7 to it, and a later `import torch`
    finds it already mapped.  MCQ_HIP_RUNTIME=system skips this (pure-ctypes users without torch)."""
    if os.environ.get("MCQ_HIP_RUNTIME", "") == "system":
        return None
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    try:
        return C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        return None


_hip = None


def hip_runtime():
    """The HIP runtime library this process uses (PyTorch's bundled copy when torch is installed, else the system's)."""
    global _hip
    if _hip is None:
        _hip = _share_torch_hip_runtime()
        if _hip is None:
            _hip = C.CDLL("libamdhip64.so")
    return _hip


def cu_masked_stream(cu_ids, n_cus):
    """A HIP stream whose kernels run on the compute units `cu_ids` only (hipExtStreamCreateWithCUMask): what a job list uses to give each of
    its side-by-side launches CUs of its own, so that a CU's instruction cache holds ONE sweep kernel instead of a dozen (jobs.JobSet).
    Returns the raw hipStream_t (an int), or None when the runtime refuses."""
    words = (int(n_cus) + 31) // 32
    mask = (C.c_uint32 * words)()
    for c in cu_ids:
        mask[int(c) >> 5] |= 1 << (int(c) & 31)
    st = C.c_void_p()
    fn = hip_runtime().hipExtStreamCreateWithCUMask
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    if fn(C.byref(st), words, mask) != 0 or not st.value:
        return None
    return st.value


def destroy_stream(handle):
    """hipStreamDestroy for a stream made by cu_masked_stream (after the work on it has been waited for)."""
    fn = hip_runtime().hipStreamDestroy
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p]
    fn(C.c_void_p(handle))


def lib():
    """Load csrc/libmcq_hip.so (building it if the sources are newer and hipcc exists)."""
    global _lib
    if _lib is None:
        hip_runtime()
        so = _build.SO
        # A diagnostic build of the same library (tools/stamp_profile.sh, tools/wave_times.sh).  Swapping the product library
        # through the environment is opt-in: without MCQ_ALLOW_DIAG=1 the variable is refused, not silently honoured.
        diag = os.environ.get("MCQ_DIAG_LIB")
        if diag:
            if os.environ.get("MCQ_ALLOW_DIAG") != "1":
                raise McqError("MCQ_DIAG_LIB is set but MCQ_ALLOW_DIAG=1 is not: refusing to replace libmcq_hip.so")
            if "oracle" in os.path.basename(diag):
                raise McqError("MCQ_DIAG_LIB must be a build of csrc/mcq_hip.hip, never the oracle")
            so = diag
        elif _build.stale():
            try:
                _build.build()
            except RuntimeError as e:
                if not os.path.exists(so):
                    raise McqError(f"libmcq_hip.so is not built and could not be built: {e}") from e
        try:
            L = C.CDLL(so)
        except OSError as e:
            raise McqError(f"cannot load {so}: {e}") from e
        L.mcq_abi_version.restype = C.c_int
        L.mcq_last_error.restype = C.c_char_p
        L.mcq_device_count.restype = C.c_int
        L.mcq_default_lanes.restype = C.c_int32
        L.mcq_default_lanes.argtypes = [C.c_int32]
        L.mcq_stream_layout.restype = None
        L.mcq_stream_layout.argtypes = [C.c_void_p, C.c_void_p]
        L.mcq_default_lanes_n.restype = C.c_int32
        L.mcq_default_lanes_n.argtypes = [C.c_int32, C.c_int32]
        L.mcq_effective_lanes.restype = C.c_int32
        L.mcq_effective_lanes.argtypes = [C.POINTER(abi.Params)]
        L.mcq_device_simds.restype = C.c_int32
        L.mcq_state_bytes.restype = C.c_size_t
        L.mcq_state_bytes.argtypes = [C.c_int32, C.c_int32]
        L.mcq_state_bytes_for.restype = C.c_size_t
        L.mcq_state_bytes_for.argtypes = [C.POINTER(abi.Params)]
        L.mcq_workspace_bytes.restype = C.c_size_t
        L.mcq_workspace_bytes.argtypes = [C.POINTER(abi.Params)]
        L.mcq_run_device.restype = C.c_int
        L.mcq_run_device.argtypes = [C.POINTER(abi.Params), C.c_void_p, C.POINTER(abi.Outputs), C.c_void_p, C.c_size_t, C.c_void_p]
        L.mcq_run_device_timed.restype = C.c_int
        L.mcq_run_device_timed.argtypes = [C.POINTER(abi.Params), C.c_void_p, C.POINTER(abi.Outputs), C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.mcq_trace_stats_device.restype = C.c_int
        L.mcq_trace_stats_device.argtypes = [C.POINTER(abi.Params), C.POINTER(abi.Outputs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mcq_pack_summary_device.restype = C.c_int
        L.mcq_pack_summary_device.argtypes = [C.POINTER(abi.Params), C.POINTER(abi.Outputs), C.c_int64, C.POINTER(abi.PackSlot), C.c_void_p, C.c_void_p]
        L.mcq_beta_table_device.restype = C.c_int
        L.mcq_beta_table_device.argtypes = [C.POINTER(abi.Params), C.c_void_p, C.c_void_p, C.c_void_p]
        L.mcq_run_host.restype = C.c_int
        L.mcq_run_host.argtypes = [C.POINTER(abi.Params), C.c_void_p, C.POINTER(abi.Outputs), C.POINTER(C.c_double)]
        if L.mcq_abi_version() != abi.ABI_VERSION:
            raise McqError("libmcq_hip.so ABI version mismatch; rebuild")
        _lib = L
    return _lib


def _check(rc):
    if rc == abi.OK:
        return
    msg = lib().mcq_last_error().decode(errors="replace")
    if rc == abi.EINVAL:
        raise ValueError(msg)
    if rc == abi.ENOMEM:
        raise MemoryError(msg)
    raise McqError(msg)


def device_count():
    return lib().mcq_device_count()


def effective_lanes(params):
    """Lanes of a wavefront per chain a launch with these parameters runs with on the current device."""
    return int(lib().mcq_effective_lanes(C.byref(params)))


def beta_table_device(params):
    """The beta(step) table(s) the sweep reads, computed on the GPU (mcq_beta_table_device): float64 [n_steps] or
    [n_sets, n_steps], plus the float32 factor table of the accept bracket.  Inspection / tests."""
    import torch

    L = lib()
    p = abi.copy_params(params)
    sets = max(1, int(p.n_sets))
    n = int(p.n_steps)
    dev = torch.device("cuda", torch.cuda.current_device())
    b = torch.zeros(sets * n, dtype=torch.float64, device=dev)
    c = torch.zeros(sets * n, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev)
    _check(L.mcq_beta_table_device(C.byref(p), b.data_ptr(), c.data_ptr(), C.c_void_p(st.cuda_stream)))
    st.synchronize()
    shape = (sets, n) if sets > 1 else (n,)
    return b.cpu().numpy().reshape(shape), c.cpu().numpy().reshape(shape)


def run_host(params, seeds, trace=True, states=True):
    """All chains on the GPU with NumPy (host) buffers; returns ({field: ndarray}, kernel_seconds)."""
    L = lib()
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    if seeds.shape != (params.n_chains,):
        raise ValueError("seeds must have one entry per chain")
    p = abi.copy_params(params)
    p.trace = abi.trace_mode(trace)
    tab = abi.host_beta_table(p) if not p.beta_table else None  # beta as the reference's own NumPy arithmetic gives it
    if tab is not None:
        p.beta_table = tab.ctypes.data
    arrays = {k: np.zeros(shape, dtype=abi.OUTPUT_DTYPES[k])
              for k, shape in abi.output_shapes(p, trace=trace, states=states).items()}
    out = abi.Outputs()
    for k, a in arrays.items():
        setattr(out, k, a.ctypes.data)
    secs = C.c_double(0.0)
    _check(L.mcq_run_host(C.byref(p), seeds.ctypes.data, C.byref(out), C.byref(secs)))
    return arrays, secs.value


class DeviceRun:
    """Device-resident buffers for repeated launches (bench.py, multi-GPU driver).

    torch is used only as the allocator / stream provider; the kernels are launched by
    libmcq_hip.so through raw device pointers."""

    def __init__(self, params, seeds, trace=True, states=True, device=None, stream_words=True):
        """stream_words=False leaves mcq_outputs.stream_words NULL (a job list has no use for it: one small kernel less per launch)."""
        import torch

        self.torch = torch
        self.L = lib()
        self.p = abi.copy_params(params)
        self.p.trace = abi.trace_mode(trace)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        tdt = {np.int32: torch.int32, np.int64: torch.int64, np.uint8: torch.uint8, np.uint64: torch.int64, np.uint32: torch.int32}
        self.t = {}
        self.out = abi.Outputs()
        with torch.cuda.device(self.device):
            for k, shape in abi.output_shapes(self.p, trace=trace, states=states).items():
                if k == "stream_words" and not stream_words:
                    continue
                self.t[k] = torch.empty(shape, dtype=tdt[abi.OUTPUT_DTYPES[k]], device=self.device)
                setattr(self.out, k, self.t[k].data_ptr())
            s = np.ascontiguousarray(seeds, dtype=np.uint32)
            if s.shape != (self.p.n_chains,):
                raise ValueError("seeds must have one entry per chain")
            self.seeds = torch.from_numpy(s.view(np.int32).copy()).to(self.device)
            tab = abi.host_beta_table(self.p) if not self.p.beta_table else None  # beta as the reference's own NumPy arithmetic gives it
            if tab is not None:
                self.beta = torch.from_numpy(tab).to(self.device)
                self.p.beta_table = self.beta.data_ptr()
            self.ws_bytes = int(self.L.mcq_workspace_bytes(C.byref(self.p)))
            if self.ws_bytes == 0:
                _check(abi.EINVAL)
            self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.device)

    def launch(self, stream=None):
        """Enqueue init + sweep on `stream` (default: torch's current stream); asynchronous."""
        torch = self.torch
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        with torch.cuda.device(self.device):
            _check(self.L.mcq_run_device(C.byref(self.p), self.seeds.data_ptr(), C.byref(self.out),
                                         self.ws.data_ptr(), self.ws_bytes, C.c_void_p(st.cuda_stream)))

    def launch_timed(self, stream=None):
        """Like launch(), but brackets the init and sweep kernels with HIP events recorded on the launch
        stream and waits for them; returns (init_ms, sweep_ms)."""
        torch = self.torch
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        i_ms, s_ms = C.c_float(0), C.c_float(0)
        with torch.cuda.device(self.device):
            _check(self.L.mcq_run_device_timed(C.byref(self.p), self.seeds.data_ptr(), C.byref(self.out), self.ws.data_ptr(),
                                               self.ws_bytes, C.c_void_p(st.cuda_stream), C.byref(i_ms), C.byref(s_ms)))
        return i_ms.value, s_ms.value

    def trace_stats(self, n_bins=100, stream=None):
        """Per-step sum / sum of squares / count of the energy trace and binned accepted / proposed counts, computed on
        the device from the resident trace (mcq_trace_stats_device).  Bin edges are those of the reference's
        plot_acceptance_rates_binned: np.linspace(0, n_steps, n_bins + 1), left-closed (experiments.py:660-686).
        Returns a dict of NumPy arrays (synchronises)."""
        torch = self.torch
        st = torch.cuda.current_stream(self.device) if stream is None else stream
        n = int(self.p.n_steps)
        edges = np.linspace(0, n, n_bins + 1)
        lo = np.ceil(edges).astype(np.int64)  # step s is in bin b iff edges[b] <= s < edges[b+1]  <=>  ceil(edges[b]) <= s < ceil(edges[b+1])
        lo[-1] = max(n, int(lo[-1]))
        with torch.cuda.device(self.device), torch.cuda.stream(st):
            s_sum = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
            s_sq = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
            s_cnt = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
            b_lo = torch.from_numpy(lo).to(self.device)
            b_acc = torch.zeros(n_bins, dtype=torch.int64, device=self.device)
            b_pro = torch.zeros(n_bins, dtype=torch.int64, device=self.device)
            _check(self.L.mcq_trace_stats_device(C.byref(self.p), C.byref(self.out), s_sum.data_ptr(), s_sq.data_ptr(), s_cnt.data_ptr(),
                                                 n_bins, b_lo.data_ptr(), b_acc.data_ptr(), b_pro.data_ptr(), C.c_void_p(st.cuda_stream)))
        st.synchronize()
        return {"step_sum": s_sum.cpu().numpy(), "step_sumsq": s_sq.cpu().numpy(), "step_count": s_cnt.cpu().numpy(),
                "bin_centers": (edges[:-1] + edges[1:]) / 2, "bin_accepted": b_acc.cpu().numpy(), "bin_proposed": b_pro.cpu().numpy()}

    def results(self):
        """Copy every output back as NumPy arrays (synchronises)."""
        res = {}
        for k, t in self.t.items():
            a = t.cpu().numpy()
            res[k] = a.view(np.uint64) if abi.OUTPUT_DTYPES[k] is np.uint64 else a.view(np.uint32) if abi.OUTPUT_DTYPES[k] is np.uint32 else a
        return res

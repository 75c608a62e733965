"""ctypes loader of libddp_amd.so (the C ABI of include/ddp_amd.h).

There is NO fallback: if the shared library is missing, or no HIP device is present when a handle is
requested, this raises.  Nothing here imports the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DDP_AMD_LIB", os.path.join(_HERE, "libddp_amd.so"))      # the override is for A/B timing of kernel variants

dp = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p

EXPORTS = [
    "ddp_last_error", "ddp_version", "ddp_device_count", "ddp_create", "ddp_create_with_stream", "ddp_destroy", "ddp_sync", "ddp_reload_env", "ddp_last_kernel", "ddp_sh_timeouts", "ddp_sh_timeout_info", "ddp_stream",
    "ddp_malloc", "ddp_free", "ddp_memcpy_h2d", "ddp_memcpy_d2h", "ddp_memset", "ddp_host_alloc", "ddp_host_free", "ddp_host_trim",
    "ddp_event_create", "ddp_event_destroy", "ddp_event_record", "ddp_event_elapsed_ms",
    "ddp_back_pass_f64_dev", "ddp_back_pass_f64", "ddp_boxqp_f64_dev", "ddp_boxqp_f64",
    "ddp_cost_len", "ddp_forward_pass_f64_dev", "ddp_forward_pass_f64", "ddp_df_f64_dev", "ddp_df_f64",
    "ddp_ilqg_default_opts", "ddp_ilqg_f64", "ddp_ilqg_f64_dev", "ddp_ilqg_warm_f64", "ddp_ilqg_warm_f64_dev", "ddp_ilqg_ex_f64", "ddp_ilqg_ex_f64_dev", "ddp_ilqg_set_timing", "ddp_ilqg_queue_f64", "ddp_ilqg_queue_f64_dev", "ddp_ilqg_mpc_f64", "ddp_ilqg_mpc_f64_dev", "ddp_mpc_shift_f64_dev", "ddp_costfun_f64_dev", "ddp_batch_stats_f64_dev",
    "ddp_kl_terms_f64_dev", "ddp_kl_terms_f64", "ddp_back_pass_gps_f64_dev", "ddp_back_pass_gps_f64",
    "ddp_forward_covariance_f64_dev", "ddp_forward_covariance_f64", "ddp_kl_div_f64_dev", "ddp_kl_div_f64",
    "ddp_kl_dual_begin_f64_dev", "ddp_kl_dual_retry_f64_dev", "ddp_kl_dual_update_f64_dev",
    "ddp_ilqgkl_default_opts", "ddp_ilqgkl_f64_dev", "ddp_ilqgkl_f64",
    "ddp_comm_rccl_info", "ddp_comm_unique_id", "ddp_comm_create", "ddp_comm_destroy", "ddp_allreduce_stats_f64_dev",
]


class BPDesc(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("n", "m", "N", "B", "fx_tv", "fx_batched", "cost_tv", "cost_batched",
                                       "regType", "has_lims")]


class KLCostTerms(C.Structure):
    _fields_ = [("cx", vp), ("cu", vp), ("cxx", vp), ("cxu", vp), ("cuu", vp), ("eta", vp), ("eta_tv", C.c_int)]


class KLDual(C.Structure):
    _fields_ = [(k, vp) for k in ("etab", "eta", "del_", "divergence", "satisfied", "status", "live", "pend", "iters", "nback")]


class QPOpts(C.Structure):
    _fields_ = [("maxIter", C.c_int), ("minGrad", C.c_double), ("minRelImprove", C.c_double),
                ("stepDec", C.c_double), ("minStep", C.c_double), ("Armijo", C.c_double)]


class Problem(C.Structure):
    _fields_ = [("kind", C.c_int), ("n", C.c_int), ("m", C.c_int), ("N", C.c_int), ("B", C.c_int),
                ("A", vp), ("Bm", vp), ("dyn_tv", C.c_int), ("dyn_batched", C.c_int), ("Q", vp), ("R", vp),
                ("g", C.c_double), ("l", C.c_double), ("h", C.c_double), ("d", C.c_double),
                ("goal", C.c_double * 4), ("cost_diag", C.c_int), ("diff_wrap", C.c_uint32)]


class ILQGOpts(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("dlambda", C.c_double), ("lambda_factor", C.c_double),
                ("lambda_max", C.c_double), ("lambda_min", C.c_double), ("tol_fun", C.c_double),
                ("tol_grad", C.c_double), ("max_iter", C.c_int), ("regType", C.c_int),
                ("reduce_ratio_min", C.c_double), ("n_alpha", C.c_int), ("alpha", C.c_double * 16)]


class ILQGKLOpts(C.Structure):
    _fields_ = [("kl_step", C.c_double), ("max_iter", C.c_int), ("etabracket", C.c_double * 3), ("del0", C.c_double)]


ILQGKL_NSTATS = 12

_lib = None
hip_runtime_note = "library not loaded yet"


class DDPError(RuntimeError):
    pass


def _share_torch_hip():
    """ONE HIP runtime per process: a PyTorch-ROCm wheel bundles its own libamdhip64 / libhsa-runtime64 (torch/lib), libddp_amd.so is
    linked against the system ROCm.  Whoever loads first decides which copy the other binds to; with the system copy first, a later
    `import torch` finds "No HIP GPUs".  If torch is installed but not imported yet, its bundled runtime is loaded (globally) before the
    library, which is what happens anyway when torch is imported first (bench.py, the tests).  DDP_AMD_SHARE_TORCH_HIP=0 switches this off.
    The preload only happens when the bundled runtime has the SAME major version as the ROCm the library was linked against (the soname
    of its DT_NEEDED libamdhip64.so.N): a torch wheel built for another ROCm major is left alone (the library then runs on the system
    runtime and a later `import torch` may not see the GPU — that combination needs torch imported first).  What was done is recorded in
    `hip_runtime_note` (shown by `ddp_amd.runtime_info()`)."""
    global hip_runtime_note
    import sys
    if "torch" in sys.modules:
        hip_runtime_note = "torch was imported first: its HIP runtime is the process's"
        return
    if os.environ.get("DDP_AMD_SHARE_TORCH_HIP", "1") == "0":
        hip_runtime_note = "DDP_AMD_SHARE_TORCH_HIP=0: system ROCm runtime"
        return
    try:
        import glob
        import importlib.util
        import re
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            hip_runtime_note = "torch not installed: system ROCm runtime"
            return
        tlib = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        cand = os.path.join(tlib, "libamdhip64.so")
        if not os.path.exists(cand):
            hip_runtime_note = "torch has no bundled libamdhip64: system ROCm runtime"
            return
        # major version of the runtime the library wants (its DT_NEEDED entry) and of the one torch bundles (versioned file next to it)
        want = None
        with open(LIB_PATH, "rb") as f:
            m_ = re.search(rb"libamdhip64\.so\.(\d+)", f.read())
            want = int(m_.group(1)) if m_ else None
        have = None
        for fn in glob.glob(cand + ".*"):
            m2 = re.search(r"libamdhip64\.so\.(\d+)", os.path.basename(fn))
            if m2:
                have = int(m2.group(1))
        real = os.path.basename(os.path.realpath(cand))
        m3 = re.search(r"libamdhip64\.so\.(\d+)", real)
        if m3:
            have = int(m3.group(1))
        if want is not None and have is not None and want != have:
            hip_runtime_note = "torch bundles libamdhip64.so.%d, libddp_amd.so wants .so.%d: NOT preloaded (import torch first if both are needed)" % (have, want)
            return
        C.CDLL(cand, mode=C.RTLD_GLOBAL)
        hip_runtime_note = "preloaded torch's bundled HIP runtime (%s) so that a later `import torch` shares it" % real
    except (OSError, ImportError, ValueError) as exc:
        hip_runtime_note = "preload of torch's HIP runtime failed (%s): system ROCm runtime" % exc


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DDPError("libddp_amd.so not built (%s): run `python __graft_entry__.py` or "
                           "`python differentialdynamicprogramming.jl_amd/build.py`; there is no CPU fallback" % LIB_PATH)
        _share_torch_hip()
        L = C.CDLL(LIB_PATH)
        L.ddp_last_error.restype = C.c_char_p
        L.ddp_version.restype = C.c_char_p
        L.ddp_stream.restype = vp
        L.ddp_stream.argtypes = [vp]
        L.ddp_last_kernel.restype = C.c_char_p
        L.ddp_last_kernel.argtypes = [vp, C.c_int]
        for name in EXPORTS:
            fn = getattr(L, name)
            if name not in ("ddp_last_error", "ddp_version", "ddp_stream", "ddp_last_kernel"):
                fn.restype = C.c_int
        L.ddp_ilqg_default_opts.restype = None
        L.ddp_ilqgkl_default_opts.restype = None
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise DDPError("libddp_amd: %s (rc=%d)" % (lib().ddp_last_error().decode(), rc))


def _env_snapshot():
    return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("DDP_")))


class Handle:
    """One HIP stream + scratch on one device (ddp_create / ddp_destroy)."""

    def __init__(self, device=0, stream=None):
        """`stream`: optional raw hipStream_t (int) to adopt, e.g. torch.cuda.current_stream().cuda_stream"""
        self._h = vp()
        if stream is None:
            check(lib().ddp_create(int(device), C.byref(self._h)))
        else:
            check(lib().ddp_create_with_stream(int(device), vp(stream), C.byref(self._h)))
        self.device = device
        self._env = _env_snapshot()

    def close(self):
        if self._h:
            lib().ddp_destroy(self._h)
            self._h = vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def raw(self):
        """the ddp_handle for a C call.  The library reads its DDP_* switches once per handle; a host that changes them (the tests
        do, per case) gets them re-read here, so `os.environ[...] = ...` keeps working as it did when every launch called getenv."""
        snap = _env_snapshot()
        if snap != self._env:
            self._env = snap
            lib().ddp_reload_env(self._h)
        return self._h

    def last_kernel(self, which=0):
        """kernel of the last back_pass (0) / forward_pass (1) dispatch (ddp_last_kernel)"""
        return lib().ddp_last_kernel(self._h, int(which)).decode()

    def sh_timeouts(self):
        """tiles of the shared-operand backward pass that gave their trajectories to the per-trajectory kernels after a timed-out wait"""
        r = lib().ddp_sh_timeouts(self._h)
        if r < 0:
            check(r)
        return r

    def sh_timeout_info(self):
        """what the tiles counted by `sh_timeouts` were waiting for (ddp_sh_timeout_info): a list of dicts — work-group, group, the chunk
        (of 8 time steps) it waited for, the progress word it last saw, ms waited, XCD, groups of that launch, launch number — and the 16
        progress words as they stand now; empty in a healthy run"""
        buf = (C.c_int * 80)()
        r = lib().ddp_sh_timeout_info(self._h, buf, 80)
        if r < 0:
            check(r)
        keys = ("work_group", "group", "chunk", "progress_seen", "waited_ms", "xcc", "groups", "launch")
        recs = [dict(zip(keys, (int(buf[8 * i + e]) for e in range(8)))) for i in range(r)]
        for d in recs:
            d["published_seen"] = d["progress_seen"] & ((1 << 24) - 1); d["finished_seen"] = bool(d["progress_seen"] & (1 << 30))
        return {"records": recs, "progress_now": [int(buf[64 + g]) for g in range(16)]}

    def sync(self):
        check(lib().ddp_sync(self._h))

    # --- device buffers for callers without their own allocator
    def malloc(self, nbytes):
        p = vp()
        check(lib().ddp_malloc(self._h, C.c_size_t(nbytes), C.byref(p)))
        return p

    def free(self, p):
        check(lib().ddp_free(self._h, p))

    def to_device(self, arr):
        arr = np.asfortranarray(arr)
        p = self.malloc(arr.nbytes)
        check(lib().ddp_memcpy_h2d(self._h, p, arr.ctypes.data_as(vp), C.c_size_t(arr.nbytes)))
        return p

    def to_host(self, p, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype, order="F")
        check(lib().ddp_memcpy_d2h(self._h, out.ctypes.data_as(vp), p, C.c_size_t(out.nbytes)))
        return out


_default = {}


def default_handle(device=0):
    if device not in _default:
        _default[device] = Handle(device)
    return _default[device]


class _Pinned:
    """a block of page-locked host memory from the library's cache (ddp_host_alloc); arrays made from it keep it alive"""
    __slots__ = ("ptr", "nbytes", "__weakref__")

    def __init__(self, nbytes):
        p = vp()
        check(lib().ddp_host_alloc(C.c_size_t(max(int(nbytes), 1)), C.byref(p)))
        self.ptr, self.nbytes = p.value, int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                lib().ddp_host_free(vp(self.ptr))
                self.ptr = None
        except Exception:
            pass

    @property
    def __array_interface__(self):
        return {"shape": (max(self.nbytes, 1),), "typestr": "|u1", "data": (self.ptr, False), "version": 3}


def result_array(shape, dtype=np.float64):
    """Fortran-ordered array for a RESULT of a host-pointer call.  Page-locked memory from the library's cache when the array is large
    (>= 1 MB; DDP_PINNED_RESULTS=0: never): the device writes straight into it at link speed, and the block comes back from the cache on
    the next call of the same size instead of fresh pageable pages that fault on first touch.  Contents are unspecified until the call
    has filled the array (every result array is written in full)."""
    shape = tuple(int(s) for s in np.atleast_1d(shape))
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dt.itemsize
    if nbytes < (1 << 20) or os.environ.get("DDP_PINNED_RESULTS") == "0":
        return np.zeros(shape, dtype=dt, order="F")
    blk = _Pinned(nbytes)
    return np.asarray(blk)[:nbytes].view(dt).reshape(shape, order="F")


def f64(a):
    """Fortran-ordered contiguous float64 view/copy (Julia memory layout)."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def ptr(a):
    return None if a is None else a.ctypes.data_as(vp)

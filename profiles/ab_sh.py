"""A/B of the shared-LTI backward pass (csrc/back_pass_sh.hip) against the per-trajectory kernels on the C2 shape, device-resident
operands, HIP events around 40 passes after 10 warm-up passes:  python profiles/ab_sh.py [B ...]"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddp_amd import _lib
from oracle import np_restatement as npr

L = _lib.lib(); h = _lib.default_handle()
n, m, N = 10, 2, 1000
rng = np.random.default_rng(1234)
P = npr.make_lq_problem(rng)


def run(B, sh_min, reps=40, ngroups=1):
    os.environ["DDP_SH_MIN_B"] = str(sh_min)
    cx = 0.01 * rng.standard_normal((n, N, B)); cu = 0.001 * rng.standard_normal((m, N, B))
    lam = np.array([1.0, 0.625, 0.39, 0.1])[np.arange(B) % ngroups]
    d = [h.to_device(x) for x in (cx, cu, P["Q"], np.zeros((n, m)), P["R"], P["A"], P["B"], lam)]
    o = [h.malloc(8 * s * N * B) for s in (m * n, m, m * m, n, n * n)] + [h.malloc(16 * B), h.malloc(4 * B)]
    desc = _lib.BPDesc(n, m, N, B, 0, 0, 0, 0, 1, 0)
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.ddp_event_create(h.raw, C.byref(e0)); L.ddp_event_create(h.raw, C.byref(e1))
    def go():
        _lib.check(L.ddp_back_pass_f64_dev(h.raw, C.byref(desc), *d, None, None, None, *o))
    for _ in range(10):
        go()
    h.sync()
    L.ddp_event_record(h.raw, e0)
    for _ in range(reps):
        go()
    L.ddp_event_record(h.raw, e1)
    ms = C.c_float()
    L.ddp_event_elapsed_ms(h.raw, e0, e1, C.byref(ms))
    for p_ in d + o:
        h.free(p_)
    t = ms.value / reps
    gb = 1184.0 * (N - 1) * B / 1e9
    return t, gb / t


for B in [int(x) for x in sys.argv[1:]] or [1024, 2048, 4096, 8192, 32768]:
    for tag, mn, ng in (("per-trajectory", 1 << 30, 1), ("shared", 1, 1), ("shared 4 groups", 1, 4)):
        t, bw = run(B, mn, ngroups=ng)
        print("B=%6d %-16s back %.4f ms  %.0f GB/s  frac %.3f" % (B, tag, t, bw * 1e3, bw * 1e3 / 8000), flush=True)

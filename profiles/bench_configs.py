"""Pass time (back_pass + forward_pass, one α) of the non-headline BASELINE configs with device-resident operands:
   C3  pendcart n=4 m=1 N=600, control limits (boxQP), B=4096
   C4  large-state LTV n=64 m=8 N=256, per-trajectory time-varying dynamics, ONE time-invariant cost (a2 layout: 75.4 KB per step), B=1024 per GPU
   C4TV the same with cxx, cxu, cuu time-varying and per trajectory (a3 layout, SURVEY 8d's C4: 112.8 KB per step = 29.4 GB per launch)
   C2TV the C2 shape (n=10, m=2, N=1000, B=1024) in the LTV / TV-cost layout (per-trajectory fx, fu, cxx, cxu, cuu; SURVEY 8d: 4.47 MB/pass)
   C5  one pass of the KL-constrained iteration (C3 + KL): back_pass_gps + forward_pass + forward_covariance + kl_div_wiki, B=4096
   offA / offB  shapes BASELINE does not name (the reference's back_pass is size-generic, backward_pass.jl:162-252): n=12 m=3 N=500
       B=2048 with per-trajectory time-varying dynamics (one-tile matrix-core kernel); n=6 m=2 N=1000 B=4096 LTI with control limits (padded row kernel)
   offE  n=48 m=6 N=300 B=1024: between the mid-size kernels and n = 64 (back_pass_mf2 with 3 tiles of 16 states; forward_big_kernel)
   offL  the C2 shape (n=10 m=2 N=1000 B=1024, LTI) with control limits +-0.05 (back_pass_mxg<LIMS>)
   offC / offD  n=24 m=4 and n=32 m=8, N=300, B=1024, per-trajectory time-varying dynamics: the mid-size kernels (back_pass_mid.hip, forward_mid_kernel)
Prints one JSON line per config.  Informational; the graded line is bench.py's."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ddp_amd  # noqa: E402
from ddp_amd import _lib  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
L = _lib.lib()
h = ddp_amd.Handle(0, stream=torch.cuda.current_stream(dev).cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
f64 = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))).to(dev)
empty = lambda c, dt=torch.float64: torch.empty(int(c), dtype=dt, device=dev)


def run(name, prob, n, m, N, B, dx0, du0, lims, regType, fx_desc, steps=10, warmup=2, gen=None, cost_desc=None):
    CL = N + 1 if prob.kind == 1 else N
    one = np.array([1.0])
    dl = f64(lims) if lims is not None else None
    dx, du, dc, dcs = empty(n * N * B), empty(m * N * B), empty(CL * B), empty(B)
    _lib.check(L.ddp_forward_pass_f64_dev(h.raw, C.byref(prob), None, None, p(dx0), p(du0), None, _lib.ptr(one), 1, p(dl) if dl is not None else None,
                                          None, p(dx), p(du), p(dc), p(dcs)))
    dcx, dcu = empty(n * N * B), empty(m * N * B)
    pend = prob.kind == 1
    dfx = empty(n * n * N * B) if pend else None
    dfu = empty(n * m * N * B) if pend else None
    _lib.check(L.ddp_df_f64_dev(h.raw, C.byref(prob), p(dx), p(du), None, p(dcx), p(dcu), p(dfx) if pend else None, p(dfu) if pend else None))
    fx, fu, fx_tv, fx_b = (dfx, dfu, 1, 1) if pend else fx_desc
    dQ, dR = prob._Q, prob._R
    dcxu = torch.zeros(n * m, dtype=torch.float64, device=dev)
    ctv = cb = 0
    if cost_desc is not None:                                   # time-varying, per-trajectory cost Hessians (backward_pass.jl:179-215)
        dQ, dcxu, dR, ctv, cb = cost_desc
    dlam = torch.ones(B, dtype=torch.float64, device=dev)
    dK, dk, dQuu, dVx, dVxx, ddV = empty(m * n * N * B), empty(m * N * B), empty(m * m * N * B), empty(n * N * B), empty(n * n * N * B), empty(2 * B)
    ddiv = torch.zeros(B, dtype=torch.int32, device=dev)
    dxn, dun, dcn, dcsn = empty(n * N * B), empty(m * N * B), empty(CL * B), empty(B)
    desc = _lib.BPDesc(n, m, N, B, fx_tv, fx_b, ctv, cb, regType, int(lims is not None))

    def step(ev=None):
        if ev: L.ddp_event_record(h.raw, ev[0])
        _lib.check(L.ddp_back_pass_f64_dev(h.raw, C.byref(desc), p(dcx), p(dcu), p(dQ), p(dcxu), p(dR), p(fx), p(fu), p(dlam),
                                           p(dl) if dl is not None else None, p(du), None, p(dK), p(dk), p(dQuu), p(dVx), p(dVxx), p(ddV), p(ddiv)))
        if ev: L.ddp_event_record(h.raw, ev[1])
        _lib.check(L.ddp_forward_pass_f64_dev(h.raw, C.byref(prob), p(dK), p(dk), p(dx0), p(du), p(dx), _lib.ptr(one), 1,
                                              p(dl) if dl is not None else None, None, p(dxn), p(dun), p(dcn), p(dcsn)))
        if ev: L.ddp_event_record(h.raw, ev[2])

    steps = int(os.environ.get("DDP_BC_STEPS", steps)); warmup = int(os.environ.get("DDP_BC_WARMUP", warmup))
    if n >= 64:                                                 # C4: a pass is 10 ms, fewer of them
        steps, warmup = max(5, steps // 4), max(1, warmup // 4)     # A/B runs: many steps, clocks settled
    for _ in range(warmup):
        step()
    evs = []
    for _ in range(steps):
        ev = [C.c_void_p() for _ in range(3)]
        for e in ev:
            L.ddp_event_create(h.raw, C.byref(e))
        evs.append(ev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for ev in evs:
        step(ev)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    bp, fp = [], []
    for ev in evs:
        ms = C.c_float()
        L.ddp_event_elapsed_ms(h.raw, ev[0], ev[1], C.byref(ms)); bp.append(ms.value)
        L.ddp_event_elapsed_ms(h.raw, ev[1], ev[2], C.byref(ms)); fp.append(ms.value)
    tv = fx_tv
    bp_bytes = ((n + m) + (n * n + n * m if tv else 0) + (n * n + n * m + m * m if ctv else 0) + (m if lims is not None else 0) + (m * n + m + n + n * n + m * m)) * 8 * (N - 1) * B
    fp_bytes = ((m * n + m + n + m) + (n * n + n * m if (tv and not pend) else 0) + (n + m + 1)) * 8 * N * B
    extra = {}
    if n == 64 and m == 8:
        # C4 sits on the ridge: the binding roofline is the fp64 matrix pipe.  Products per trajectory-step as the kernel
        # issues them: 608 v_mfma_f64_16x16x4 (SQ_INSTS_VALU_MFMA_F64 = 158 760 960 per launch of 1 024 x 255 steps, profiles/r04_c4_pmc.txt;
        # 620 with limits is not counted here), 2 048 flop each; peak 78.6 TFLOP/s dense fp64 matrix (MI355X_MICROARCH.md)
        flop = 608.0 * 2048.0 * (N - 1) * B
        extra["back_pass_roofline_mfma"] = {"bound": "mfma_f64", "achieved_TFs": round(flop / (np.mean(bp) * 1e-3) / 1e12, 2), "peak_TFs": 78.6,
                                           "frac": round(flop / (np.mean(bp) * 1e-3) / 78.6e12, 4), "flop_per_launch": flop}
    out = {"config": name, "n": n, "m": m, "N": N, "batch": B, "iterations_per_s": round(B * steps / el, 1), "ms_per_pass_batch": round(1e3 * el / steps, 3),
           "back_pass_ms": round(float(np.mean(bp)), 3), "forward_ms": round(float(np.mean(fp)), 3),
           "back_pass_ms_median": round(float(np.median(bp)), 4), "back_pass_ms_min": round(float(np.min(bp)), 4),
           "back_pass_alg_bytes_per_launch": int(bp_bytes), "back_pass_alg_GBs": round(bp_bytes / (np.mean(bp) * 1e-3) / 1e9, 1), "back_pass_frac_of_8TBs": round(bp_bytes / (np.mean(bp) * 1e-3) / 8e12, 4),
           "forward_alg_GBs": round(fp_bytes / (np.mean(fp) * 1e-3) / 1e9, 1), "diverged": int(ddiv.sum().item()),
           "back_pass_kernel": h.last_kernel(0), "forward_kernel": h.last_kernel(1)}
    out.update(extra)
    print(json.dumps(out))


def c3(B=4096):
    n, m, N = 4, 1, 600
    prob = _lib.Problem()
    prob.kind, prob.n, prob.m, prob.N, prob.B = 1, n, m, N, B
    Q, R = f64(np.diag([10.0, 1, 2, 1])), f64(np.array([[1.0]]))
    prob.Q, prob.R = Q.data_ptr(), R.data_ptr()
    prob._Q, prob._R = Q, R
    prob.g, prob.l, prob.h, prob.d = 9.82, 0.35, 0.01, 0.99
    prob.cost_diag = 1
    for i, v in enumerate([np.pi, 0, 0, 0]):
        prob.goal[i] = v
    rng = np.random.default_rng(0)
    x0 = np.tile(np.array([np.pi - 0.6, 0, 0, 0])[:, None], (1, B)); x0[0] += rng.uniform(-0.1, 0.1, B)
    u0 = 2.0 * np.sin(np.arange(N) / 37.0)[None, :, None] * np.ones((1, 1, B))
    nolims = os.environ.get("DDP_C3_NOLIMS") == "1"            # A/B: what the control limits (boxQP) cost
    run("C3 pendcart" + (" (no lims)" if nolims else " lims"), prob, n, m, N, B, f64(x0), f64(u0),
        None if nolims else float(os.environ.get("DDP_C3_LIMSCALE", "5.0")) * np.array([[-1.0, 1.0]]), 2, None)


def c2tv(B=1024):
    """the C2 shape (n=10, m=2, N=1000) in the LTV / TV-cost layout of backward_pass.jl:179-215 — fx, fu, cxx, cxu, cuu
    time-varying and per trajectory, 4.47 MB per pass and trajectory: the north star's "coalesced HBM loads of fx/fu/cxx/cuu" at n = 10"""
    import scipy.linalg as sla
    n, m, N = 10, 2, 1000
    rng = np.random.default_rng(1234)
    torch.manual_seed(2)
    h_ = 0.01
    a0 = rng.standard_normal((n, n))
    A = sla.expm(h_ * (a0 - a0.T))
    Bm = h_ * rng.standard_normal((n, m))
    NB = N * B
    dA = (f64(A).reshape(1, -1) * (1.0 + 0.01 * torch.rand(NB, 1, dtype=torch.float64, device=dev))).reshape(-1).contiguous()
    dB = (f64(Bm).reshape(1, -1) * (1.0 + 0.01 * torch.rand(NB, 1, dtype=torch.float64, device=dev))).reshape(-1).contiguous()
    sc = 1.0 + 0.1 * torch.rand(NB, 1, dtype=torch.float64, device=dev)
    dcxx = (f64(h_ * np.eye(n)).reshape(1, -1) * sc).reshape(-1).contiguous()
    dcxu = torch.zeros(n * m * NB, dtype=torch.float64, device=dev)
    dcuu = (f64(0.1 * h_ * np.eye(m)).reshape(1, -1) * sc).reshape(-1).contiguous()
    prob = _lib.Problem()
    prob.kind, prob.n, prob.m, prob.N, prob.B = 0, n, m, N, B
    Q, R = f64(h_ * np.eye(n)), f64(0.1 * h_ * np.eye(m))
    prob.A, prob.Bm, prob.Q, prob.R = dA.data_ptr(), dB.data_ptr(), Q.data_ptr(), R.data_ptr()
    prob._Q, prob._R = Q, R
    prob.dyn_tv, prob.dyn_batched, prob.cost_diag = 1, 1, 1
    rng = np.random.default_rng(1000)
    x0 = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B))
    u0 = 0.1 * rng.standard_normal((m, N, B))
    run("C2 LTV / TV-cost layout", prob, n, m, N, B, f64(x0), f64(u0), None, 1, (dA, dB, 1, 1), cost_desc=(dcxx, dcxu, dcuu, 1, 1))


def off_shape(tag, n, m, N, B, ltv, lims):
    """a shape with no exact instantiation: LQ problem of demo_linear's recipe at (n, m); ltv: fx, fu time-varying per trajectory"""
    import scipy.linalg as sla
    rng = np.random.default_rng(4321 + n)
    torch.manual_seed(3)
    h_ = 0.01
    a0 = rng.standard_normal((n, n))
    A = sla.expm(h_ * (a0 - a0.T))
    Bm = h_ * rng.standard_normal((n, m))
    prob = _lib.Problem()
    prob.kind, prob.n, prob.m, prob.N, prob.B = 0, n, m, N, B
    Q, R = f64(h_ * np.eye(n)), f64(0.1 * h_ * np.eye(m))
    if ltv:
        NB = N * B
        dA = (f64(A).reshape(1, -1) * (1.0 + 0.01 * torch.rand(NB, 1, dtype=torch.float64, device=dev))).reshape(-1).contiguous()
        dB = (f64(Bm).reshape(1, -1) * (1.0 + 0.01 * torch.rand(NB, 1, dtype=torch.float64, device=dev))).reshape(-1).contiguous()
        prob.dyn_tv, prob.dyn_batched = 1, 1
        fxd = (dA, dB, 1, 1)
    else:
        dA, dB = f64(A), f64(Bm)
        fxd = (dA, dB, 0, 0)
    prob.A, prob.Bm, prob.Q, prob.R = dA.data_ptr(), dB.data_ptr(), Q.data_ptr(), R.data_ptr()
    prob._Q, prob._R = Q, R
    prob._keep = (dA, dB)
    x0 = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B))
    u0 = 0.1 * rng.standard_normal((m, N, B))
    L_ = None if not lims else float(os.environ.get("DDP_OFF_LIMS", "0.05")) * np.stack([-np.ones(m), np.ones(m)], 1)
    run("%s n=%d m=%d %s%s" % (tag, n, m, "LTV per-trajectory dynamics" if ltv else "LTI", " lims" if lims else ""), prob, n, m, N, B, f64(x0), f64(u0),
        L_, 1, fxd)


def c5(B=4096):
    """one pass of the KL-constrained iteration at the C5 shape (pendcart n=4, m=1, N=600, limits): back_pass_gps + forward_pass +
    forward_covariance + kl_div_wiki, each bracketed by HIP events (iLQGkl.jl:100,132,133, klutils.jl:114)"""
    n, m, N = 4, 1, 600
    prob = _lib.Problem()
    prob.kind, prob.n, prob.m, prob.N, prob.B = 1, n, m, N, B
    Q, R = f64(np.diag([10.0, 1, 2, 1])), f64(np.array([[1.0]]))
    prob.Q, prob.R = Q.data_ptr(), R.data_ptr()
    prob.g, prob.l, prob.h, prob.d = 9.82, 0.35, 0.01, 0.99
    prob.cost_diag = 1
    for i, v in enumerate([np.pi, 0, 0, 0]):
        prob.goal[i] = v
    rng = np.random.default_rng(0)
    x0 = np.tile(np.array([np.pi - 0.6, 0, 0, 0])[:, None], (1, B)); x0[0] += rng.uniform(-0.1, 0.1, B)
    u0 = 2.0 * np.sin(np.arange(N) / 37.0)[None, :, None] * np.ones((1, 1, B)) + 0.05 * rng.standard_normal((1, N, B))
    dx0, du = f64(x0), f64(u0)
    dl = f64(5.0 * np.array([[-1.0, 1.0]]))
    one = np.array([1.0])
    NB = N * B
    dx, dun, dc, dcs = empty(n * NB), empty(m * NB), empty((N + 1) * B), empty(B)
    _lib.check(L.ddp_forward_pass_f64_dev(h.raw, C.byref(prob), None, None, p(dx0), p(du), None, _lib.ptr(one), 1, p(dl), None, p(dx), p(dun), p(dc), p(dcs)))
    dcx, dcu, dfx, dfu = empty(n * NB), empty(m * NB), empty(n * n * NB), empty(n * m * NB)
    _lib.check(L.ddp_df_f64_dev(h.raw, C.byref(prob), p(dx), p(du), None, p(dcx), p(dcu), p(dfx), p(dfu)))
    cxx = f64(np.repeat(np.diag([10.0, 1, 2, 1])[:, :, None], N, 2)); cxu = torch.zeros(n * m * N, dtype=torch.float64, device=dev)
    cuu = torch.ones(m * m * N, dtype=torch.float64, device=dev)
    Kp, kz, Sp = torch.zeros(m * n * NB, dtype=torch.float64, device=dev), torch.zeros(m * NB, dtype=torch.float64, device=dev), torch.ones(m * m * NB, dtype=torch.float64, device=dev)
    kl = [empty(n * NB), empty(m * NB), empty(n * n * NB), empty(m * n * NB), empty(m * m * NB)]
    _lib.check(L.ddp_kl_terms_f64_dev(h.raw, n, m, N, B, p(Kp), p(kz), p(Sp), *map(p, kl)))
    eta = torch.ones(B, dtype=torch.float64, device=dev)
    terms = _lib.KLCostTerms(*[t.data_ptr() for t in kl], eta.data_ptr(), 0)
    desc = _lib.BPDesc(n, m, N, B, 1, 1, 1, 0, 1, 1)
    dK, dk, dQuu, dQuui, dVx, dVxx, ddV = empty(m * n * NB), empty(m * NB), empty(m * m * NB), empty(m * m * NB), empty(n * NB), empty(n * n * NB), empty(2 * B)
    ddiv = torch.zeros(B, dtype=torch.int32, device=dev)
    dxn = empty(n * NB)
    R1 = f64(1e-3 * np.eye(n))
    sig, kld, klm = empty((n + m) ** 2 * NB), empty(NB), empty(B)

    def step(ev=None):
        if ev: L.ddp_event_record(h.raw, ev[0])
        _lib.check(L.ddp_back_pass_gps_f64_dev(h.raw, C.byref(desc), p(dcx), p(dcu), p(cxx), p(cxu), p(cuu), p(dfx), p(dfu), C.byref(terms), p(dl), p(du),
                                               None, p(dK), p(dk), p(dQuu), p(dQuui), p(dVx), p(dVxx), p(ddV), p(ddiv)))
        if ev: L.ddp_event_record(h.raw, ev[1])
        _lib.check(L.ddp_forward_pass_f64_dev(h.raw, C.byref(prob), p(dK), p(dk), p(dx0), p(du), p(dx), _lib.ptr(one), 1, p(dl), None, p(dxn), p(dun), p(dc), p(dcs)))
        if ev: L.ddp_event_record(h.raw, ev[2])
        _lib.check(L.ddp_forward_covariance_f64_dev(h.raw, n, m, N, B, p(dfx), 1, p(R1), p(dK), p(dQuui), p(sig)))
        if ev: L.ddp_event_record(h.raw, ev[3])
        _lib.check(L.ddp_kl_div_f64_dev(h.raw, n, m, N, B, p(dxn), p(dx), p(sig), p(dK), p(dk), p(dQuui), p(Kp), p(kz), p(Sp), p(Sp), p(kld), p(klm)))
        if ev: L.ddp_event_record(h.raw, ev[4])

    steps, warmup = int(os.environ.get("DDP_BC_STEPS", 10)), int(os.environ.get("DDP_BC_WARMUP", 2))
    for _ in range(warmup):
        step()
    evs = []
    for _ in range(steps):
        ev = [C.c_void_p() for _ in range(5)]
        for e in ev:
            L.ddp_event_create(h.raw, C.byref(e))
        evs.append(ev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for ev in evs:
        step(ev)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = np.zeros((steps, 4))
    for i, ev in enumerate(evs):
        for j in range(4):
            v = C.c_float()
            L.ddp_event_elapsed_ms(h.raw, ev[j], ev[j + 1], C.byref(v)); ms[i, j] = v.value
    mean = ms.mean(axis=0)
    # algorithmic doubles per step and trajectory (the [.,.,N] cost Hessians are shared by the batch and stay in the caches)
    bpb = ((n + m) + (n * n + n * m) + (n + m + n * n + m * n + m * m) + m + (m * n + m + 2 * m * m + n + n * n)) * 8 * (N - 1) * B
    fpb = ((m * n + m + n + m) + (n + m + 1)) * 8 * N * B
    fcb = (n * n + m * n + m * m + (n + m) ** 2) * 8 * N * B
    klb = (2 * n + (n + m) ** 2 + 2 * (m * n + m) + 3 * m * m + 1) * 8 * N * B
    print(json.dumps({"config": "C5 KL pass (pendcart lims + KL)", "n": n, "m": m, "N": N, "batch": B, "iterations_per_s": round(B * steps / el, 1),
                      "ms_per_pass_batch": round(1e3 * el / steps, 3), "back_pass_ms": round(float(mean[0]), 3), "forward_ms": round(float(mean[1]), 3),
                      "forward_covariance_ms": round(float(mean[2]), 3), "kl_div_ms": round(float(mean[3]), 3),
                      "back_pass_ms_median": round(float(np.median(ms[:, 0])), 4), "back_pass_ms_min": round(float(ms[:, 0].min()), 4),
                      "back_pass_alg_bytes_per_launch": int(bpb), "back_pass_alg_GBs": round(bpb / (mean[0] * 1e-3) / 1e9, 1),
                      "back_pass_frac_of_8TBs": round(bpb / (mean[0] * 1e-3) / 8e12, 4), "forward_alg_GBs": round(fpb / (mean[1] * 1e-3) / 1e9, 1),
                      "forward_covariance_alg_GBs": round(fcb / (mean[2] * 1e-3) / 1e9, 1), "kl_div_alg_GBs": round(klb / (mean[3] * 1e-3) / 1e9, 1),
                      "diverged": int((ddiv != 0).sum().item())}))


def c4(B=1024, tv_cost=False):
    """tv_cost=False: per-trajectory time-varying DYNAMICS with one time-invariant Q, R for the batch (the a2 method, backward_pass.jl:162-177:
    9 424 doubles per step); tv_cost=True: SURVEY 8(d)'s C4 — cxx, cxu, cuu time-varying and per trajectory as well (the a3 method,
    backward_pass.jl:179-215: 14 096 doubles = 112.8 KB per step and trajectory, 29.4 GB per launch at B = 1 024)"""
    import scipy.linalg as sla
    n, m, N = 64, 8, 256
    rng = np.random.default_rng(1)
    torch.manual_seed(1)                                        # the per-trajectory perturbation of A below: same problem in every run
    h_ = 0.01
    a0 = rng.standard_normal((n, n))
    A = sla.expm(h_ * (a0 - a0.T))
    Bm = h_ * rng.standard_normal((n, m))
    # per-trajectory, time-varying layout (a3): fx[n,n,N,B] — built on the device to avoid a 8.6 GB host array
    dA = f64(A).reshape(1, -1) * (1.0 + 0.01 * torch.rand(N * B, 1, dtype=torch.float64, device=dev))
    dB = f64(Bm).reshape(1, -1) * torch.ones(N * B, 1, dtype=torch.float64, device=dev)
    dA, dB = dA.reshape(-1).contiguous(), dB.reshape(-1).contiguous()
    prob = _lib.Problem()
    prob.kind, prob.n, prob.m, prob.N, prob.B = 0, n, m, N, B
    Q, R = f64(h_ * np.eye(n)), f64(0.1 * h_ * np.eye(m))
    prob.A, prob.Bm, prob.Q, prob.R = dA.data_ptr(), dB.data_ptr(), Q.data_ptr(), R.data_ptr()
    prob._Q, prob._R = Q, R
    prob.dyn_tv, prob.dyn_batched = 1, 1
    x0 = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B))
    u0 = 0.1 * rng.standard_normal((m, N, B))
    c4lims = os.environ.get("DDP_C4_LIMS")                      # e.g. DDP_C4_LIMS=0.05: control limits ±0.05 (boxQP path)
    lims = None if not c4lims else float(c4lims) * np.stack([-np.ones(m), np.ones(m)], 1)
    if tv_cost:
        # the reference's a3 layout: cxx[n,n,N], cxu[n,m,N], cuu[m,m,N] per trajectory (8.6 GB + 1.1 GB + 0.13 GB at B = 1 024), built on the device
        sc = 1.0 + 0.1 * torch.rand(N * B, 1, dtype=torch.float64, device=dev)
        dcxx = (f64(h_ * np.eye(n)).reshape(1, -1) * sc).reshape(-1).contiguous()
        dcxu = (f64(1e-3 * h_ * rng.standard_normal((n, m))).reshape(1, -1) * sc).reshape(-1).contiguous()
        dcuu = (f64(0.1 * h_ * np.eye(m)).reshape(1, -1) * sc).reshape(-1).contiguous()
        run("C4TV large-state LTV, time-varying per-trajectory cost (a3 layout)" + (" lims" if c4lims else ""), prob, n, m, N, B, f64(x0), f64(u0), lims, 1,
            (dA, dB, 1, 1), steps=5, warmup=1, cost_desc=(dcxx, dcxu, dcuu, 1, 1))
        return
    run("C4 large-state LTV, one time-invariant cost (a2 layout)" + (" lims" if c4lims else ""), prob, n, m, N, B, f64(x0), f64(u0), lims, 1, (dA, dB, 1, 1), steps=5, warmup=1)
    if os.environ.get("DDP_C4_SOLVE", "1") == "1":
        solve("C4 full iLQG solves (device-resident driver)", prob, n, m, N, B, f64(x0), f64(u0), nalpha=4)


def solve(name, prob, n, m, N, B, dx0, du0, nalpha=11, max_iter=50):
    """whole iLQG solves through ddp_ilqg_f64_dev with device buffers; phase split from the :time_* trace keys"""
    o = _lib.ILQGOpts()
    L.ddp_ilqg_default_opts(C.byref(o))
    o.max_iter = max_iter
    al = 10.0 ** np.linspace(0, -3, nalpha)
    o.n_alpha = nalpha
    for i, a in enumerate(al):
        o.alpha[i] = a
    CL = N
    x, u = empty(n * N * B), empty(m * N * B)
    K, k, Quu = empty(m * n * N * B), empty(m * N * B), empty(m * m * N * B)
    Vx, Vxx, cost, stats = empty(n * N * B), empty(n * n * N * B), empty(CL * B), empty(8 * B)
    cap = 4 * max_iter + 1000
    timing = np.full((3, cap), np.nan)
    git = C.c_int(0)
    _lib.check(L.ddp_ilqg_set_timing(h.raw, _lib.ptr(timing), cap))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(L.ddp_ilqg_f64_dev(h.raw, C.byref(prob), C.byref(o), p(dx0), p(du0), None, p(x), p(u), p(K), p(k), p(Quu), p(Vx), p(Vxx),
                                  p(cost), p(stats), 0, None, C.byref(git)))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    L.ddp_ilqg_set_timing(h.raw, None, 0)
    st = stats.cpu().numpy().reshape(B, 8).T if False else stats.cpu().numpy().reshape(8, B, order="F")
    g = git.value
    print(json.dumps({"config": name, "batch": B, "seconds": round(el, 4), "batch_iterations": g,
                      "mean_iterations": round(float(st[1].mean()), 2), "exit_reasons": {int(a): int(b) for a, b in zip(*np.unique(st[0].astype(int), return_counts=True))},
                      "time_derivs_s": round(float(np.nansum(timing[0, :g])), 4), "time_backward_s": round(float(np.nansum(timing[1, :g])), 4),
                      "time_forward_s": round(float(np.nansum(timing[2, :g])), 4), "n_alpha": nalpha, "mean_cost": round(float(st[7].mean()), 4)}))


if __name__ == "__main__":
    which = sys.argv[1:] or ["c3", "c4"]
    if "c3" in which:
        c3(int(os.environ.get("DDP_C3_B", 4096)))
    if "c2tv" in which:
        c2tv()
    if "c4" in which:
        c4()
    if "c4tv" in which:
        c4(tv_cost=True)
    if "c5" in which:
        c5()
    if "offA" in which:
        off_shape("offA", 12, 3, 500, 2048, True, False)
    if "offC" in which:                                          # between the row kernels (n <= 14) and n = 64: which kernel should own 15 <= n <= 32?
        off_shape("offC", int(os.environ.get("DDP_OFFC_N", 24)), int(os.environ.get("DDP_OFFC_M", 4)), int(os.environ.get("DDP_OFFC_T", 300)),
                  int(os.environ.get("DDP_OFFC_B", 1024)), os.environ.get("DDP_OFFC_LTI") != "1", False)
    if "offL" in which:                                          # the C2 shape WITH control limits (iLQG(...; lims) on the LQ example): box-QP on the one-tile kernel
        off_shape("offL", 10, 2, 1000, 1024, False, True)
    if "offE" in which:                                          # 32 < n < 64: the run-time-sized matrix-core kernel back_pass_mf2 (round 5: embedded in the (64, 8) kernel through 23 GB of padded copies)
        off_shape("offE", 48, 6, 300, 1024, True, False)
    if "offD" in which:                                          # the 8 x 8 control system (4 < m <= 8): one coordinate per lane, gains on the matrix cores
        off_shape("offD", 32, 8, 300, 1024, True, False)
    if "offX" in which:                                          # any shape from the environment (experiments): DDP_OFFX="n m N B ltv lims"
        n_, m_, N_, B_, ltv_, lims_ = (int(v) for v in os.environ.get("DDP_OFFX", "10 2 1000 1024 0 1").split())
        off_shape("offX", n_, m_, N_, B_, bool(ltv_), bool(lims_))
    if "offB" in which:
        off_shape("offB", 6, 2, 1000, 4096, False, os.environ.get("DDP_OFF_NOLIMS") != "1")

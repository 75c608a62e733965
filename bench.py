#!/usr/bin/env python
"""bench.py — iLQG backward+forward pass throughput on MI355X (the BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input, with every operand already
resident in HBM:  back_pass (N-1 Riccati steps)  +  forward_pass (N-step closed-loop rollout, one α)
for B = 1024 independent trajectories of BASELINE config 2 (demo_linear: n=10, m=2, N=1000, LTI,
no control limits, regType 1).  With N GPUs every rank owns its own B trajectories (weak scaling,
no data-path exchange); the only collective is one all-reduce (RCCL) of a 4-double statistics vector
per step (Σ new cost, Σ expected reduction terms, #diverged) — the line-search cost reduction of
SURVEY.md §8(e).

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     — dominant kernel (back_pass_kernel): algorithmic bytes / HIP-event time vs 8 TB/s
  cpu_baseline — the CPU oracle (C restatement of the reference, single thread) on a bounded sample
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def make_workload(seed, n, m, N, B):
    """SURVEY.md §8(d) C2 inputs (mirrors src/demo_linear.jl:8-26) with NumPy's generator."""
    import scipy.linalg as sla
    rng = np.random.default_rng(1234)                     # A, B shared by every rank (LTI)
    h = 0.01
    A0 = rng.standard_normal((n, n))
    A = sla.expm(h * (A0 - A0.T))
    Bm = h * rng.standard_normal((n, m))
    Q = h * np.eye(n)
    R = 0.1 * h * np.eye(m)
    rng = np.random.default_rng(seed)                     # per-rank trajectories
    x0 = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B))
    u0 = 0.1 * rng.standard_normal((m, N, B))
    return A, Bm, Q, R, x0, u0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="trajectories per GPU (BASELINE config 2: 1024)")
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="trajectories for the CPU baseline (0 = auto ~10-20 s)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import ddp_amd
    from ddp_amd import _lib
    L = _lib.lib()
    # the library launches on torch's current stream, so RCCL collectives and kernels are stream-ordered
    stream = torch.cuda.current_stream(dev).cuda_stream
    h = ddp_amd.Handle(local, stream=stream)

    n, m, N, B = 10, 2, args.horizon, args.batch
    A, Bm, Q, R, x0, u0 = make_workload(1000 + rank, n, m, N, B)

    def dev_f64(a):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))).to(dev)

    def empty(count, dtype=torch.float64):
        return torch.empty(count, dtype=dtype, device=dev)

    dA, dB, dQ, dR, dx0, du0 = map(dev_f64, (A, Bm, Q, R, x0, u0))
    dcxu = torch.zeros(n * m, dtype=torch.float64, device=dev)
    dlam = torch.ones(B, dtype=torch.float64, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())

    prob = _lib.Problem()
    prob.kind, prob.n, prob.m, prob.N, prob.B = 0, n, m, N, B
    prob.A, prob.Bm, prob.Q, prob.R = dA.data_ptr(), dB.data_ptr(), dQ.data_ptr(), dR.data_ptr()
    prob.dyn_tv, prob.dyn_batched = 0, 0

    # nominal trajectory + its derivatives (outside the timed region: STEP 1 of the iteration)
    dx, du, dc, dcs = empty(n * N * B), empty(m * N * B), empty(N * B), empty(B)
    one = np.array([1.0])
    _lib.check(L.ddp_forward_pass_f64_dev(h.raw, C.byref(prob), None, None, p(dx0), p(du0), None, _lib.ptr(one), 1, None, None,
                                          p(dx), p(du), p(dc), p(dcs)))
    dcx, dcu = empty(n * N * B), empty(m * N * B)
    _lib.check(L.ddp_df_f64_dev(h.raw, C.byref(prob), p(dx), p(du), None, p(dcx), p(dcu), None, None))

    dK, dk, dQuu = empty(m * n * N * B), empty(m * N * B), empty(m * m * N * B)
    dVx, dVxx, ddV = empty(n * N * B), empty(n * n * N * B), empty(2 * B)
    ddiv = torch.zeros(B, dtype=torch.int32, device=dev)
    dxn, dun, dcn, dcsn = empty(n * N * B), empty(m * N * B), empty(N * B), empty(B)
    stats = torch.zeros(4, dtype=torch.float64, device=dev)
    desc = _lib.BPDesc(n, m, N, B, 0, 0, 0, 0, 1, 0)

    def step(ev=None):
        if ev is not None:
            _lib.check(L.ddp_event_record(h.raw, ev[0]))
        _lib.check(L.ddp_back_pass_f64_dev(h.raw, C.byref(desc), p(dcx), p(dcu), p(dQ), p(dcxu), p(dR), p(dA), p(dB), p(dlam),
                                           None, None, None, p(dK), p(dk), p(dQuu), p(dVx), p(dVxx), p(ddV), p(ddiv)))
        if ev is not None:
            _lib.check(L.ddp_event_record(h.raw, ev[1]))
        _lib.check(L.ddp_forward_pass_f64_dev(h.raw, C.byref(prob), p(dK), p(dk), p(dx0), p(du), p(dx), _lib.ptr(one), 1, None,
                                              None, p(dxn), p(dun), p(dcn), p(dcsn)))
        if ev is not None:
            _lib.check(L.ddp_event_record(h.raw, ev[2]))
        if world > 1:
            # the single collective of the path: batch-level line-search statistics (latency-bound, 32 B)
            stats[0] = dcsn.sum(); stats[1] = ddV[0::2].sum(); stats[2] = ddV[1::2].sum(); stats[3] = ddiv.sum()
            dist.all_reduce(stats)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    events = []
    for _ in range(args.steps):
        ev = [C.c_void_p() for _ in range(3)]
        for e in ev:
            _lib.check(L.ddp_event_create(h.raw, C.byref(e)))
        events.append(ev)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    bp_ms, fp_ms = [], []
    for ev in events:
        ms = C.c_float(0)
        _lib.check(L.ddp_event_elapsed_ms(h.raw, ev[0], ev[1], C.byref(ms))); bp_ms.append(ms.value)
        _lib.check(L.ddp_event_elapsed_ms(h.raw, ev[1], ev[2], C.byref(ms))); fp_ms.append(ms.value)
        for e in ev:
            L.ddp_event_destroy(h.raw, e)
    ndiv = int(ddiv.sum().item())
    assert ndiv == 0, "synthetic LQ batch must not diverge"
    zsum = float(dcsn.sum().item())
    assert np.isfinite(zsum)

    # ---- roofline of the dominant kernel (back_pass): algorithmic bytes of SURVEY.md §8(d) / DESIGN.md
    bp_read = (n + m) * 8                                  # cx_i, cu_i per step (LTI, time-invariant cost, no limits)
    bp_write = (m * n + m + n + n * n + m * m) * 8         # K_i, k_i, Vx_i, Vxx_i, Quu_i
    fp_bytes = ((m * n + m + n + m) + (n + m + 1)) * 8     # forward: reads K,k,x,u, writes xnew,unew,c
    bp_bytes_launch = (bp_read + bp_write) * (N - 1) * B
    bp_avg_ms = float(np.mean(bp_ms))
    fp_avg_ms = float(np.mean(fp_ms))
    achieved = bp_bytes_launch / (bp_avg_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get("back_pass_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "back_pass_kernel<10,2,LTI>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "bytes_per_launch": bp_bytes_launch, "avg_launch_ms": round(bp_avg_ms, 4),
                "forward_kernel": {"avg_launch_ms": round(fp_avg_ms, 4), "bytes_per_launch": fp_bytes * N * B,
                                   "achieved_GBs": round(fp_bytes * N * B / (fp_avg_ms * 1e-3) / 1e9, 1)},
                "pass_bytes": (bp_read + bp_write + fp_bytes) * N,
                "note": "each trajectory is a length-N dependency chain: at B=1024 (1 wave per SIMD) the fraction is "
                        "latency/occupancy-limited, not bandwidth-limited"}

    out = None
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(A, Bm, Q, R, x0, dx, du, dcx, dcu, n, m, N, B, args.cpu_sample)
        value = B * world * args.steps / elapsed
        out = {"metric": "iLQG iterations/sec (backward+forward, n=10 m=2 T=1000)", "value": round(value, 1),
               "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE config 2: demo_linear LTI n=10 m=2 N=%d, batch=%d trajectories per GPU, no control "
                                      "limits, regType=1, lambda=1; one step = back_pass + forward_pass(alpha=1) over the batch" % (N, B),
                          "batch_per_gpu": B, "n": n, "m": m, "N": N, "sharding": "batch (independent trajectories), "
                          "one 32-byte RCCL all-reduce of line-search statistics per step when n_gpus>1"},
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


def cpu_baseline(A, Bm, Q, R, x0, dx, du, dcx, dcu, n, m, N, B, sample):
    """Times the CPU oracle (C restatement of the reference's back_pass + forward_pass, single thread —
    the reference is single-threaded; Julia itself is not installed) on a bounded sample of the same
    workload.  Checker/baseline only — never on the measured GPU path."""
    from oracle import oracle_ctypes as oc
    lib = oc.lib()
    prob = oc.make_problem("lq", n, m, N, A=A, B=Bm, Q=Q, R=R)
    x = dx.cpu().numpy().reshape((n, N, B), order="F")
    u = du.cpu().numpy().reshape((m, N, B), order="F")
    cx = dcx.cpu().numpy().reshape((n, N, B), order="F")
    cu = dcu.cpu().numpy().reshape((m, N, B), order="F")

    def run(S):
        f = lambda a: np.asfortranarray(a[..., :S])
        xs, us, cxs, cus, x0s = f(x), f(u), f(cx), f(cu), f(x0)
        K = np.zeros((m, n, N, S), order="F"); k = np.zeros((m, N, S), order="F"); Quu = np.zeros((m, m, N, S), order="F")
        Vx = np.zeros((n, N, S), order="F"); Vxx = np.zeros((n, n, N, S), order="F"); dV = np.zeros((2, S), order="F")
        xn = np.zeros((n, N, S), order="F"); un = np.zeros((m, N, S), order="F"); cn = np.zeros((N, S), order="F")
        cxu = np.zeros((n, m), order="F")
        P = oc._p
        t0 = time.perf_counter()
        nd = lib.ddp_oracle_pass_batch_lq(C.byref(prob), S, P(cxs), P(cus), P(oc._f(Q)), P(cxu), P(oc._f(R)), C.c_double(1.0), 1,
                                          P(x0s), P(us), P(xs), C.c_double(1.0), P(K), P(k), P(Quu), P(Vx), P(Vxx), P(dV), P(xn),
                                          P(un), P(cn))
        return time.perf_counter() - t0, nd

    t_probe, _ = run(32)
    S = sample if sample > 0 else int(min(B, max(64, 15.0 / (t_probe / 32))))
    S = min(S, B)
    t, nd = run(S)
    return {"value": round(S / t, 1), "unit": "iterations/s", "cores": 1, "kind": "port",
            "sample": "%d of the %d trajectories of the same workload, 1 backward + 1 forward pass each, %.1f s on one host core; "
                      "C restatement of the reference (oracle/ddp_oracle.c), NOT Julia (no Julia toolchain in the image)" % (S, B, t)}


if __name__ == "__main__":
    main()

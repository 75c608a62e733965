#!/usr/bin/env python
"""bench.py — iLQG backward+forward pass throughput on MI355X (the BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input, with every operand already
resident in HBM:  back_pass (N-1 Riccati steps)  +  forward_pass (N-step closed-loop rollout, one α, and
its cost) for B = 1024 independent trajectories of BASELINE config 2 (demo_linear: n=10, m=2, N=1000, LTI,
no control limits, regType 1).  With N GPUs every rank owns its own B trajectories (weak scaling, no
data-path exchange); the only collective is one all-reduce (RCCL) of a 4-double statistics vector per
step (Σ new cost, Σ expected-reduction terms, #diverged) — the line-search cost reduction of SURVEY.md §8(e).

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     — dominant kernel (the back_pass kernel): algorithmic bytes / HIP-event time vs 8 TB/s
  cpu_baseline — the CPU oracle (C restatement of the reference, single thread) on a bounded sample
  machine_filling — the same pass at a batch that fills the GPU (outside the timed region, informational)
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
N_STATE, N_CTRL = 10, 2


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
    u0 = (0.1 * rng.standard_normal((m, N, B), dtype=np.float32)).astype(np.float64) if B > 8192 else 0.1 * rng.standard_normal((m, N, B))
    return A, Bm, Q, R, x0, u0


class PassBench:
    """Device-resident state of one batch + the step() that is timed."""

    def __init__(self, torch, dev, handle, lib, rank, n, m, N, B):
        from ddp_amd import _lib
        self.torch, self.dev, self.h, self.L, self._lib = torch, dev, handle, lib, _lib
        self.n, self.m, self.N, self.B = n, m, N, B
        A, Bm, Q, R, x0, u0 = make_workload(1000 + rank, n, m, N, B)
        self.host = (A, Bm, Q, R, x0)

        def dev_f64(a):
            return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="F"))).to(dev)

        def empty(count, dtype=torch.float64):
            return torch.empty(count, dtype=dtype, device=dev)

        self.dA, self.dB, self.dQ, self.dR, self.dx0, du0 = map(dev_f64, (A, Bm, Q, R, x0, u0))
        del u0
        self.dcxu = torch.zeros(n * m, dtype=torch.float64, device=dev)
        self.dlam = torch.ones(B, dtype=torch.float64, device=dev)
        self.p = lambda t: C.c_void_p(t.data_ptr())
        p = self.p
        prob = _lib.Problem()
        prob.kind, prob.n, prob.m, prob.N, prob.B = 0, n, m, N, B
        prob.A, prob.Bm, prob.Q, prob.R = self.dA.data_ptr(), self.dB.data_ptr(), self.dQ.data_ptr(), self.dR.data_ptr()
        prob.dyn_tv, prob.dyn_batched = 0, 0
        prob.cost_diag = 1                                    # Q = h·I, R = 0.1h·I (src/demo_linear.jl:17-18): cost inside the rollout kernel
        self.prob = prob
        self.one = np.array([1.0])
        # nominal trajectory + its derivatives (outside the timed region: STEP 1 of the iteration)
        self.dx, self.du, dc, dcs = empty(n * N * B), empty(m * N * B), empty(N * B), empty(B)
        _lib.check(lib.ddp_forward_pass_f64_dev(handle.raw, C.byref(prob), None, None, p(self.dx0), p(du0), None, _lib.ptr(self.one), 1,
                                                None, None, p(self.dx), p(self.du), p(dc), p(dcs)))
        self.dcx, self.dcu = empty(n * N * B), empty(m * N * B)
        _lib.check(lib.ddp_df_f64_dev(handle.raw, C.byref(prob), p(self.dx), p(self.du), None, p(self.dcx), p(self.dcu), None, None))
        self.dK, self.dk, self.dQuu = empty(m * n * N * B), empty(m * N * B), empty(m * m * N * B)
        self.dVx, self.dVxx, self.ddV = empty(n * N * B), empty(n * n * N * B), empty(2 * B)
        self.ddiv = torch.zeros(B, dtype=torch.int32, device=dev)
        self.dxn, self.dun, self.dcn, self.dcsn = empty(n * N * B), du0, dc, dcs   # u0 / initial-cost buffers are reused
        self.desc = _lib.BPDesc(n, m, N, B, 0, 0, 0, 0, 1, 0)
        self.stats = torch.zeros(4, dtype=torch.float64, device=dev)
        self.comm = None                                      # set by main(): sharding.CApiComm when --collective capi

    def step(self, ev=None, dist=None):
        _lib, L, h, p = self._lib, self.L, self.h, self.p
        if ev is not None:
            _lib.check(L.ddp_event_record(h.raw, ev[0]))
        _lib.check(L.ddp_back_pass_f64_dev(h.raw, C.byref(self.desc), p(self.dcx), p(self.dcu), p(self.dQ), p(self.dcxu), p(self.dR),
                                           p(self.dA), p(self.dB), p(self.dlam), None, None, None, p(self.dK), p(self.dk),
                                           p(self.dQuu), p(self.dVx), p(self.dVxx), p(self.ddV), p(self.ddiv)))
        if ev is not None:
            _lib.check(L.ddp_event_record(h.raw, ev[1]))
        _lib.check(L.ddp_forward_pass_f64_dev(h.raw, C.byref(self.prob), p(self.dK), p(self.dk), p(self.dx0), p(self.du), p(self.dx),
                                              _lib.ptr(self.one), 1, None, None, p(self.dxn), p(self.dun), p(self.dcn), p(self.dcsn)))
        if ev is not None:
            _lib.check(L.ddp_event_record(h.raw, ev[2]))
        if dist is not None:
            # the single collective of the path: batch-level line-search statistics (latency-bound, 32 B)
            _lib.check(L.ddp_batch_stats_f64_dev(h.raw, self.B, p(self.dcsn), p(self.ddV), p(self.ddiv), p(self.stats)))
            if self.comm is not None:
                self.comm.allreduce(self.stats.data_ptr(), 4, 0)     # RCCL through the C ABI (ddp_allreduce_stats_f64_dev)
            elif dist.get_backend() == "gloo":                        # test mode (DDP_BENCH_BACKEND=gloo)
                t = self.stats.cpu()
                dist.all_reduce(t)
                self.stats.copy_(t)
            else:
                # in line on the pass's stream: moving it to a stream of its own (event record / wait per step) measured SLOWER
                # (+30 µs per step instead of +12..23) than the all-reduce itself
                dist.all_reduce(self.stats)

    def line_search_variant(self, fence, steps=20, warmup=5):
        """the "full line-search" variant — one back_pass + the rollouts of ALL 11 step sizes of the reference's
        default α = exp10.(range(0, stop=-3, length=11)) (iLQG.jl:148) per trajectory; informational, outside the timed region."""
        torch, _lib, L, h, p = self.torch, self._lib, self.L, self.h, self.p
        n, m, N, B = self.n, self.m, self.N, self.B
        alpha = 10.0 ** np.linspace(0, -3, 11)
        na = len(alpha)
        xn = torch.empty(n * N * B * na, dtype=torch.float64, device=self.dev)
        un = torch.empty(m * N * B * na, dtype=torch.float64, device=self.dev)
        cn = torch.empty(N * B * na, dtype=torch.float64, device=self.dev)
        cs = torch.empty(B * na, dtype=torch.float64, device=self.dev)

        def step():
            _lib.check(L.ddp_back_pass_f64_dev(h.raw, C.byref(self.desc), p(self.dcx), p(self.dcu), p(self.dQ), p(self.dcxu), p(self.dR),
                                               p(self.dA), p(self.dB), p(self.dlam), None, None, None, p(self.dK), p(self.dk),
                                               p(self.dQuu), p(self.dVx), p(self.dVxx), p(self.ddV), p(self.ddiv)))
            _lib.check(L.ddp_forward_pass_f64_dev(h.raw, C.byref(self.prob), p(self.dK), p(self.dk), p(self.dx0), p(self.du), p(self.dx),
                                                  _lib.ptr(alpha), na, None, None, p(xn), p(un), p(cn), p(cs)))
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        el = time.perf_counter() - t0
        assert np.isfinite(float(cs.sum().item()))
        return {"definition": "1 back_pass + forward_pass for all 11 step sizes alpha = 10^(0..-3) per trajectory (SURVEY.md 8d, iLQG.jl:148)",
                "value": round(B * steps / el, 1), "unit": "iterations/s", "ms_per_step": round(1e3 * el / steps, 4), "n_alpha": na}

    def timed(self, steps, warmup, fence, dist=None):
        _lib, L, h = self._lib, self.L, self.h
        # HIP events bracket the kernels of THREE steps spread over the timed region: an event record costs ~25 µs of stream time here
        # (profiles/sync_latency.py: 0.697 ms per step without events; 15 records in a region added 0.38 ms whatever its length), three
        # of them on every 0.7 ms step would be 10 % of the very number being measured.  Launch times are stable to +-0.5 %.
        every = max(1, steps // 3)
        if os.environ.get("DDP_BENCH_NOEVENTS") == "1":       # diagnosis of the fixed cost of a timed region
            every = 10 ** 9
        events = {}
        for i in range(min(every // 2, steps - 1) if every < 10 ** 9 else steps, steps, every):
            ev = [C.c_void_p() for _ in range(3)]
            for e in ev:
                _lib.check(L.ddp_event_create(h.raw, C.byref(e)))
            events[i] = ev
        for _ in range(warmup):                                  # right in front of the timed region (the events exist already): no idle gap
            self.step(None, dist)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(events.get(i), dist)
        fence()
        elapsed = time.perf_counter() - t0
        bp_ms, fp_ms = [], []
        for ev in events.values():
            ms = C.c_float(0)
            _lib.check(L.ddp_event_elapsed_ms(h.raw, ev[0], ev[1], C.byref(ms))); bp_ms.append(ms.value)
            _lib.check(L.ddp_event_elapsed_ms(h.raw, ev[1], ev[2], C.byref(ms))); fp_ms.append(ms.value)
            for e in ev:
                L.ddp_event_destroy(h.raw, e)
        assert int(self.ddiv.sum().item()) == 0, "synthetic LQ batch must not diverge"
        assert np.isfinite(float(self.dcsn.sum().item()))
        return elapsed, float(np.mean(bp_ms)) if bp_ms else float("nan"), float(np.mean(fp_ms)) if fp_ms else float("nan")

    def roofline(self, bp_avg_ms, fp_avg_ms):
        """algorithmic bytes of SURVEY.md §8(d) / DESIGN.md for the dominant kernel (back_pass)"""
        n, m, N, B = self.n, self.m, self.N, self.B
        bp_read = (n + m) * 8                                  # cx_i, cu_i per step (LTI, time-invariant cost, no limits)
        bp_write = (m * n + m + n + n * n + m * m) * 8         # K_i, k_i, Vx_i, Vxx_i, Quu_i
        fp_bytes = ((m * n + m + n + m) + (n + m + 1)) * 8     # forward: reads K,k,x,u, writes xnew,unew,c
        bp_bytes_launch = (bp_read + bp_write) * (N - 1) * B
        achieved = bp_bytes_launch / (bp_avg_ms * 1e-3) / 1e9
        # what the dispatchers actually launched in the timed region (ddp_last_kernel), not a re-derivation of their rules
        kern = self.h.last_kernel(0) + ("<LTI, shared>" if self.h.last_kernel(0) == "sh_back_kernel" else "<LTI>")
        fwd = self.h.last_kernel(1)
        return {"bound": "hbm", "kernel": kern, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "bytes_per_launch": bp_bytes_launch,
                "avg_launch_ms": round(bp_avg_ms, 4), "avg_launch_ms_samples": "HIP events around three launches spread over the timed region",
                "forward_kernels": {"kernels": fwd + " (cost fused, ddp_problem::cost_diag)" if os.environ.get("DDP_FORWARD_FUSE", "1") != "0"
                                    else "forward_dpp_kernel + cost_kernel", "avg_launch_ms": round(fp_avg_ms, 4),
                                    "bytes_per_launch": fp_bytes * N * B,
                                    "achieved_GBs": round(fp_bytes * N * B / (fp_avg_ms * 1e-3) / 1e9, 1)},
                "pass_bytes": (bp_read + bp_write + fp_bytes) * N}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="trajectories per GPU (BASELINE config 2: 1024)")
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--collective", choices=["torch", "capi"], default="torch",
                    help="who issues the per-step statistics all-reduce: torch.distributed (backend nccl = RCCL) or the C ABI's own RCCL "
                         "communicator (ddp_allreduce_stats_f64_dev; torch.distributed then only ships the 128-byte id)")
    ap.add_argument("--preheat", type=int, default=200, help="untimed passes BEFORE the warmup so that the GPU clocks have settled (a pass is "
                    "0.7 ms; a cold device ramps its clocks over the first ~50 ms); 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic (two rocprofv3 --pmc child runs of a 3-step "
                    "bench after the timed region); the committed figure of profiles/pmc_traffic.json is replayed instead")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C3 / C4 pass lines (profiles/bench_configs.py in a child process)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="trajectories for the CPU baseline (0 = auto ~10-20 s)")
    ap.add_argument("--fill-batch", type=int, default=32768, help="machine-filling batch reported next to the headline (0 = skip)")
    ap.add_argument("--dry-run", action="store_true", help="no device work: the ranks rendezvous (gloo), build their shard of the workload on "
                    "the host and share the statistics vector exactly as the timed run does; rank 0 prints the launch-contract line "
                    "(n_ranks_seen, per-rank seeds and input digests).  What a CPU-only host can check of `--gpus N` (tests/test_sharding_gloo.py)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world == 1 and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU under torch.distributed.run,
        # exactly the command line the driver uses) and pass their output through — rank 0 of the children prints the one JSON line
        return spawn_ranks(args.gpus)
    if args.dry_run:
        return dry_run(args, torch, dist, world, rank)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    backend = os.environ.get("DDP_BENCH_BACKEND", "nccl")   # "gloo": test mode — several ranks may share one GPU (RCCL refuses that), the
    local = local % torch.cuda.device_count() if backend == "gloo" else local     # statistics vector travels through host memory
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "RANK" in os.environ          # under torch.distributed.run the collective path runs even with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import ddp_amd
    from ddp_amd import _lib
    L = _lib.lib()
    # the library launches on torch's current stream, so RCCL collectives and kernels are stream-ordered
    stream = torch.cuda.current_stream(dev).cuda_stream
    h = ddp_amd.Handle(local, stream=stream)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    n, m, N, B = N_STATE, N_CTRL, args.horizon, args.batch
    pb = PassBench(torch, dev, h, L, rank, n, m, N, B)
    if use_dist and args.collective == "capi":
        from ddp_amd import sharding
        pb.comm = sharding.CApiComm(h, rank, world)
    if args.preheat > 0:
        # Untimed passes, then untimed dress rehearsals of the timed region (same K, bounded).  What they are for (profiles/sync_latency3.py):
        # a region that follows an idle gap of even a few milliseconds runs 5-7 % slower (0.735-0.75 ms per step after 50 ms of sleep
        # against 0.69-0.70 back to back: the clocks fall quickly), and so does the first region after a long burst followed by a
        # host-side pause; regions run back to back repeat to +-1 %.
        pb.timed(1, args.preheat, fence, dist if use_dist else None)
        for r in range(int(os.environ.get("DDP_BENCH_REHEARSALS", "3"))):
            e_, _, _ = pb.timed(max(1, min(args.steps, 50)), args.warmup, fence, dist if use_dist else None)
            if os.environ.get("DDP_BENCH_VERBOSE") == "1":
                print("rehearsal %d: %.4f ms per step" % (r, 1e3 * e_ / max(1, min(args.steps, 50))), file=sys.stderr)
    elapsed, bp_ms, fp_ms = pb.timed(args.steps, args.warmup, fence, dist if use_dist else None)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if backend == "gloo" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    roofline = pb.roofline(bp_ms, fp_ms)
    stats_vec = pb.stats.cpu().numpy() if use_dist else None
    # HBM traffic per launch: measured after the timed region by two rocprofv3 --pmc child runs (FETCH_SIZE, WRITE_SIZE: separate passes)
    # of a 3-step bench of the same workload (rank 0, single GPU); where that is not possible (rocprofv3 missing, multi-rank run,
    # --no-traffic) the committed figure for THIS kernel at THIS batch is replayed — never a number of another batch or kernel
    roofline["traffic_source"] = None
    profiled = "rocprofiler" in os.environ.get("LD_PRELOAD", "") or bool(os.environ.get("ROCP_TOOL_LIBRARIES"))     # already under rocprofv3
    if rank == 0 and world == 1 and not args.no_traffic and not profiled and os.environ.get("DDP_BENCH_CHILD") != "1":
        try:
            tr, src = measure_traffic(roofline["kernel"].split("<")[0], B, N)
            roofline["traffic"], roofline["traffic_source"] = tr, src
        except Exception as exc:
            roofline["traffic_source"] = "live PMC measurement failed: %s" % str(exc)[-200:]
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if roofline["traffic"] is None and os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            key = roofline["kernel"].split("<")[0] + "_bytes_per_launch_B%d" % B
            roofline["traffic"] = tj.get(key)
            note = ("profiles/pmc_traffic.json[%s]: %s — replayed from the committed profile, not measured in this run" % (key, tj.get("_source", "?"))
                    if key in tj else "no PMC profile committed for %s" % key)
            roofline["traffic_source"] = note if not roofline["traffic_source"] else roofline["traffic_source"] + "; " + note
        except Exception:
            pass
    if roofline["kernel"].startswith("sh_back_kernel"):
        roofline["note"] = ("config 2 shares fx, fu, cxx, cxu, cuu and lambda across the batch, so the MATRIX half of the recursion (Vxx_i, K_i, Quu_i: "
                            "7 fp64 MFMA per step) is computed once per distinct lambda by one chain wave and every trajectory runs only its affine "
                            "half (Vx_i, k_i, dV: a 16x12 mat-vec per step) plus the write-back of all outputs, which the API contract keeps "
                            "(1 184 B per trajectory-step: the algorithmic bytes are unchanged).  At B=1024 the launch is bound by that one "
                            "999-step serial matrix chain (~1 000 cycles per step, MFMA-latency-bound), not by HBM; from B=2048 on it is "
                            "HBM-write-bound (machine_filling).  The per-trajectory-operand form of the same shape (nothing shared) is the "
                            "C2-LTV line of other_configs.")
    else:
        roofline["note"] = ("each trajectory is a length-N dependency chain: at B=1024 (one wave per SIMD) the fraction is "
                            "latency/occupancy-limited, not bandwidth-limited; see machine_filling for a batch that fills the GPU")

    out = None
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(pb, args.cpu_sample)
        ls = None
        if world == 1 and not args.no_other_configs:
            try:
                ls = pb.line_search_variant(fence)
            except Exception as exc:                            # informational: never in the way of the headline
                ls = {"error": str(exc)[-200:]}
        fill = None
        if world == 1 and args.fill_batch and args.fill_batch != B:
            A_, Bm_, Q_, R_, x0_ = pb.host
            del pb
            torch.cuda.empty_cache()
            pf = PassBench(torch, dev, h, L, rank, n, m, N, args.fill_batch)
            e2, b2, f2 = pf.timed(5, 2, fence, None)
            r2 = pf.roofline(b2, f2)
            fill = {"batch_per_gpu": args.fill_batch, "value": round(args.fill_batch * 5 / e2, 1), "unit": "iterations/s",
                    "ms_per_step": round(1e3 * e2 / 5, 3), "back_pass_kernel": r2["kernel"], "back_pass_ms": r2["avg_launch_ms"],
                    "back_pass_roofline_frac": r2["frac"], "forward_ms": r2["forward_kernels"]["avg_launch_ms"]}
            del pf
        other = None
        if world == 1 and not args.no_other_configs:
            other = other_configs()
        value = B * world * args.steps / elapsed
        out = {"metric": "iLQG iterations/sec (backward+forward, n=10 m=2 T=1000)", "value": round(value, 1),
               "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preheat_steps": args.preheat,
               "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "BASELINE config 2: demo_linear LTI n=10 m=2 N=%d, batch=%d trajectories per GPU, no control "
                                      "limits, regType=1, lambda=1; one step = back_pass + forward_pass(alpha=1) over the batch" % (N, B),
                          "batch_per_gpu": B, "n": n, "m": m, "N": N, "sharding": "batch (independent trajectories), "
                          "one 32-byte RCCL all-reduce of line-search statistics per step when n_gpus>1 (issued by %s)" % ("the C ABI, ddp_allreduce_stats_f64_dev" if args.collective == "capi" else "torch.distributed")},
               "roofline": roofline, "cpu_baseline": cpu, "machine_filling": fill, "full_line_search": ls, "other_configs": other}
        out["n_ranks_seen"] = dist.get_world_size() if use_dist else 1     # from the communicator, not from the command line
        # tiles of the shared-operand backward pass whose cross-work-group wait ran out (results stay correct — the tile is handed to the
        # per-trajectory kernels — but a stall inside the headline kernel is a defect): must be 0; when it is not, who waited for what
        out["sh_timeouts"] = h.sh_timeouts()
        if out["sh_timeouts"]:
            out["sh_timeout_info"] = h.sh_timeout_info()
        if use_dist:
            # the vector the ranks share per step (Σ new cost, Σ dV[1], Σ dV[2], #diverged — summed over the ranks) after the last step
            out["collective"] = {"issued_by": args.collective, "stats": [float(v) for v in stats_vec]}
            if args.collective == "capi":
                from ddp_amd import sharding
                ver, pre = sharding.CApiComm.rccl_info()
                out["collective"].update(rccl_version=ver, rccl_instance="the one already resident in the process" if pre else "loaded by libddp_amd")
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return out


def dry_run(args, torch, dist, world, rank):
    """The launch contract of `bench.py --gpus N` without a device: same environment handling, same per-rank seeds (1000 + rank, the ones
    PassBench uses), the same 4-double statistics vector through an all-reduce — on gloo, with host tensors.  Nothing here is a measurement."""
    import hashlib
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    n, m, N = N_STATE, N_CTRL, min(args.horizon, 50)
    B = min(args.batch, 64)
    seed = 1000 + rank
    A, Bm, Q, R, x0, u0 = make_workload(seed, n, m, N, B)
    digest = int(hashlib.sha256(x0.tobytes() + u0.tobytes()).hexdigest()[:12], 16)
    shared = int(hashlib.sha256(A.tobytes() + Bm.tobytes()).hexdigest()[:12], 16)
    stats = torch.tensor([float(np.sum(x0 * x0)), float(np.sum(u0)), float(B), 0.0], dtype=torch.float64)
    mine = stats.clone()
    rows = [None] * world
    if use_dist:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dist.all_gather_object(rows, {"rank": rank, "seed": seed, "inputs_digest": digest, "shared_operands_digest": shared, "local_stats0": float(mine[0])})
    else:
        rows = [{"rank": rank, "seed": seed, "inputs_digest": digest, "shared_operands_digest": shared, "local_stats0": float(mine[0])}]
    out = None
    if rank == 0:
        out = {"dry_run": True, "metric": "iLQG iterations/sec (backward+forward, n=10 m=2 T=1000)", "value": None, "n_gpus": world,
               "n_ranks_seen": dist.get_world_size() if use_dist else 1, "scaling": "weak", "batch_per_gpu": args.batch,
               "ranks": rows, "collective": {"stats": [float(v) for v in stats]}}
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return out


def spawn_ranks(n):
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # the host driver only supports dmabuf IPC (RCCL across processes)
    r = subprocess.run(cmd, env=env)
    if r.returncode != 0:
        raise SystemExit(r.returncode)
    return None


def measure_traffic(kernel, B, N):
    """roofline.traffic measured for THIS run's kernel: HBM bytes per launch = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024
    (/opt/skills/guides/MI355X_MICROARCH.md: both counters are in 1024-byte units and on gfx950 FETCH_SIZE reports half the bytes of a
    wide read stream), each counter in its own `rocprofv3 --pmc` pass with --kernel-trace only, around a 3-step child run of this file."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        raise RuntimeError("rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix="ddp_pmc_", dir="/tmp")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [rp, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                   "--steps", "3", "--warmup", "1", "--preheat", "0", "--batch", str(B), "--horizon", str(N), "--no-cpu-baseline", "--no-other-configs",
                   "--fill-batch", "0", "--no-traffic"]
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", DDP_BENCH_CHILD="1"), capture_output=True, text=True, timeout=180)       # (some boxes take longer than this to start rocprofv3 + torch: the caller then replays the committed figure and says so)
            acc = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        acc.append(float(row["Counter_Value"]))
            if not acc:
                raise RuntimeError("no %s rows for %s" % (ctr, kernel))
            vals[ctr] = (sum(acc) / len(acc), len(acc))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = int(vals["FETCH_SIZE"][0] * 1024 * 2 + vals["WRITE_SIZE"][0] * 1024)
    return total, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) around a 3-step child "
                   "run of bench.py at the same batch; mean over %d / %d launches of %s; FETCH_SIZE x1024 x2 + WRITE_SIZE x1024 per "
                   "MI355X_MICROARCH.md" % (vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1], kernel))


def other_configs():
    """Pass time of BASELINE configs 3 and 4 on this GPU (profiles/bench_configs.py in a child process, after the timed region):
    informational lines so that every configuration has a number in the driver's record; the graded value is config 2's."""
    import subprocess
    env = dict(os.environ, DDP_C4_SOLVE="0")
    env.setdefault("DDP_BC_WARMUP", "20")                       # clocks settled here too (C3: 60 passes of 0.6 ms, C4: 25 of 10 ms)
    env.setdefault("DDP_BC_STEPS", "40")
    out = []
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "bench_configs.py"), "c3", "c2tv", "c4", "c4tv", "c5", "offA", "offB", "offC", "offD", "offE", "offL"], env=env, capture_output=True,
                           text=True, timeout=400)
        for line in r.stdout.splitlines():
            if line.startswith("{"):
                out.append(json.loads(line))
        if not out:
            out = {"error": (r.stderr or "no output")[-300:]}
    except Exception as exc:
        out = {"error": str(exc)}
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if isinstance(out, list) and os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            for o in out:
                tag = o["config"].split()[0]
                tag = "C2TV" if tag == "C2" else tag.upper()          # off-shape lines: offA -> OFFA_* keys of pmc_traffic.json
                o["back_pass_pmc_bytes_per_launch"] = tj.get("%s_back_pass_bytes_per_launch_B%d" % (tag, o["batch"]))
                busy = tj.get("%s_back_pass_mfma_busy_frac" % tag)
                if busy is not None:
                    o["back_pass_mfma_busy_frac"] = busy
        except Exception:
            pass
    return out


def cpu_baseline(pb, sample):
    """Times the CPU oracle (C restatement of the reference's back_pass + forward_pass, single thread —
    the reference is single-threaded; Julia itself is not installed) on a bounded sample of the same
    workload.  Checker/baseline only — never on the measured GPU path."""
    from oracle import oracle_ctypes as oc
    lib = oc.lib()
    n, m, N, B = pb.n, pb.m, pb.N, pb.B
    A, Bm, Q, R, x0 = pb.host
    prob = oc.make_problem("lq", n, m, N, A=A, B=Bm, Q=Q, R=R)
    x = pb.dx.cpu().numpy().reshape((n, N, B), order="F")
    u = pb.du.cpu().numpy().reshape((m, N, B), order="F")
    cx = pb.dcx.cpu().numpy().reshape((n, N, B), order="F")
    cu = pb.dcu.cpu().numpy().reshape((m, N, B), order="F")

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
    per = t_probe / 32
    S = sample if sample > 0 else int(max(64, 15.0 / per))
    reps = max(1, -(-S // B))                                # more passes than trajectories: repeat the batch
    S1 = min(S, B)
    t = 0.0
    for _ in range(reps):
        dt, nd = run(S1)
        t += dt
    out = {"value": round(S1 * reps / t, 1), "unit": "iterations/s", "cores": 1, "kind": "port",
           "sample": "%d passes (%d of the %d trajectories of the same workload x %d), 1 backward + 1 forward pass each, %.1f s on "
                     "one host core; C restatement of the reference (oracle/ddp_oracle.c), NOT Julia (no Julia toolchain in the "
                     "image)" % (S1 * reps, S1, B, reps, t)}
    # the same restatement on every host core at once (the reference itself is single-threaded; B independent solves
    # parallelise trivially): one Python thread per core, each inside the C call (ctypes drops the GIL)
    try:
        import threading
        T = len(os.sched_getaffinity(0))
        try:                                                          # a container CPU quota is the real core count
            q, per_us = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                T = max(1, min(T, int(float(q) / float(per_us))))
        except Exception:
            pass
        per_thread = int(min(B // max(T, 1), max(8, 4.0 / per))) if T > 1 else 0
        if per_thread >= 1:
            res = [None] * T

            def work(i):
                f = lambda a: np.asfortranarray(a[..., i * per_thread:(i + 1) * per_thread])      # noqa: E731
                xs, us, cxs, cus, x0s = f(x), f(u), f(cx), f(cu), f(x0)
                S = per_thread
                K = np.zeros((m, n, N, S), order="F"); k = np.zeros((m, N, S), order="F"); Quu = np.zeros((m, m, N, S), order="F")
                Vx = np.zeros((n, N, S), order="F"); Vxx = np.zeros((n, n, N, S), order="F"); dV = np.zeros((2, S), order="F")
                xn = np.zeros((n, N, S), order="F"); un = np.zeros((m, N, S), order="F"); cn = np.zeros((N, S), order="F")
                cxu = np.zeros((n, m), order="F")
                P = oc._p
                res[i] = (P, xs, us, cxs, cus, x0s, K, k, Quu, Vx, Vxx, dV, xn, un, cn, cxu)

            for i in range(T):
                work(i)
            Qf, Rf = oc._f(Q), oc._f(R)

            nrep = max(1, int(3.0 / (per * per_thread)))              # ~3 s of work per core

            def call(i):
                P, xs, us, cxs, cus, x0s, K, k, Quu, Vx, Vxx, dV, xn, un, cn, cxu = res[i]
                lib.ddp_oracle_pass_batch_lq_rep(C.byref(prob), per_thread, nrep, P(cxs), P(cus), P(Qf), P(cxu), P(Rf), C.c_double(1.0), 1,
                                                 P(x0s), P(us), P(xs), C.c_double(1.0), P(K), P(k), P(Quu), P(Vx), P(Vxx), P(dV), P(xn), P(un),
                                                 P(cn))

            th = [threading.Thread(target=call, args=(i,)) for i in range(T)]
            t0 = time.perf_counter()
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            dt = time.perf_counter() - t0
            out["all_cores"] = {"value": round(T * per_thread * nrep / dt, 1), "unit": "iterations/s", "cores": T,
                                "sample": "%d passes, %d per core on %d cores, %.1f s" % (T * per_thread * nrep, per_thread * nrep, T, dt)}
    except Exception as exc:                                          # the single-core figure is the contract; this one is extra
        out["all_cores"] = {"error": str(exc)}
    return out


if __name__ == "__main__":
    main()

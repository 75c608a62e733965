#!/usr/bin/env python
"""A/B of the rollout kernels on the headline workload (C2: n=10, m=2, N=1000, B=1024, one α) inside ONE process:
row kernel (forward_pass_dpp.hip, DDP_FORWARD_PIPE=0) against the work-group pipeline (forward_pass_pipe.hip, =1).
Prints the largest differences of xnew / unew / cnew / csum and the HIP-event time per launch of both.

    python profiles/ab_forward.py [B] [N] [reps]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import ddp_amd  # noqa: E402
from ddp_amd import _lib  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    h = ddp_amd.Handle(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    pb = bench.PassBench(torch, dev, h, L, 0, 10, 2, N, B)
    p = pb.p
    pb.step()                                        # K, k of a backward pass
    torch.cuda.synchronize()
    n, m = 10, 2
    out = {}
    for mode in ("0", "2", "1"):
        os.environ["DDP_FORWARD_PIPE"] = mode
        xn = torch.zeros(n * N * B, dtype=torch.float64, device=dev)
        un = torch.zeros(m * N * B, dtype=torch.float64, device=dev)
        cn = torch.zeros(N * B, dtype=torch.float64, device=dev)
        cs = torch.zeros(B, dtype=torch.float64, device=dev)

        def fwd():
            _lib.check(L.ddp_forward_pass_f64_dev(h.raw, C.byref(pb.prob), p(pb.dK), p(pb.dk), p(pb.dx0), p(pb.du), p(pb.dx),
                                                  _lib.ptr(pb.one), 1, None, None, p(xn), p(un), p(cn), p(cs)))
        for _ in range(20):
            fwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fwd()
        e1.record()
        torch.cuda.synchronize()
        out[mode] = (xn.cpu().numpy(), un.cpu().numpy(), cn.cpu().numpy(), cs.cpu().numpy(), e0.elapsed_time(e1) / reps)
    print("two-row pipe kernel  %.4f ms per launch" % out["2"][4])
    a, b = out["0"], out["1"]
    for name, x, y in zip(("xnew", "unew", "cnew", "csum"), a[:4], b[:4]):
        d = np.max(np.abs(x - y)) / max(np.max(np.abs(x)), 1e-300)
        print("%-5s max|row - pipe| / max|row| = %.3e   finite %s" % (name, d, bool(np.all(np.isfinite(y)))))
    if hasattr(L, "ddp_debug_pipe_prof") or os.environ.get("DDP_PIPE_PROF"):
        try:
            buf = (C.c_longlong * 40)()
            L.ddp_debug_pipe_prof(buf)
            NCH = (N + 7) // 8                                   # the last launch was forward_pipe4_kernel: chunks of 8 steps
            names = ["output", "dma", "chain", "prep", "-"]
            for w in range(4):
                v = [buf[8 * w + q] / NCH for q in range(5)]
                print("  %-8s per period of 8 steps (s_memtime ticks): work %7.1f  barrier %7.1f  dma issue %7.1f  K transposition %7.1f" % (names[w], v[0], v[1], v[2], v[3]))
        except Exception as exc:
            print("no phase profile:", exc)
    print("row kernel   %.4f ms per launch" % a[4])
    print("pipe kernel  %.4f ms per launch" % b[4])


if __name__ == "__main__":
    main()

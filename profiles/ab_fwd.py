"""A/B of the C2 rollout kernels (B = 1024, N = 1000): DDP_FORWARD_PIPE = 1 (one row per rollout), 2 (two rows), 0 (row kernel);
HIP-event time per launch through the device entry."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ddp_amd
from ddp_amd import _lib
import ctypes as C
from oracle import np_restatement as npr

rng = np.random.default_rng(0)
n, m, N, B = 10, 2, 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = npr.make_lq_problem(rng, T=N)
prob = ddp_amd.LQProblem(P["A"], P["B"], P["Q"], P["R"])
x0 = np.ones((n, B)) + 0.1 * rng.standard_normal((n, B))
u0 = 0.1 * rng.standard_normal((m, N, B))
x, u, c = ddp_amd.forward_pass(ddp_amd.GaussianPolicy(), x0, u0, None, 1.0, prob, None)
K = 0.05 * rng.standard_normal((m, n, N, B)); k = 0.01 * rng.standard_normal((m, N, B))
pol = ddp_amd.GaussianPolicy(N, n, m, K, k)
ref = None
for mode in ("0", "2", "1", "2", "1"):
    os.environ["DDP_FORWARD_PIPE"] = mode
    _lib.default_handle().raw                                   # picks the switch up (ddp_reload_env)
    out = ddp_amd.forward_pass(pol, x0, u, x, 1.0, prob, None)
    t0 = time.perf_counter()
    for _ in range(5): out = ddp_amd.forward_pass(pol, x0, u, x, 1.0, prob, None)
    if ref is None: ref = out
    err = max(float(np.abs(a - b).max() / np.abs(b).max()) for a, b in zip(out, ref))
    print("mode", mode, _lib.default_handle().last_kernel(1), "max rel diff to the row kernel %.2e" % err)

"""What a reference user sees at B = 1 (the closure-based drop-in of INTEGRATION.md calls the library once per pass with host arrays):
one backward + one forward pass of the BASELINE shape (n = 10, m = 2, N = 1000) through the host-pointer entries of the Python mirror,
and a whole iLQG solve, next to the C restatement of the reference (oracle/, one host core).  VERDICT r03, weak item 10."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ddp_amd
from oracle import np_restatement as npr
from oracle import oracle_ctypes as oc

rng = np.random.default_rng(0)
n, m, N = 10, 2, 1000
P = npr.make_lq_problem(rng, T=N)
prob = ddp_amd.LQProblem(P["A"], P["B"], P["Q"], P["R"])
x0, u0 = P["x0"], P["u0"]
p = oc.make_problem("lq", n, m, N, A=P["A"], B=P["B"], Q=P["Q"], R=P["R"])
x, u, c = ddp_amd.forward_pass(ddp_amd.GaussianPolicy(), x0, u0, None, 1.0, prob, None)
cx, cu = P["Q"] @ x, P["R"] @ u
zero = np.zeros((n, m))


def gpu_pass():
    div, pol, Vx, Vxx, dV = ddp_amd.back_pass(cx, cu, P["Q"], zero, P["R"], P["A"], P["B"], 1.0, 1, None, x, u)
    return ddp_amd.forward_pass(pol, x0, u, x, 1.0, prob, None)


def cpu_pass():
    d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx, cu, P["Q"], zero, P["R"], P["A"], P["B"], 1.0, 1, None, x, u)
    return oc.forward_pass(p, (K, k), x0, u, x, 1.0, None)


def timeit(f, reps):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps


out = {"shape": "n=10 m=2 N=1000 B=1, host arrays in, host arrays out"}
out["gpu_pass_ms"] = round(1e3 * timeit(gpu_pass, 200), 4)
out["oracle_cpu_pass_ms"] = round(1e3 * timeit(cpu_pass, 50), 4)
t0 = time.perf_counter(); r = ddp_amd.iLQG(prob, x0, u0, timing=False); t1 = time.perf_counter()
t0 = time.perf_counter(); r = ddp_amd.iLQG(prob, x0, u0, timing=False); t1 = time.perf_counter()
ro = oc.ilqg(p, x0, u0); t2 = time.perf_counter()
out["gpu_ilqg_solve_ms"] = round(1e3 * (t1 - t0), 3); out["oracle_cpu_ilqg_solve_ms"] = round(1e3 * (t2 - t1), 3)
out["ilqg_iterations"] = [int(r[6]["stats"][1, 0]) if hasattr(r[6]["stats"], "shape") and r[6]["stats"].ndim == 2 else int(r[6]["stats"][1]), int(ro[6]["iter"])]
print(json.dumps(out))

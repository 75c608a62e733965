"""PCIe-inclusive rate of the host-pointer flavour: one back_pass + forward_pass over
B=1024 trajectories of BASELINE config 2 with every operand starting and ending in HOST memory."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddp_amd  # noqa: E402
from bench import make_workload  # noqa: E402

n, m, N, B = 10, 2, 1000, 1024
A, Bm, Q, R, x0, u0 = make_workload(1000, n, m, N, B)
prob = ddp_amd.LQProblem(A, Bm, Q, R)
x, u, c = ddp_amd.forward_pass(ddp_amd.GaussianPolicy(), x0, u0, None, 1.0, prob, None)
cx = np.asfortranarray(np.einsum("ij,jtb->itb", Q, x)); cu = np.asfortranarray(np.einsum("ij,jtb->itb", R, u))     # Julia's layout, as a caller has it
for it in range(6):
    t0 = time.perf_counter()
    div, pol, Vx, Vxx, dV = ddp_amd.back_pass(cx, cu, Q, np.zeros((n, m)), R, A, Bm, 1.0, 1, None, x, u)
    t1 = time.perf_counter()
    xn, un, cn = ddp_amd.forward_pass(pol, x0, u, x, 1.0, prob, None)
    t2 = time.perf_counter()
    print("host-pointer pass %d: back %.1f ms  forward %.1f ms  -> %.0f iterations/s (PCIe + host allocation inclusive)"
          % (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), B / (t2 - t0)))

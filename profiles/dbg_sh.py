import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["DDP_SH_MIN_B"] = "1"
import ddp_amd
from oracle import oracle_ctypes as oc
from test_gpu_shared_lti import _lti
for N in (16, 40):
    rng = np.random.default_rng(10 * N + 1)
    B = 37
    cx, cu, cxx, cxu, cuu, A, Bm, u = _lti(rng, N, B)
    lam = 0.37
    div, pol, Vx, Vxx, dV = ddp_amd.back_pass(cx, cu, cxx, cxu, cuu, A, Bm, lam, 1, None, None, u)
    print("N", N, "kernel", ddp_amd._lib.default_handle().last_kernel(0))
    for b in (0, 1, 2, 3, 4, 5, 36):
        d, (K, k, Quu), vx, vxx, dv = oc.back_pass(cx[..., b], cu[..., b], cxx, cxu, cuu, A, Bm, lam, 1, None, None, u[..., b])
        ek = np.abs(pol.k[..., b] - k).max(0) / (np.abs(k).max() + 1e-300)
        ev = np.abs(Vx[..., b] - vx).max(0) / (np.abs(vx).max() + 1e-300)
        eK = np.abs(pol.K[..., b] - K).max((0, 1)) / np.abs(K).max()
        eV = np.abs(Vxx[..., b] - vxx).max((0, 1)) / np.abs(vxx).max()
        print(" b", b, "div", div[b], d, "\n  k err by step", np.array2string(ek, precision=1, max_line_width=250), "\n  Vx err by step", np.array2string(ev, precision=1, max_line_width=250),
              "\n  K", np.array2string(eK, precision=1, max_line_width=250), "\n  Vxx", np.array2string(eV, precision=1, max_line_width=250))

"""Phase profile of back_pass_mid_kernel (a -DMID_PROF build: profiles/build_variant.sh midprof back_pass_mid.hip "-DMID_PROF -mllvm -amdgpu-mfma-vgpr-form=1",
run with DDP_AMD_LIB=.../build/libddp_midprof.so): s_memtime ticks (100 MHz) per step and phase of trajectory 0 at B = 1024."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddp_amd as ddp

names = ["F -> LDS", "W = Vxx F", "G = F'W + cost", "gains", "K, Y, Vx", "K'Y product", "sym -> image", "loop head", "(G: products", "epilogue", "Vxx store", "= G total incl. load batch + sync)"]
for n, m, lims in [(24, 4, False), (16, 2, False), (24, 4, True), (32, 8, False), (24, 8, False), (20, 6, True)]:
    N, B = 300, 1024
    rng = np.random.default_rng(1)
    fx = np.ascontiguousarray(np.eye(n)[:, :, None, None] * 0.98 + 0.02 * rng.standard_normal((n, n, N, B)))
    fu = 0.1 * rng.standard_normal((n, m, N, B))
    cxx = np.eye(n); cxu = np.zeros((n, m)); cuu = 0.1 * np.eye(m)
    cx = rng.standard_normal((n, N, B)); cu = 0.1 * rng.standard_normal((m, N, B))
    u = 0.1 * rng.standard_normal((m, N, B)); x = None
    L = np.stack([-0.05 * np.ones(m), 0.05 * np.ones(m)], 1) if lims else None
    out = None
    for rep in range(3):
        out = ddp.back_pass(cx, cu, cxx, cxu, cuu, fx, fu, 1.0, 1, L, x, u)
    t = out[3][..., 0].reshape(-1, order="F")[:12] / (N - 1)
    print("n=%d m=%d lims=%d  ticks/step: " % (n, m, lims) + "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, t)) + "   total %.0f" % (t[:8].sum() + t[8:11].sum()))

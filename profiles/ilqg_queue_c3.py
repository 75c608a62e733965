"""32 768 pendulum solves (BASELINE config 3 per trajectory: N = 600, control limits, regType 2) through 4 096 resident slots
(ddp_ilqg_queue_f64_dev) against eight lock-step batches of 4 096 (ddp_ilqg_f64_dev) — device-resident operands, wall time of the
calls:   python profiles/ilqg_queue_c3.py [P] [slots]"""
import ctypes as C
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddp_amd
from ddp_amd import _lib

P = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
L = _lib.lib(); h = _lib.default_handle()
n, m, N = 4, 1, 600
rng = np.random.default_rng(1234)
x0 = np.tile(np.array([np.pi - 0.6, 0.0, 0.0, 0.0])[:, None], (1, P)); x0[0] += rng.uniform(-0.1, 0.1, P)
u0 = np.zeros((m, N, P))
prob = ddp_amd.PendcartProblem()
o = ddp_amd._ilqg_opts(10.0 ** np.linspace(0.2, -3, 6), 1e-8, 1e-8, 1000, 1.0, 1.0, 1.6, 1e15, 1e-6, 2, 0.0)
lims = np.array([[-5.0, 5.0]])
dl = h.to_device(lims)
dx0, du0 = h.to_device(x0), h.to_device(u0)


def outs(B):
    CL = N + 1
    return [h.malloc(8 * s * B) for s in (n * N, m * N, m * n * N, m * N, m * m * N, n * N, n * n * N, CL, 8)]


def stats_of(ptr, B):
    return h.to_host(ptr, (8, B))


# queue
dQ, dR = h.to_device(prob.Q), h.to_device(np.atleast_2d(prob.R))


def dev_problem(B):
    d = ddp_amd._DevProblem(prob, N, B)                      # (host pointers: for the _dev entries Q and R must live on the device)
    d.struct.Q, d.struct.R = dQ.value, dR.value
    return d


dp = dev_problem(P)
o_q = outs(P)
git = C.c_int(0)
for rep in range(2):
    h.sync(); t0 = time.time()
    _lib.check(L.ddp_ilqg_queue_f64_dev(h.raw, C.byref(dp.struct), C.byref(o), S, dx0, du0, dl, *o_q, C.byref(git)))
    h.sync(); tq = time.time() - t0
st_q = stats_of(o_q[8], P)
print("queue: %d problems through %d slots: %.3f s, %d global iterations, iterations per solve median %d max %d, status %s"
      % (P, S, tq, git.value, np.median(st_q[1]), st_q[1].max(), dict(zip(*np.unique(st_q[0].astype(int), return_counts=True)))))
if os.environ.get('DDP_QUEUE_ONLY') == '1':
    sys.exit(0)
# lock-step batches of S
dpb = dev_problem(S)
o_b = outs(S)
tb = 0.0; gits = []
st_b = np.zeros((8, P))
for rep in range(2):
    tb = 0.0; gits = []
    for c in range(0, P, S):
        h.sync(); t0 = time.time()
        _lib.check(L.ddp_ilqg_f64_dev(h.raw, C.byref(dpb.struct), C.byref(o), C.c_void_p(dx0.value + 8 * n * c), C.c_void_p(du0.value + 8 * m * N * c), dl,
                                      *o_b, 0, None, C.byref(git)))
        h.sync(); tb += time.time() - t0; gits.append(git.value)
        st_b[:, c:c + S] = stats_of(o_b[8], S)
print("lock step: %d batches of %d: %.3f s, global iterations %s" % (P // S, S, tb, gits))
print("same summaries: %s;  speed-up %.2fx;  %.0f solves/s" % (np.array_equal(st_q, st_b), tb / tq, P / tq))

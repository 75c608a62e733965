"""Phase profile of the ONE matrix chain of the shared-LTI backward pass (csrc/back_pass_sh.hip, sh_chain) at the headline shape
(n = 10, m = 2, N = 1000, B = 1024, one λ):

    bash profiles/build_variant.sh shprof back_pass_sh.hip "-DSH_PROF"
    DDP_AMD_LIB=$PWD/differentialdynamicprogramming.jl_amd/build/libddp_shprof.so python profiles/sh_phase_profile.py [B]

The chain wave stamps s_memtime (shader clock) at four points of every step and sums the differences; the kernel also records wall-clock
marks (100 MHz): first work-group in, chain wave in / out, last wave out.  The instrumentation costs ~5 scalar memory instructions and a
drain of the LDS queue per step, so the profiled step is longer than the production step: the production time per step is printed next to
it (HIP events around the unprofiled library when DDP_AMD_LIB_PLAIN is given, else the figure from the bench)."""
import ctypes as C
import json
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddp_amd import _lib
from oracle import np_restatement as npr

L = _lib.lib(); h = _lib.default_handle()
n, m, N = 10, 2, 1000
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(1234)
P = npr.make_lq_problem(rng)
os.environ["DDP_SH_MIN_B"] = "1"
cx = 0.01 * rng.standard_normal((n, N, B)); cu = 0.001 * rng.standard_normal((m, N, B))
lam = np.ones(B)
d = [h.to_device(x) for x in (cx, cu, P["Q"], np.zeros((n, m)), P["R"], P["A"], P["B"], lam)]
o = [h.malloc(8 * s * N * B) for s in (m * n, m, m * m, n, n * n)] + [h.malloc(16 * B), h.malloc(4 * B)]
desc = _lib.BPDesc(n, m, N, B, 0, 0, 0, 0, 1, 0)
e0, e1 = C.c_void_p(), C.c_void_p()
L.ddp_event_create(h.raw, C.byref(e0)); L.ddp_event_create(h.raw, C.byref(e1))


def go():
    _lib.check(L.ddp_back_pass_f64_dev(h.raw, C.byref(desc), *d, None, None, None, *o))


for _ in range(20):
    go()
h.sync()
reps = 100
L.ddp_event_record(h.raw, e0)
for _ in range(reps):
    go()
L.ddp_event_record(h.raw, e1)
ms = C.c_float()
L.ddp_event_elapsed_ms(h.raw, e0, e1, C.byref(ms))
t_ms = ms.value / reps
assert h.last_kernel(0) == "sh_back_kernel"
out = {"B": B, "N": N, "ms_per_backward_dispatch": t_ms}
try:
    f = L.ddp_sh_prof
except AttributeError:
    f = None
if f is not None:
    f.restype = C.c_int
    buf = (C.c_ulonglong * 64)()
    _lib.check(f(h.raw, buf))
    p = list(buf)
    if p[8] + p[9]:
        names = ["six products (W = V F, G = F'W + H) to the first use of G", "2x2 gain solve to the issue of the update product",
                 "update product until V is in the operand registers", "step records (LDS) + divergence branches"]
        for so, tag in ((0, "plain steps"), (1, "steps that symmetrise on the chain (every 4th)")):
            cnt = p[8 + so]
            if not cnt:
                continue
            ph = [p[4 * so + e] / cnt for e in range(4)]
            out[tag] = {"steps": cnt, "cycles_per_step": sum(ph), "phases": dict(zip(names, ph))}
        tot = sum(p[0:8]) / (p[8] + p[9])
        out["profiled_cycles_per_step_mean"] = tot
        out["wall_ns"] = {"chain_wave": 10 * (p[17] - p[16]), "first_workgroup_in_to_last_wave_out": 10 * (p[18] - p[19]),
                          "chain_start_after_kernel_start": 10 * (p[16] - p[19]), "tail_after_chain_end": 10 * (p[18] - p[17]),
                          "after_chain_end": {"builder_done": 10 * (p[20] - p[17]), "publisher_done": 10 * (p[21] - p[17]),
                                              "last_tile_has_last_chunk": 10 * (p[23] - p[17]), "last_affine_wave_done": 10 * (p[24] - p[17]),
                                              "last_wave_out": 10 * (p[18] - p[17])}}
        out["chain_ns_per_step_profiled"] = 10 * (p[17] - p[16]) / (N - 1)
        t_end = p[17]
        out["last_four_chunks_us_after_chain_end (group 0, its first tile)"] = {
            name: [round((p[base + c] - t_end) / 100.0, 2) if p[base + c] else None for c in range(4)]
            for name, base in (("chain: chunk computed", 40), ("builder: chunk built", 44), ("publisher: chunk's stores issued", 48),
                               ("tile DMA wave: chunk seen published", 52), ("affine wave: chunk (records + gradients) in the LDS", 56),
                               ("affine wave: chunk's results stored", 60))}
        if p[35]:
            out["affine_wave_ticks_per_chunk (first tile)"] = {"wait for its gradient loads": p[32] / p[35], "wait for the record chunk": p[33] / p[35],
                                                                "8 steps + result stores": p[34] / p[35], "chunks": p[35],
                                                                "chunks already there when asked for": p[36]}
    else:
        out["note"] = "library built without -DSH_PROF: no phase sums"
try:
    ft = L.ddp_sh_prof_tiles
except AttributeError:
    ft = None
if ft is not None and f is not None:
    ft.restype = C.c_int
    tb = (C.c_int * (4 * 1024))()
    W = ft(h.raw, tb, 1024)
    if W > 0:
        t = np.array(list(tb)[: 4 * W]).reshape(W, 4)
        ce = int(p[17] & 0x7fffffff)
        rel = (t[:, :3] - ce) / 100.0                            # us after the chain wave's end
        out["tiles"] = {"count": W, "xcc_of_tiles": {int(x): int((t[:, 3] == x).sum()) for x in np.unique(t[:, 3])}}
        for k, name in enumerate(("DMA wave done", "first affine wave done", "first writer wave done")):
            out["tiles"][name + " (us after chain end): min / median / max"] = [round(float(v), 2) for v in (rel[:, k].min(), np.median(rel[:, k]), rel[:, k].max())]
            out["tiles"][name + " by xcc (median)"] = {int(x): round(float(np.median(rel[t[:, 3] == x, k])), 2) for x in np.unique(t[:, 3])}
        worst = np.argsort(-rel[:, 1])[:8]
        out["tiles"]["slowest tiles (index, xcc, DMA, affine, writer)"] = [[int(i), int(t[i, 3])] + [round(float(v), 2) for v in rel[i]] for i in worst]
print(json.dumps(out, indent=1))

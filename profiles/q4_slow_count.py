"""How often the straight-line two-iteration box-QP of the pendulum backward pass (csrc/back_pass_q4.hip, boxqp1_two_iterations) hands a
step to the generic loop: a -DQ4_COUNT_SLOW build of the library (profiles/build_variant.sh q4count back_pass_q4.hip "-DQ4_COUNT_SLOW",
DDP_AMD_LIB pointing at it) counts per trajectory-step; this runs the C3 pass of profiles/bench_configs.py and a whole C3 solve."""
import ctypes as C
import os
import runpy
import sys

sys.argv = [sys.argv[0], "c3"]
os.environ.setdefault("DDP_BC_WARMUP", "0"); os.environ.setdefault("DDP_BC_STEPS", "1")
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_configs.py"), run_name="__main__")
L = ns["L"]
out = (C.c_ulonglong * 7)()
f = L.ddp_q4_slow_count
f.argtypes = [C.POINTER(C.c_ulonglong)]; f.restype = C.c_int
assert f(out) == 0
print("C3 pass(es): steps through the generic loop %d of %d trajectory-steps = %.4f" % (out[0], out[1], out[0] / max(1, out[1])))
print("  why: H <= 0: %d, |grad| < minGrad at the warm start: %d, no descent: %d, Armijo back-off: %d, a third iteration: %d" % tuple(out[2:7]))

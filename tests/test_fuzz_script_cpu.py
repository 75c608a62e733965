"""The randomised sweep (tests/fuzz_gpu_parity.py) runs on a GPU box by hand; its case generator and the CPU-only conditioning
mode are exercised here so the script cannot rot: the two CPU restatements must agree on the generated cases."""
import importlib.util
import os

import numpy as np

from conftest import relerr

HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    spec = importlib.util.spec_from_file_location("fuzz_gpu_parity", os.path.join(HERE, "fuzz_gpu_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_generated_cases_agree_between_the_cpu_restatements():
    from oracle import np_restatement as npr
    from oracle import oracle_ctypes as oc
    fz = _load()
    seen = set()
    for case in range(12):
        c = fz.gen_case(np.random.default_rng([7, case]))
        if c["n"] >= 40:
            continue                                          # keep the CPU suite short
        seen.add((c["n"], c["m"]))
        for b in range(min(c["B"], 2)):
            d1, (K1, k1, Q1), vx1, vxx1, dv1 = fz.ref_back_pass(oc.back_pass, c, b)
            d2, (K2, k2, Q2), vx2, vxx2, dv2 = fz.ref_back_pass(npr.back_pass, c, b)
            assert d1 == d2
            if d1 == 0:
                assert relerr(K1, K2) < 1e-9 and relerr(vxx1, vxx2) < 1e-9 and relerr(k1, k2) < 1e-9
    assert len(seen) >= 2

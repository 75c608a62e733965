"""CPU-side checks of the C ABI (no compute calls): the library loads, exports every symbol that
include/ddp_amd.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

import ddp_amd
from ddp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ddp_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ddp_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 28
    for s in syms:
        assert hasattr(L, s), "missing export %s" % s
    assert set(syms) == set(_lib.EXPORTS), set(syms) ^ set(_lib.EXPORTS)


def test_no_cpu_fallback_without_gpu():
    L = _lib.lib()
    if L.ddp_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(ddp_amd.DDPError, match="no CPU fallback"):
        ddp_amd.Handle(0)
    import numpy as np
    with pytest.raises(ddp_amd.DDPError):
        ddp_amd.boxQP(np.eye(2), np.ones(2), -np.ones(2), np.ones(2), np.zeros(2))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "differentialdynamicprogramming.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".jl")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle_ctypes" not in src and "np_restatement" not in src and "ddp_oracle" not in src, f


def test_kernels_that_count_their_memory_operations_do_not_spill():
    """The pipeline kernels wait with `s_waitcnt vmcnt(N)` where N is the number of loads THEY issued (forward_pass_pipe.hip,
    back_pass_mx*.hip, back_pass_sh.hip, back_pass_q4.hip).  A register spill would add scratch loads / stores the count does not know
    about: the build records every kernel's resource usage (build.py, -Rpass-analysis=kernel-resource-usage) and none of these kernels
    may use scratch.  (the bit-identity tests on the GPU were the only guard against a compiler change.)"""
    import glob
    import json
    import __graft_entry__ as g
    g.build()
    files = glob.glob(os.path.join(ROOT, "differentialdynamicprogramming.jl_amd", "build", "*.usage.json"))
    assert files, "no resource-usage records: build.py did not run with -Rpass-analysis=kernel-resource-usage"
    usage = {}
    for f in files:
        usage.update(json.load(open(f)))
    counted = [k for k in usage if any(s in k for s in ("forward_pipe", "sh_back_kernel", "back_pass_mx2_kernel", "back_pass_mx_kernel",
                                                        "back_pass_q4c_kernel", "back_pass_q4l_kernel"))]
    assert len(counted) >= 8, counted
    for k in counted:
        u = usage[k]
        assert int(u["ScratchSize"]) == 0 and int(u["VGPRs Spill"]) == 0 and u["Dynamic Stack"] == "False", (k, u)

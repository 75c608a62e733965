#!/bin/bash
# back pass of the headline workload (C2, B = 1024) under variant builds of the library:  bash profiles/ab_mx_exp.sh lib1.so lib2.so ...
# ("" = the product build); DDP_MX2=0 selects the one-wave kernel
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export DDP_AMD_LIB=$PWD/$lib; else unset DDP_AMD_LIB; fi
  python - <<'PY'
import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, bench, ddp_amd
from ddp_amd import _lib
dev = torch.device("cuda", 0); L = _lib.lib()
h = ddp_amd.Handle(0, stream=torch.cuda.current_stream(dev).cuda_stream)
pb = bench.PassBench(torch, dev, h, L, 0, 10, 2, 1000, int(os.environ.get("AB_B", "1024")))
p = pb.p
def bp():
    _lib.check(L.ddp_back_pass_f64_dev(h.raw, C.byref(pb.desc), p(pb.dcx), p(pb.dcu), p(pb.dQ), p(pb.dcxu), p(pb.dR), p(pb.dA), p(pb.dB), p(pb.dlam), None, None, None, p(pb.dK), p(pb.dk), p(pb.dQuu), p(pb.dVx), p(pb.dVxx), p(pb.ddV), p(pb.ddiv)))
for _ in range(300): bp()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(300): bp()
e1.record(); torch.cuda.synchronize()
print(os.environ.get("DDP_AMD_LIB", "product")[-14:], "back pass %.4f ms" % (e0.elapsed_time(e1) / 300))
PY
done

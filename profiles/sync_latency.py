import sys, time, os
sys.path.insert(0, "/root/repo")
import torch
import bench, ctypes as C
import ddp_amd
from ddp_amd import _lib
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
L = _lib.lib()
h = ddp_amd.Handle(0, stream=torch.cuda.current_stream(dev).cuda_stream)
pb = bench.PassBench(torch, dev, h, L, 0, 10, 2, 1000, 1024)
for _ in range(200): pb.step()
torch.cuda.synchronize()
def run(K, spin):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K): pb.step()
    if spin:
        ev = torch.cuda.Event(); ev.record()
        while not ev.query(): pass
    torch.cuda.synchronize()
    return time.perf_counter() - t0
for K in (0, 1, 5, 20, 50, 200):
    for spin in (False, True):
        ts = [run(K, spin) for _ in range(7)]
        print("K=%3d spin=%d  median %.4f ms total, %.4f ms per step" % (K, spin, 1e3*sorted(ts)[3], 1e3*sorted(ts)[3]/max(K,1)))

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
fence = torch.cuda.synchronize
pb.timed(1, 200, fence, None)
for K in (20, 20, 20, 50, 50, 20, 20):
    e, b, f = pb.timed(K, 5, fence, None)
    print("timed(K=%d): %.4f ms per step (%.3f ms total)" % (K, 1e3 * e / K, 1e3 * e))

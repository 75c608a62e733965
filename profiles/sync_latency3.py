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
os.environ["DDP_BENCH_NOEVENTS"] = "1"
def seq(label, pre, K, reps, sleep=0.0):
    if pre: pb.timed(1, pre, fence, None)
    out = []
    for r in range(reps):
        if sleep: time.sleep(sleep)
        e, _, _ = pb.timed(K, 0, fence, None)
        out.append(1e3 * e / K)
    print(label, " ".join("%.4f" % v for v in out))
seq("after a 200-step burst, 12 regions of K=20 back to back:", 200, 20, 12)
seq("after a 1000-step burst, 12 regions of K=20:           ", 1000, 20, 12)
seq("no burst, 12 regions of K=20 with 50 ms sleeps between:  ", 0, 20, 12, 0.05)
seq("after a 200-step burst, 6 regions of K=100:              ", 200, 100, 6)
seq("after 20 steps, 6 regions of K=20:                        ", 20, 20, 6)

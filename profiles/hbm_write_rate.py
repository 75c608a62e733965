"""What a write-dominated kernel can reach: fill (write-only), copy (read + write) and reduction (read-only) rates of the HBM on this GPU,
next to the 8 TB/s figure the roofline fractions are quoted against.  The backward pass writes 92 % of its traffic (K, k, Quu, Vx, Vxx)."""
import time
import torch
dev = torch.device("cuda", 0)
n = 1 << 31                                                   # 16 GB of float64
x = torch.empty(n, dtype=torch.float64, device=dev)
y = torch.empty(n // 2, dtype=torch.float64, device=dev)
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
t = timed(lambda: x.fill_(1.0)); print("fill   (write only) : %.2f TB/s" % (8 * n / t / 1e12))
t = timed(lambda: y.copy_(x[: n // 2])); print("copy   (read+write) : %.2f TB/s total traffic" % (2 * 8 * (n // 2) / t / 1e12))
t = timed(lambda: x.sum()); print("sum    (read only)  : %.2f TB/s" % (8 * n / t / 1e12))

# the write pattern of a machine-filling backward pass: B far-apart streams (one per trajectory, 800 KB apart), each receiving a burst of
# `steps` x 800 bytes per visit, all streams visited before the next burst — against the same bytes written as one contiguous run
B, N = 32768, 64
v = torch.empty((B, N, 100), dtype=torch.float64, device=dev)              # [trajectory, step, 100 doubles]: Vxx[:, :, i, b] of 64 steps
for steps in (1, 4, 8, 16, 64):
    def strided():
        for t in range(0, N, steps):
            v[:, t:t + steps, :].fill_(1.0)
    t = timed(strided, reps=3)
    print("%5d streams x bursts of %5d bytes: %.2f TB/s" % (B, steps * 800, 8 * v.numel() / t / 1e12))

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

import torch, time, numpy as np
d = torch.empty(1 << 27, dtype=torch.float64, device="cuda")   # 1 GiB
d.fill_(1.0)
hp = torch.empty(1 << 27, dtype=torch.float64)                  # pageable
hn = torch.empty(1 << 27, dtype=torch.float64).pin_memory()
for name, h in (("pageable", hp), ("pinned", hn)):
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); h.copy_(d); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("D2H %s: %.1f GB/s" % (name, 1.0737 / dt))
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); d.copy_(h); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("H2D %s: %.1f GB/s" % (name, 1.0737 / dt))
a = np.empty(1 << 27); b = np.ones(1 << 27)
t = time.perf_counter(); a[:] = b; dt = time.perf_counter() - t
print("host memcpy 1 GiB single thread: %.1f GB/s" % (1.0737 / dt))

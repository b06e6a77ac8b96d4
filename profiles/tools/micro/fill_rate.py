"""Write-only HBM rate of this box: torch fill of the optimal-completion output size (1.95 GB), and a copy."""
import torch
dev = torch.device("cuda:0")
n = 512 * 4096 * 116
x = torch.empty(n, dtype=torch.long, device=dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = t(lambda: x.fill_(-1))
print("fill  %.3f ms  %.2f TB/s" % (ms, n * 8 / ms / 1e9))
y = torch.empty_like(x)
ms = t(lambda: y.copy_(x))
print("copy  %.3f ms  %.2f TB/s (read + write)" % (ms, 2 * n * 8 / ms / 1e9))

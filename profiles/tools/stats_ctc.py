import os, sys, ctypes, torch

sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F, _cabi
dev = torch.device("cuda:0")
T, N, V, K = int(os.environ.get("T", 512)), int(os.environ.get("N", 4096)), int(os.environ.get("V", 256)), 16
g = torch.Generator(device=dev).manual_seed(3)
logits = torch.randn((T, N, V + 1), device=dev, generator=g)
peak = torch.randint(0, V + 1, (T, N, 1), device=dev, generator=g)
logits.scatter_add_(2, peak, torch.full((T, N, 1), float(os.environ.get("SCALE", 12.0)), device=dev))
L = _cabi.lib()
buf = (ctypes.c_ulonglong * 16)()
L.pdt_debug_read_stats(buf, 1)
F.ctc_prefix_search(logits, K); torch.cuda.synchronize()
L.pdt_debug_read_stats(buf, 1)
names = ["producer frames", "short list", "miss: too few", "miss: too many", "consumer completes", "lean tier fails", "third entry wins: 1 prefix", "third entry wins: several", "simd0 prod", "simd0 cons", "simd1 prod", "simd1 cons", "simd2 prod", "simd2 cons", "simd3 prod", "simd3 cons"]
for i, nm in enumerate(names):
    print("%-20s %10d  %6.2f%%" % (nm, buf[i], 100.0 * buf[i] / (N * T)))

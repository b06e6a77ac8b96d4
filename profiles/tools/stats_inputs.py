import os, sys, ctypes, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
import bench
from pydrobert_amd import functional as F, _cabi
dev = torch.device("cuda:0")
T, N, V, K = 512, 4096, 256, 16
L = _cabi.lib()
names = ["producer frames", "short list", "miss: too few", "miss: too many", "consumer completes", "lean tier fails", "third entry wins: 1 prefix", "third entry wins: several"]
def stats(lg, tag):
    buf = (ctypes.c_ulonglong * 16)()
    L.pdt_debug_read_stats(buf, 1)
    y, yl, yp = F.ctc_prefix_search(lg, K); torch.cuda.synchronize()
    L.pdt_debug_read_stats(buf, 1)
    print(tag, {nm: round(100.0 * buf[i] / (N * T), 3) for i, nm in enumerate(names)}, "mean len %.1f" % yl.float().mean().item(), "zero-mass utt %.3f" % (yp[:, 0] == 0).float().mean().item())
stats(bench.speechlike_logits(T, N, V, dev, 5, bench.synthetic_bigram_dicts(V)), "speech-like")
g = torch.Generator(device=dev).manual_seed(9)
lg = torch.randn((T, N, V + 1), device=dev, generator=g)
peak = torch.randint(0, V + 1, (T, N, 1), device=dev, generator=g)
lg.scatter_add_(2, peak, torch.full((T, N, 1), 8.0, device=dev)); stats(lg, "peak +8")

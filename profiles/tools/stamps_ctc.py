import os, sys, ctypes, torch
os.environ.setdefault("PDT_AMD_LIB", os.path.abspath("pydrobert-pytorch_amd/csrc/build/variants/stamps/lib.so"))
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F, _cabi
dev = torch.device("cuda:0")
T, N, V, K = 512, int(os.environ.get("N", 4096)), int(os.environ.get("V", 256)), 16
g = torch.Generator(device=dev).manual_seed(3)
logits = torch.randn((T, N, V + 1), device=dev, generator=g)
peak = torch.randint(0, V + 1, (T, N, 1), device=dev, generator=g)
logits.scatter_add_(2, peak, torch.full((T, N, 1), 12.0, device=dev))
if os.environ.get("SPEECH"):  # blank-dominated rows (bench.speechlike_logits)
    import bench
    logits = bench.speechlike_logits(T, N, V, dev, 5, bench.synthetic_bigram_dicts(V))
L = _cabi.lib()
buf = (ctypes.c_ulonglong * 16)()
F.ctc_prefix_search(logits, K); torch.cuda.synchronize()
L.pdt_debug_read_stamps(buf, 1)
F.ctc_prefix_search(logits, K); torch.cuda.synchronize()
L.pdt_debug_read_stamps(buf, 1)
tot = sum(buf[i] for i in range(14))
names = ["wait ready", "top-M list", "masses+merge", "rounds(rest)", "state+trie", "isp/nxt", "output walk", "lean prep", "lean sort", "lean shfl", "chm/info", "walk: checkpoints", "walk: segments", "-"]
for i, nm in enumerate(names):
    print("%-14s %8.1f cycles/frame  %5.1f%%" % (nm, buf[i] / (N * T), 100.0 * buf[i] / tot))
print("total %.1f cycles/frame/wave" % (tot / (N * T)))

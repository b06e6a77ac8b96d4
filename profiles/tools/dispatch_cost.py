import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from pydrobert_amd import functional as F, _decoding as D
dev = torch.device("cuda:0")
N, K, V, S = 1024, 16, 1000, 100
g = torch.Generator(device=dev).manual_seed(4)
lpt = torch.randn((N, K, V), device=dev, generator=g).log_softmax(-1)
lpp = torch.randn((N, K), device=dev, generator=g)
yb = torch.randint(0, V, (S, N, K), device=dev, generator=g)
raw = D._beam_search_advance_op._init_fn
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
with torch.no_grad():
    print("op    %.1f us" % t(lambda: F.beam_search_advance(lpt, K, lpp, yb)))
    print("raw   %.1f us" % t(lambda: raw(lpt, K, lpp, yb, None)))
    print("empty %.1f us" % t(lambda: torch.empty((S + 1, N, K), device=dev, dtype=torch.long)))

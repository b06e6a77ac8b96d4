"""Both step functions at the C3 shape (N=1024, K=16, V=1000, S=100): bench.py's C3_ctc_prefix_search_advance and
C3_beam_search_advance, event-timed per call (host glue included) -- and under rocprofv3 the kernels alone."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from bench import event_ms, peaky_logits
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
N, K, V, S = 1024, 16, 1000, 100
g = torch.Generator(device=dev).manual_seed(4)
lpt = torch.randn((N, K, V), device=dev, generator=g).log_softmax(-1)
lpp = torch.randn((N, K), device=dev, generator=g)
yb = torch.randint(0, V, (S, N, K), device=dev, generator=g)
ybl = torch.full((N, K), S, device=dev)
print("beam_search_advance ms", ["%.4f" % event_ms(lambda: F.beam_search_advance(lpt, K, lpp, yb, ybl)) for _ in range(3)])
print("beam_search_advance (no lens) ms", ["%.4f" % event_ms(lambda: F.beam_search_advance(lpt, K, lpp, yb)) for _ in range(3)])
lg = peaky_logits(S + 1, N, V, dev, 0x5EED0003)
nb, b = torch.zeros((N, 1), device=dev), torch.ones((N, 1), device=dev)
yh = torch.zeros((0, N, 1), dtype=torch.long, device=dev)
last = lens = torch.zeros((N, 1), dtype=torch.long, device=dev)
isp = torch.ones((N, 1, 1), dtype=torch.bool, device=dev)
for t in range(S + 1):
    p = lg[t].softmax(1)
    nonext, blank = p[:, :V].contiguous(), p[:, V].contiguous()
    args = ((nonext.unsqueeze(1).expand(N, nb.shape[1], V), nonext, blank), K, (nb, b), yh, last, lens, isp)
    if t < S:
        yh, last, lens, (nb, b), isp, _, _ = F.ctc_prefix_search_advance(*args)
print("ctc_prefix_search_advance ms", ["%.4f" % event_ms(lambda: F.ctc_prefix_search_advance(*args)) for _ in range(3)])
ext = args[0][0].contiguous()
args2 = ((ext, args[0][1], args[0][2]),) + args[1:]
print("ctc_prefix_search_advance (dense ext rows) ms", ["%.4f" % event_ms(lambda: F.ctc_prefix_search_advance(*args2)) for _ in range(3)])
# host cost per call: many calls back to back, wall clock (the GPU queue absorbs them)
import time
for name, fn in (("beam_search_advance (no lens)", lambda: F.beam_search_advance(lpt, K, lpp, yb)),
                 ("ctc_prefix_search_advance", lambda: F.ctc_prefix_search_advance(*args))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s: host %.1f us per call to enqueue, %.1f us per call until the queue drained" % (name, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))

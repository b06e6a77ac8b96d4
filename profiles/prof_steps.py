"""A few launches of ONE step function at the C3 shape (N=1024, K=16, V=1000, S=100) for rocprofv3 passes:
python3 profiles/prof_steps.py beam|ctc"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F
from bench import peaky_logits
dev = torch.device("cuda:0")
N, K, V, S = 1024, 16, 1000, 100
g = torch.Generator(device=dev).manual_seed(4)
if sys.argv[1] == "beam":
    lpt = torch.randn((N, K, V), device=dev, generator=g).log_softmax(-1)
    lpp = torch.randn((N, K), device=dev, generator=g)
    yb = torch.randint(0, V, (S, N, K), device=dev, generator=g)
    for _ in range(5):
        out = F.beam_search_advance(lpt, K, lpp, yb)
else:
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
    for _ in range(5):
        out = F.ctc_prefix_search_advance(*args)
torch.cuda.synchronize()
print(out[1][0])

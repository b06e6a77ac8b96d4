"""The two step functions at the C3 shape (N=1024, K=16, V=1000, S=100), ten launches each: run under
rocprofv3 --kernel-trace --stats for the kernels' own durations (variant builds through PDT_AMD_LIB)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from bench import peaky_logits
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
N, K, V, S = 1024, 16, 1000, 100
g = torch.Generator(device=dev).manual_seed(4)
lpt = torch.randn((N, K, V), device=dev, generator=g).log_softmax(-1)
lpp = torch.randn((N, K), device=dev, generator=g)
yb = torch.randint(0, V, (S, N, K), device=dev, generator=g)
for _ in range(10):
    F.beam_search_advance(lpt, K, lpp, yb)
lg = peaky_logits(S + 1, N, V, dev, 0x5EED0003)
p = lg[S].softmax(1)
nonext, blank = p[:, :V].contiguous(), p[:, V].contiguous()
nb, b = torch.rand((N, K), device=dev, generator=g), torch.rand((N, K), device=dev, generator=g)
last = yb[-1].clone()
lens = torch.full((N, K), S, device=dev)
isp = torch.eye(K, dtype=torch.bool, device=dev).expand(N, K, K).contiguous()
args = ((nonext.unsqueeze(1).expand(N, K, V), nonext, blank), K, (nb, b), yb, last, lens, isp)
for _ in range(10):
    F.ctc_prefix_search_advance(*args)
torch.cuda.synchronize()

"""beam_search_advance, one step at the C3 shape (N=1024, K=16, V=1000, S=100) -- bench.py's C3_beam_search_advance."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from bench import event_ms
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
N, K, V, S = 1024, 16, 1000, 100
g = torch.Generator(device=dev).manual_seed(4)
lpt = torch.randn((N, K, V), device=dev, generator=g).log_softmax(-1)
lpp = torch.randn((N, K), device=dev, generator=g)
yb = torch.randint(0, V, (S, N, K), device=dev, generator=g)
ybl = torch.full((N, K), S, device=dev)
print("beam_search_advance ms", ["%.4f" % event_ms(lambda: F.beam_search_advance(lpt, K, lpp, yb, ybl)) for _ in range(3)])

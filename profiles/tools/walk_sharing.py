import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from pydrobert_amd import functional as F
from bench import peaky_logits
dev = torch.device("cuda:0")
lg = peaky_logits(512, 4096, 256, dev, 0x5EED0002)
y, yl, yp = F.ctc_prefix_search(lg, 16)
S, N, W = y.shape
pos = torch.arange(S, device=dev).view(S, 1, 1)
valid = pos < yl.unsqueeze(0)
same = ((y == y[:, :, :1]) & valid).all(2) & valid.all(2)   # all 16 equal and present
cp = same.long().cumprod(0).sum(0)          # common prefix length per utterance
print("mean len", yl.float().mean().item(), "min-len mean", yl.min(1).values.float().mean().item(), "common prefix mean", cp.float().mean().item())
print("tokens total", int(yl.sum()), "shared tokens x16", int(cp.sum()) * 16, "frac", float(cp.sum() * 16) / float(yl.sum()))

"""A few frames of CTCPrefixSearch with the GRU-cell language model of bench.py (hidden 256 ->
Linear(256, 1000)) at C3's shape, for `rocprofv3 --kernel-trace --stats`: which kernel runs the
dense logit GEMM (the path's only MFMA-eligible work) and what share of a frame it is."""
import sys

import torch

sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
import bench
from pydrobert_amd import modules as M

dev = torch.device("cuda:0")
T, N, V, K = 64, 1024, 1000, 16
lg = bench.peaky_logits(T, N, V, dev, 0x5EED0003)
torch.manual_seed(5)
search = M.CTCPrefixSearch(K, 0.2, bench.make_gru_lm(M, V).to(dev))
with torch.no_grad():
    search(lg[:4])
    torch.cuda.synchronize()
    y, yl, yp = search(lg)
torch.cuda.synchronize()
print(yl[0, :4].tolist(), yp[0, :4].tolist())

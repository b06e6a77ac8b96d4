"""A few launches of the one-kernel CTC search at a given shape (for rocprofv3 passes):
python3 profiles/prof_ctc_shape.py T N V [K] [launches]"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F
from bench import peaky_logits
T, N, V = (int(x) for x in sys.argv[1:4])
K = int(sys.argv[4]) if len(sys.argv) > 4 else 16
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
dev = torch.device("cuda:0")
lg = peaky_logits(T, N, V, dev, 0x5EED0006)
for _ in range(reps):
    y, yl, yp = F.ctc_prefix_search(lg, K)
torch.cuda.synchronize()
print(yl[0], yp[0])

import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
T, N, V, K = 512, 4096, 256, 16
g = torch.Generator(device=dev).manual_seed(3)
logits = torch.randn((T, N, V + 1), device=dev, generator=g)
peak = torch.randint(0, V + 1, (T, N, 1), device=dev, generator=g)
logits.scatter_add_(2, peak, torch.full((T, N, 1), 12.0, device=dev))
for _ in range(3):
    y, yl, yp = F.ctc_prefix_search(logits, K)
torch.cuda.synchronize()
print(yl[0], yp[0])

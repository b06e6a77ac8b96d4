"""C4 sparse_image_warp timing: (2048, 1, 1000, 80), 3 control + 4 pinned points, order 2."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
N, T, Fq = 2048, 1000, 80
g = torch.Generator(device=dev).manual_seed(7)
img = torch.randn((N, 1, T, Fq), device=dev, generator=g)
src = torch.rand((N, 3, 2), device=dev, generator=g) * torch.tensor([T - 1.0, Fq - 1.0], device=dev)
dst = src + torch.randn((N, 3, 2), device=dev, generator=g)
fn = lambda: F.sparse_image_warp(img, src, dst, pinned_boundary_points=1, include_flow=False)
for _ in range(2): fn()
ts = []
for _ in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort(); print("sparse_image_warp ms min %.3f median %.3f" % (ts[0], ts[len(ts)//2]))

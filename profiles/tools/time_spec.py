"""C4 spec_augment_apply timing (N=2048, T=1000, F=80), parameters drawn once."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import modules as M
dev = torch.device("cuda:0")
N, T, Fq = 2048, 1000, 80
g = torch.Generator(device=dev).manual_seed(7)
feats = torch.randn((N, T, Fq), device=dev, generator=g)
lens = torch.randint(T // 2, T + 1, (N,), device=dev, generator=g)
sa = M.SpecAugment().to(dev)
params = sa.draw_parameters(feats, lens)
fn = lambda: sa.apply_parameters(feats, params, lens)
for _ in range(3): fn()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
ts.sort(); print("spec_augment_apply ms min %.3f median %.3f" % (ts[0], ts[len(ts)//2]))

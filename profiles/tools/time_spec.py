"""C4 SpecAugment (N=2048, T=1000, F=80): apply_parameters and forward, event-timed (A/B of img_warp.hip builds, PDT_AMD_LIB)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from pydrobert_amd import modules as M
from bench import event_ms
dev = torch.device("cuda:0")
N, T, Fq = 2048, 1000, 80
g = torch.Generator(device=dev).manual_seed(7)
feats = torch.randn((N, T, Fq), device=dev, generator=g)
lens = torch.randint(500, T + 1, (N,), device=dev, generator=g)
sa = M.SpecAugment(max_time_warp=80.0, max_freq_warp=0.0, max_time_mask=100, max_freq_mask=27,
                   max_time_mask_proportion=0.04, num_time_mask=2, num_time_mask_proportion=1.0,
                   num_freq_mask=2, interpolation_order=1)
params = sa.draw_parameters(feats, lens)
tag = os.environ.get("PDT_AMD_LIB", "default")[-30:]
print(tag, "apply ms", ["%.4f" % event_ms(lambda: sa.apply_parameters(feats, params, lens), reps=9, warm=3) for _ in range(3)])
print(tag, "forward ms", ["%.4f" % event_ms(lambda: sa(feats, lens), reps=9, warm=3) for _ in range(3)])

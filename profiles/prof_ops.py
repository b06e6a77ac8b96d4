"""Workload for the rocprofv3 passes of profiles/collect.sh: every hot kernel a few times at its
BASELINE shape (one process, so each PMC pass sees the same launches).

    python3 profiles/prof_ops.py [names...]     default: all
      string   error_rate / prefix_error_rates / optimal_completion at C2 (lev_classify + lev_bitpar,
               lev_rowsync, oc_expand) and edit_distance with costs 1/2/3 (lev_skewed, the cell-by-cell kernel)
      ctc      fused search at the bench shape (N=4096, T=512, V=256, K=16)
      ctc_flat the same with +6 instead of +12 on the peak class (the unkind input)
      ctc_long fused search at C3 (N=1024, T=1000, V=1000): the three-producer instantiation
      spec     spec_augment_apply + sparse_image_warp at C4
"""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F  # noqa: E402
from pydrobert_amd import modules as M  # noqa: E402

dev = torch.device("cuda:0")
want = set(sys.argv[1:]) or {"string", "ctc", "ctc_flat", "ctc_long", "spec"}
REPS = 3


def peaky(T, N, V, seed, scale=12.0, chunk=64):
    """bench.py's generator (chunk = T there for the bench shape: the same tensor as the bench's)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    lg = torch.empty((T, N, V + 1), device=dev)
    for t0 in range(0, T, chunk):
        part = lg[t0:t0 + chunk]
        part.normal_(generator=g)
        peak = torch.randint(0, V + 1, (part.shape[0], N, 1), device=dev, generator=g)
        part.scatter_add_(2, peak, torch.full((part.shape[0], N, 1), scale, device=dev))
    return lg


if "string" in want:
    T, N, V = 512, 4096, 256
    g = torch.Generator(device=dev).manual_seed(2)
    ref = torch.randint(0, V, (T, N), device=dev, generator=g)
    hyp = torch.randint(0, V, (T, N), device=dev, generator=g)
    for _ in range(REPS):
        F.error_rate(ref, hyp, warn=False)
        F.prefix_error_rates(ref, hyp, warn=False)
        F.edit_distance(ref, hyp, ins_cost=1.0, del_cost=2.0, sub_cost=3.0, warn=False)
        oc = F.optimal_completion(ref, hyp, warn=False)
    print("optimal_completion C =", oc.shape[-1])
    del ref, hyp, oc
if "ctc" in want:
    lg = peaky(512, 4096, 256, 0x5EED0003, chunk=512)  # bench.py's logits of rank 0
    for _ in range(REPS):
        y, yl, yp = F.ctc_prefix_search(lg, 16)
    print("ctc", float(yp[0, 0]))
    del lg
if "ctc_flat" in want:
    lg = peaky(512, 4096, 256, 0x5EED0003, scale=6.0, chunk=512)
    for _ in range(REPS):
        y, yl, yp = F.ctc_prefix_search(lg, 16)
    print("ctc_flat", float(yp[0, 0]))
    del lg
if "ctc_long" in want:
    lg = peaky(1000, 1024, 1000, 0x5EED0003)  # bench.py other_configs C3
    for _ in range(REPS):
        y, yl, yp = F.ctc_prefix_search(lg, 16)
    print("ctc_long", float(yp[0, 0]))
    del lg
if "spec" in want:
    N, T, Fq = 2048, 1000, 80
    feats = torch.randn((N, T, Fq), device=dev)
    lens = torch.randint(500, T + 1, (N,), device=dev)
    sa = M.SpecAugment(max_time_warp=80.0, max_freq_warp=0.0, max_time_mask=100, max_freq_mask=27,
                       max_time_mask_proportion=0.04, num_time_mask=2, num_time_mask_proportion=1.0,
                       num_freq_mask=2, interpolation_order=1)
    params = sa.draw_parameters(feats, lens)
    img = feats.view(N, 1, T, Fq)
    src = torch.rand((N, 3, 2), device=dev) * torch.tensor([T - 1.0, Fq - 1.0], device=dev)
    dst = src + torch.randn((N, 3, 2), device=dev)
    for _ in range(REPS):
        out = sa.apply_parameters(feats, params, lens)
        w = F.sparse_image_warp(img, src, dst, pinned_boundary_points=1, include_flow=False)
    print("spec", float(out[0, 0, 0]), float(w[0, 0, 0, 0]))
torch.cuda.synchronize()

"""Per-config timings (BASELINE configs 2-4) with HIP events; prints a small table."""
import sys, time, json, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
from pydrobert_amd import functional as F, modules as M
dev = torch.device("cuda:0")

def timeit(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))

res = {}
# C3: ctc prefix search N=1024, T=1000, V=1000, K=16 (peaky logits)
T, N, V, K = 1000, 1024, 1000, 16
g = torch.Generator(device=dev).manual_seed(3)
lg = torch.randn((T, N, V + 1), device=dev, generator=g)
lg.scatter_add_(2, torch.randint(0, V + 1, (T, N, 1), device=dev, generator=g), torch.full((T, N, 1), 12.0, device=dev))
ms = timeit(lambda: F.ctc_prefix_search(lg, K), reps=3, warm=1)
res["C3 ctc_prefix_search N=1024 T=1000 V=1000 K=16"] = dict(ms=ms, utt_per_s=N / ms * 1e3, GBs=lg.numel() * 4 / ms / 1e6)
del lg
# C3, the bare step functions at S=100 (the state comes from 100 real frames of the search itself)
N, V, K, S = 1024, 1000, 16, 100
g = torch.Generator(device=dev).manual_seed(4)
nb, b = torch.zeros((N, 1), device=dev), torch.ones((N, 1), device=dev)
y = torch.zeros((0, N, 1), dtype=torch.long, device=dev)
last = lens = torch.zeros((N, 1), dtype=torch.long, device=dev)
isp = torch.ones((N, 1, 1), dtype=torch.bool, device=dev)
for t in range(S + 1):
    lgt = torch.randn((N, V + 1), device=dev, generator=g)
    lgt.scatter_add_(1, torch.randint(0, V + 1, (N, 1), device=dev, generator=g), torch.full((N, 1), 12.0, device=dev))
    p = lgt.softmax(1)
    nonext, blank = p[:, :V].contiguous(), p[:, V].contiguous()
    ext = nonext.unsqueeze(1).expand(N, nb.shape[1], V)
    args = ((ext, nonext, blank), K, (nb, b), y, last, lens, isp)
    if t == S:
        break
    y, last, lens, (nb, b), isp, _, _ = F.ctc_prefix_search_advance(*args)
ms = timeit(lambda: F.ctc_prefix_search_advance(*args))
res["C3 bare ctc_prefix_search_advance N=1024 K=16 V=1000 S=100"] = dict(ms=ms, steps_per_s=1e3 / ms)
lpt = torch.randn((N, K, V), device=dev, generator=g).log_softmax(-1)
lpp = torch.randn((N, K), device=dev, generator=g)
yb = torch.randint(0, V, (S, N, K), device=dev, generator=g)
ybl = torch.full((N, K), S, device=dev)
ms = timeit(lambda: F.beam_search_advance(lpt, K, lpp, yb, ybl))
res["C3-like bare beam_search_advance N=1024 K=16 V=1000 S=100"] = dict(ms=ms, steps_per_s=1e3 / ms)
del lpt, yb
# C4: SpecAugment N=2048 x 1000 x 80
N, T, Fq = 2048, 1000, 80
feats = torch.randn((N, T, Fq), device=dev)
lens = torch.randint(500, T + 1, (N,), device=dev)
sa = M.SpecAugment(max_time_warp=80.0, max_freq_warp=0.0, max_time_mask=100, max_freq_mask=27,
                   max_time_mask_proportion=0.04, num_time_mask=2, num_time_mask_proportion=1.0,
                   num_freq_mask=2, interpolation_order=1)
params = sa.draw_parameters(feats, lens)
ms = timeit(lambda: sa.apply_parameters(feats, params, lens))
res["C4 spec_augment_apply N=2048 T=1000 F=80"] = dict(ms=ms, utt_per_s=N / ms * 1e3, GBs=2 * feats.numel() * 4 / ms / 1e6)
ms = timeit(lambda: sa(feats, lens))
res["C4 SpecAugment.forward (draw+apply)"] = dict(ms=ms, utt_per_s=N / ms * 1e3, GBs=2 * feats.numel() * 4 / ms / 1e6)
img = feats.view(N, 1, T, Fq)
src = torch.rand((N, 3, 2), device=dev) * torch.tensor([T - 1.0, Fq - 1.0], device=dev)
dst = src + torch.randn((N, 3, 2), device=dev)
ms = timeit(lambda: F.sparse_image_warp(img, src, dst, pinned_boundary_points=1, include_flow=False), reps=3, warm=1)
res["C4 sparse_image_warp M=3+4 order 2 noflow"] = dict(ms=ms, img_per_s=N / ms * 1e3, GBs=2 * feats.numel() * 4 / ms / 1e6)
del feats, img
# C2 ragged variant
T, N, V = 512, 4096, 256
rng = np.random.default_rng(2)
ref = rng.integers(0, V, (T, N)); hyp = rng.integers(0, V, (T, N))
rl = rng.integers(T // 2, T + 1, N); hl = rng.integers(T // 2, T + 1, N)
for n in range(N):
    if rl[n] < T: ref[rl[n], n] = V
    if hl[n] < T: hyp[hl[n], n] = V
ref, hyp = torch.from_numpy(ref).to(dev), torch.from_numpy(hyp).to(dev)
for name in ("error_rate", "prefix_error_rates", "optimal_completion"):
    ms = timeit(lambda: getattr(F, name)(ref, hyp, eos=V, warn=False))
    res["C2 ragged " + name] = dict(ms=ms, utt_per_s=N / ms * 1e3)
ms = timeit(lambda: F.error_rate(ref, hyp, eos=V, ins_cost=3.0, del_cost=3.0, sub_cost=4.0, warn=False))
res["C2 ragged error_rate NIST costs (count mode)"] = dict(ms=ms, utt_per_s=N / ms * 1e3)
ms = timeit(lambda: F.edit_distance(ref[:, :512], hyp[:, :512], eos=V, ins_cost=0.1, del_cost=0.7, sub_cost=1.3, warn=False), reps=2, warm=1)
res["C2 ragged edit_distance inexact costs (exact-unroll path), N=512"] = dict(ms=ms, utt_per_s=512 / ms * 1e3)
del ref, hyp
# spec augment backward + pad_variable
N, T, Fq = 2048, 1000, 80
feats = torch.randn((N, T, Fq), device=dev, requires_grad=True)
lens = torch.randint(500, T + 1, (N,), device=dev)
params = sa.draw_parameters(feats, lens)
y = sa.apply_parameters(feats, params, lens)
g = torch.randn_like(y)
ms = timeit(lambda: torch.autograd.grad(y, feats, g, retain_graph=True))
res["C4 spec_augment_apply backward"] = dict(ms=ms, GBs=2 * feats.numel() * 4 / ms / 1e6)
pad = torch.stack([torch.randint(0, 50, (N,), device=dev), torch.randint(0, 50, (N,), device=dev)])
x = feats.detach()
ms = timeit(lambda: F.pad_variable(x, lens, pad, "reflect"))
res["pad_variable reflect N=2048 T=1000 F=80"] = dict(ms=ms, GBs=2 * x.numel() * 4 * 0.8 / ms / 1e6)
del feats, y, g, x
# n-gram LM lookup: 16384 rows (N*K of C3) x V=1000, trigram table
rng = np.random.default_rng(5)
V = 1000
uni = {v: (float(rng.normal()), float(rng.normal())) for v in range(V)}
bi = {tuple(int(t) for t in rng.integers(0, V, 2)): (float(rng.normal()), float(rng.normal())) for _ in range(200000)}
tri = {(int(rng.integers(0, V)),) + k: float(rng.normal()) for k in list(bi)[:100000]}
lm = M.LookupLanguageModel(V, V, [uni, bi, tri]).to(dev)
hist = torch.randint(0, V, (100, 16384), device=dev)
idx = torch.full((16384,), 100, device=dev)
ms = timeit(lambda: lm.calc_idx_log_probs(hist, dict(), idx))
res["LookupLanguageModel trigram rows=16384 V=1000"] = dict(ms=ms, rows_per_s=16384 / ms * 1e3, GBs=16384 * V * 4 / ms / 1e6)
del lm, hist
# HOCD loss forward + backward at C2 shapes (V=256 classes)
T, N, V = 512, 1024, 256
ref = torch.randint(0, V, (T, N), device=dev); hyp = torch.randint(0, V, (T, N), device=dev)
logits = torch.randn((T, N, V), device=dev, requires_grad=True)
def hocd():
    loss = F.hard_optimal_completion_distillation_loss(logits, ref, hyp, warn=False)
    torch.autograd.grad(loss, logits)
ms = timeit(hocd, reps=3, warm=1)
res["hard OCD loss fwd+bwd N=1024 T=512 V=256"] = dict(ms=ms, utt_per_s=N / ms * 1e3)
del logits, ref, hyp
# C5 shard decode: N=4096, T=512, V=5000 (42 GB of logits, generated on device)
T, N, V, K = 512, 4096, 5000, 16
g = torch.Generator(device=dev).manual_seed(5)
lg = torch.empty((T, N, V + 1), device=dev)
for t0 in range(0, T, 64):
    lg[t0:t0 + 64].normal_(generator=g)
    lg[t0:t0 + 64].scatter_add_(2, torch.randint(0, V + 1, (64, N, 1), device=dev, generator=g), torch.full((64, N, 1), 12.0, device=dev))
ms = timeit(lambda: F.ctc_prefix_search(lg, K), reps=2, warm=1)
res["C5 shard ctc_prefix_search N=4096 T=512 V=5000 K=16"] = dict(ms=ms, utt_per_s=N / ms * 1e3, GBs=lg.numel() * 4 / ms / 1e6)
del lg
for k, v in res.items():
    print("%-70s %s" % (k, json.dumps({a: round(b, 3) for a, b in v.items()})))
json.dump(res, open("gpurun_out/configs.json", "w"), indent=1)

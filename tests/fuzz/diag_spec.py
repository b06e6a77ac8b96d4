import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd"); sys.path.insert(0, "tests")
import oracle
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
_t = lambda a, d: torch.from_numpy(np.ascontiguousarray(a)).to(d)
for order in (1, 2, 3):
    rng = np.random.default_rng(60 + order)
    N, T, Fq = 24, 90, 12
    feats = rng.normal(size=(N, T, Fq)).astype(np.float32)
    lens = rng.integers(1, T + 1, N); lens[:6] = [1, 2, 3, T, T - 1, 4]
    w_0 = (rng.random(N) * lens).astype(np.float32)
    w = ((rng.random(N) * 2 - 1) * np.minimum(lens / 2, 8)).astype(np.float32)
    e = torch.empty(0)
    x, ln = _t(feats, dev), _t(lens, dev)
    params = (_t(w_0, dev), _t(w, dev), e, e, e, e, e, e)
    act = F.spec_augment_apply_parameters(x, params, order, ln).cpu().numpy()
    grid = F.warp_1d_grid(params[0], params[1], ln, T, order)
    two = torch.ops.pydrobert_amd.spec_augment_apply(x, grid, None, None, None, None, None).cpu().numpy()
    exp = oracle.spec_augment_apply_parameters(feats, (w_0, w, None, None, None, None, None, None), order, lens)
    valid = np.arange(T)[None, :, None] < lens[:, None, None]
    ea = np.abs(np.where(valid, act - exp, 0)).max((1, 2)); eb = np.abs(np.where(valid, two - exp, 0)).max((1, 2)); ab = np.abs(np.where(valid, act - two, 0)).max((1, 2))
    print("order", order)
    for n in range(N):
        if max(ea[n], eb[n], ab[n]) > 2e-5: print("  n", n, "len", lens[n], "w0", w_0[n], "w", w[n], "fused-oracle %.2e grid-oracle %.2e fused-grid %.2e" % (ea[n], eb[n], ab[n]))

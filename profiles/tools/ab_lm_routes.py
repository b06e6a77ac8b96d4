"""C3 + bigram model: the factor-table search against the one-call step route (same prefixes on all but a few
near ties; the table route's fused softmax differs from torch's in the last ulp) and its time."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
import bench
from pydrobert_amd import modules as M, switches
dev = torch.device("cuda:0")
T, N, V, K = int(os.environ.get("T", 1000)), int(os.environ.get("N", 1024)), 1000, 16
dicts = bench.synthetic_bigram_dicts(V)
lm = M.LookupLanguageModel(V, V, [d.copy() for d in dicts]).to(dev)
lg = bench.speechlike_logits(T, N, V, dev, 0x5EED0009, dicts)
lens = torch.randint(T // 2, T + 1, (N,), device=dev)
for vm in (False, True):
    search = M.CTCPrefixSearch(K, 0.2, lm, valid_mixture=vm)
    outs = []
    for table in (1, 0):
        switches.set("PDT_CTC_LM_TABLE", table)
        with torch.no_grad():
            outs.append(search(lg, lens))
    switches.set("PDT_CTC_LM_TABLE", 1)
    (ty, tyl, typ), (y, yl, yp) = outs
    tm = torch.arange(ty.shape[0], device=dev).view(-1, 1, 1) < tyl.unsqueeze(0)
    m = torch.arange(y.shape[0], device=dev).view(-1, 1, 1) < yl.unsqueeze(0)
    same = (tyl == yl).all(1) & (torch.where(tm, ty, 0) == torch.where(m, y, 0)).all(0).all(1)
    d = (typ[same].double().log() - yp[same].double().log()).abs().max()
    print("valid_mixture" if vm else "fusion", "utterances with other beams: %d of %d; max |dlogp| %.2e" % (int((~same).sum()), N, float(d)), flush=True)
    with torch.no_grad():
        print("   table route %.2f ms" % bench.event_ms(lambda: search(lg), reps=3, warm=1), flush=True)

"""C3 + bigram model (speech-like logits), factor-table search: fusion and valid mixture (A/B through PDT_AMD_LIB)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
import bench
from pydrobert_amd import modules as M
dev = torch.device("cuda:0")
T, N, V, K = 1000, 1024, 1000, 16
dicts = bench.synthetic_bigram_dicts(V)
lm = M.LookupLanguageModel(V, V, [d.copy() for d in dicts]).to(dev)
lg = bench.speechlike_logits(T, N, V, dev, 0x5EED0009, dicts)
out = []
for vm in (False, True):
    search = M.CTCPrefixSearch(K, 0.2, lm, valid_mixture=vm)
    with torch.no_grad():
        out.append("%s %s" % ("mixture" if vm else "fusion", ["%.2f" % bench.event_ms(lambda: search(lg), reps=3, warm=1) for _ in range(2)]))
print(" | ".join(out))

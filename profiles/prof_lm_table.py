"""A few launches of the factor-table LM search at C3 (for rocprofv3 passes)."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
import bench
from pydrobert_amd import modules as M
dev = torch.device("cuda:0")
T, N, V, K = 1000, 1024, 1000, 16
vm = len(sys.argv) > 1 and sys.argv[1] == "valid"
dicts = bench.synthetic_bigram_dicts(V)
lm = M.LookupLanguageModel(V, V, [d.copy() for d in dicts]).to(dev)
lg = bench.speechlike_logits(T, N, V, dev, 0x5EED0009, dicts)
search = M.CTCPrefixSearch(K, 0.2, lm, valid_mixture=vm)
with torch.no_grad():
    for _ in range(3):
        y, yl, yp = search(lg)
torch.cuda.synchronize()
print(yl[0], yp[0])

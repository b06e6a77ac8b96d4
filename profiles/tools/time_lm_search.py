"""C3 with the n-gram model in the loop: CTCPrefixSearch(16, 0.2, LookupLanguageModel), N=1024 T=1000 V=1000;
PDT_CTC_LM_FUSED=0 for the three-kernel route."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
import bench
from pydrobert_amd import modules as M
dev = torch.device("cuda:0")
T, N, V, K = 1000, 1024, 1000, 16
lg = bench.peaky_logits(T, N, V, dev, 0x5EED0003)
lm = bench.synthetic_bigram_lm(M, V, dev)
search = M.CTCPrefixSearch(K, 0.2, lm)
with torch.no_grad():
    search(lg[:8])
    print("C3_search_lookup_lm ms %.2f" % bench.event_ms(lambda: search(lg), reps=2, warm=0))

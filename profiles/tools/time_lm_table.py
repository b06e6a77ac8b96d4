"""C3 + bigram model: the factor-table search against the other routes (speech-like and SURVEY logits)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
import bench
from pydrobert_amd import modules as M, switches
dev = torch.device("cuda:0")
T, N, V, K = 1000, 1024, 1000, 16
dicts = bench.synthetic_bigram_dicts(V)
lm = M.LookupLanguageModel(V, V, [d.copy() for d in dicts]).to(dev)
for name, lg in (("speechlike", bench.speechlike_logits(T, N, V, dev, 0x5EED0009, dicts)), ("survey", bench.peaky_logits(T, N, V, dev, 0x5EED0003))):
    for vm in (False, True):
        search = M.CTCPrefixSearch(K, 0.2, lm, valid_mixture=vm)
        for table in (1, 0):
            switches.set("PDT_CTC_LM_TABLE", table)
            with torch.no_grad():
                ms = bench.event_ms(lambda: search(lg), reps=3, warm=1)
            print(name, "valid_mixture" if vm else "fusion", "table" if table else "one-call", "%.2f ms" % ms, flush=True)
    del lg

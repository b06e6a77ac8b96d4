"""Consumer-wave cycles by segment in the factor-table LM search (a -DPDT_LMTAB_STAMPS build:
bash profiles/tools/build_var.sh lmtabstamps ctc_lm_table.hip -DPDT_LMTAB_STAMPS; run with PDT_AMD_LIB=<that lib>)."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
import bench
from pydrobert_amd import modules as M, _cabi
dev = torch.device("cuda:0")
T, N, V, K = 1000, 1024, 1000, 16
dicts = bench.synthetic_bigram_dicts(V)
lm = M.LookupLanguageModel(V, V, [d.copy() for d in dicts]).to(dev)
lg = bench.speechlike_logits(T, N, V, dev, 0x5EED0009, dicts)
names = ["own lists", "wait for workers", "frame", "publish (+ row wait)"]
for vm in (False, True):
    search = M.CTCPrefixSearch(K, 0.2, lm, valid_mixture=vm)
    with torch.no_grad():
        search(lg); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        _cabi.lib().pdt_debug_read_lmtab_stamps(buf, 1)
        ms = bench.event_ms(lambda: search(lg), reps=1, warm=0)
        _cabi.lib().pdt_debug_read_lmtab_stamps(buf, 1)
    tot = sum(buf[:4])
    print("valid_mixture" if vm else "fusion", "%.2f ms" % ms, "cycles per frame and utterance:",
          {n: round(buf[i] / (T * N)) for i, n in enumerate(names)}, "sum", round(tot / (T * N)))
    lnames = ["wait contexts / row", "etab + mixed row", "threshold + survivors", "sort + list + positions"]
    print("   list builder, all four waves, cycles per frame and utterance:",
          {n: round(buf[4 + i] / (T * N)) for i, n in enumerate(lnames)})

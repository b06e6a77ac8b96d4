"""Per-utterance end time / lean-tier exits / contexts of the bigram-LM search (C3 + speech-like logits);
-DPDT_UTT_STATS build of ctc_lm_table.hip: PDT_AMD_LIB=.../variants/uttlm/lib.so"""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
import bench
from pydrobert_amd import modules as M
dev = torch.device("cuda:0")
T, N, V, K = 1000, 1024, 1000, 16
L = ctypes.CDLL(os.environ["PDT_AMD_LIB"])
L.pdt_debug_read_utt_stats_lm.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
dicts = bench.synthetic_bigram_dicts(V)
lm = M.LookupLanguageModel(V, V, [d.copy() for d in dicts]).to(dev)
lg = bench.speechlike_logits(T, N, V, dev, 0x5EED0009, dicts)
for vm in (False, True):
    search = M.CTCPrefixSearch(K, 0.2, lm, valid_mixture=vm)
    buf = np.zeros((N, 4), dtype=np.uint32)
    with torch.no_grad():
        for _ in range(2):
            search(lg); torch.cuda.synchronize()
            L.pdt_debug_read_utt_stats_lm(buf.ctypes.data, N, 1)
    end = buf[:, 0].astype(np.float64); exits = buf[:, 1]; lists = buf[:, 2].astype(np.float64) / T; ctxs = buf[:, 3].astype(np.float64) / T
    print("valid mixture" if vm else "fusion", "end: p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f | lean exits mean %.1f max %d | contexts per frame mean %.2f | LISTS per frame mean %.2f max %.2f | corr(end, lists) %.2f corr(end, exits) %.2f"
          % (*np.percentile(end, [10, 50, 90, 99]), end.max(), exits.mean(), exits.max(), ctxs.mean(), lists.mean(), lists.max(), np.corrcoef(end, lists)[0, 1], np.corrcoef(end, exits)[0, 1]))

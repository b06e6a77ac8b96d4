"""Where the consumer waves of the factor-table LM search run: (XCC, SE, SH, CU, SIMD) of every utterance's
consumer (a -DPDT_UTT_STATS -DPDT_UTT_HWID build of ctc_lm_table.hip under PDT_AMD_LIB)."""
import os, sys, ctypes, collections, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
import bench
from pydrobert_amd import modules as M
dev = torch.device("cuda:0")
T, N, V, K = 100, 1024, 1000, 16
L = ctypes.CDLL(os.environ["PDT_AMD_LIB"])
L.pdt_debug_read_utt_stats_lm.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
dicts = bench.synthetic_bigram_dicts(V)
lm = M.LookupLanguageModel(V, V, [d.copy() for d in dicts]).to(dev)
lg = bench.speechlike_logits(T, N, V, dev, 0x5EED0009, dicts)
search = M.CTCPrefixSearch(K, 0.2, lm)
buf = np.zeros((N, 4), dtype=np.uint32)
with torch.no_grad():
    search(lg); torch.cuda.synchronize()
    L.pdt_debug_read_utt_stats_lm(buf.ctypes.data, N, 1)
hw = buf[:, 1]
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = (hw >> 16) & 15
per = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
print("CUs hosting consumers:", len(per), "consumers per CU: min %d max %d" % (min(per.values()), max(per.values())))
ps = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
print("consumers per (CU, SIMD): histogram", sorted(collections.Counter(ps.values()).items()))
print("SIMD ids of the consumers:", sorted(collections.Counter(simd.tolist()).items()))

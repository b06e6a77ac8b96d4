"""Where the headline launch's tail comes from: per-utterance end time of the consumer loop (-DPDT_UTT_STATS build) by
workgroup position.  PDT_AMD_LIB=.../variants/utt/lib.so python profiles/tools/utt_balance.py"""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
import bench
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
T, N, V, K = 512, 4096, 256, 16
L = ctypes.CDLL(os.environ["PDT_AMD_LIB"])
L.pdt_debug_read_utt_stats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
lg = bench.peaky_logits(T, N, V, dev, 0x5EED0003, chunk=T)
buf = np.zeros((N, 4), dtype=np.uint32)
for _ in range(3):
    F.ctc_prefix_search(lg, K); torch.cuda.synchronize()
    L.pdt_debug_read_utt_stats(buf.ctypes.data, N, 1)
end = buf[:, 0].astype(np.float64); wait = buf[:, 3].astype(np.float64)
print("end: min %.0f p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f (x16 cycles)" % (end.min(), *np.percentile(end, [10, 50, 90, 99]), end.max()))
print("wait share of the loop: p10 %.2f p50 %.2f p90 %.2f" % tuple(np.percentile(wait / end, [10, 50, 90])))
wg = np.arange(N) // 2
# the kernel maps workgroup b -> utterances through xcd_remap: utterance pair index = remapped b; XCD of the hosting workgroup = b % 8
nwg = N // 2; q = nwg // 8
xcd = wg // q  # (nwg divisible by 8: contiguous ranges per XCD)
for x in range(8):
    m = xcd == x
    print("  XCD %d: end p50 %.0f p99 %.0f max %.0f" % (x, np.median(end[m]), np.percentile(end[m], 99), end[m].max()))
# position inside the XCD's range (dispatch order)
pos = wg % q
for lo in range(0, q, q // 8):
    m = (pos >= lo) & (pos < lo + q // 8)
    print("  dispatch position %4d..%4d: end p50 %.0f max %.0f, wait share p50 %.2f" % (lo, lo + q // 8 - 1, np.median(end[m]), end[m].max(), np.median(wait[m] / end[m])))
print("  pair partner difference |end0 - end1| p50 %.0f p99 %.0f" % tuple(np.percentile(np.abs(end[0::2] - end[1::2]), [50, 99])))

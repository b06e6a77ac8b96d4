"""Per-utterance consumer cycles / lean-tier exits / list completions / producer waits of the headline CTC search
(-DPDT_UTT_STATS build of ctc_search.hip: PDT_AMD_LIB=.../variants/utt/lib.so; wave-local counters, one store per utterance) on a slow and a fast draw of the same distribution."""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
import bench
from pydrobert_amd import functional as F, _cabi
dev = torch.device("cuda:0")
T, N, V, K = 512, 4096, 256, 16
L = ctypes.CDLL(os.environ["PDT_AMD_LIB"])
L.pdt_debug_read_utt_stats.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
def run(lg, tag):
    buf = np.zeros((N, 4), dtype=np.uint32)
    F.ctc_prefix_search(lg, K); torch.cuda.synchronize()
    L.pdt_debug_read_utt_stats(buf.ctypes.data, N, 1)
    F.ctc_prefix_search(lg, K); torch.cuda.synchronize()
    L.pdt_debug_read_utt_stats(buf.ctypes.data, N, 1)
    cyc = buf[:, 0].astype(np.float64) * 16; fails = buf[:, 1]; comp = buf[:, 2]; wait = buf[:, 3].astype(np.float64) * 16
    print(tag, "consumer cycles: mean %.0f p50 %.0f p99 %.0f max %.0f | lean exits mean %.1f max %d | completions mean %.2f max %d | wait mean %.0f max %.0f"
          % (cyc.mean(), np.median(cyc), np.percentile(cyc, 99), cyc.max(), fails.mean(), fails.max(), comp.mean(), comp.max(), wait.mean(), wait.max()))
    worst = np.argsort(-cyc)[:8]
    print("   slowest utterances:", [(int(i), int(cyc[i]), int(fails[i]), int(comp[i]), int(wait[i])) for i in worst])
    # by workgroup (2 utterances) and CU-ish groups
    e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e[0].record(); F.ctc_prefix_search(lg, K); e[1].record(); torch.cuda.synchronize()
    print("   launch %.3f ms; loop-end time / 100 MHz counter: p50 %.3f ms max %.3f ms" % (e[0].elapsed_time(e[1]), np.median(cyc) / 1e5, cyc.max() / 1e5))
    # utterances by end time: who ends last, and what they did
    hist = np.histogram(fails, bins=[0, 5, 10, 20, 40, 80, 160, 513])[0]
    print("   lean exits histogram [0,5,10,20,40,80,160,512]:", hist.tolist())
    for lo, hi in ((0, 10), (10, 40), (40, 160), (160, 513)):
        m = (fails >= lo) & (fails < hi)
        if m.any(): print("     exits in [%d,%d): %d utterances, loop-end p50 %.3f ms max %.3f ms" % (lo, hi, m.sum(), np.median(cyc[m]) / 1e5, cyc[m].max() / 1e5))
    print("   corr(cycles, lean exits) %.2f  corr(cycles, completions) %.2f  corr(cycles, wait) %.2f" % (np.corrcoef(cyc, fails)[0, 1], np.corrcoef(cyc, comp)[0, 1], np.corrcoef(cyc, wait)[0, 1]))
run(bench.peaky_logits(T, N, V, dev, 3, chunk=64), "slow draw")
run(bench.peaky_logits(T, N, V, dev, 3, chunk=512), "fast draw")

"""CTCPrefixSearch + the GRU-cell LM of bench.py at the C3 shape, T frames (argv[1], default 300): wall time per
frame, the host's share (enqueue time against drain time) -- and under rocprofv3 --kernel-trace --stats the
GPU's own busy time per frame."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
import bench
from pydrobert_amd import modules as M
dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N, V, K = 1024, 1000, 16
lg = bench.speechlike_logits(T, N, V, dev, 0x5EED0008, bench.synthetic_bigram_dicts(V))
torch.manual_seed(5)
gru = bench.make_gru_lm(M, V).to(dev)
search = M.CTCPrefixSearch(K, 0.2, gru)
from pydrobert_amd import switches
with torch.no_grad():
  search(lg[:8])
  for mix in ((1, 0, 1, 0) if len(sys.argv) > 2 else (1,)):
    switches.set("PDT_CTC_STEP_MIX", mix)
    print("PDT_CTC_STEP_MIX =", mix)
    torch.cuda.synchronize()
    for _ in range(2):
        t0 = time.perf_counter()
        search(lg)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("T=%d: host returned after %.1f ms, queue drained after %.1f ms: %.1f us per frame" % (T, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) / T * 1e6))

"""BeamSearch end to end (bench.py's BeamSearch_end_to_end: bigram model, width 16, eos=0, batch 1024, V=1000,
100 iterations), with the flat selection and with the sorted list per prefix (PDT_STEP_FLAT=0), and the
host's share: wall clock of the loop's enqueue against the time the queue needs."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
import bench
from pydrobert_amd import modules as M
from pydrobert_amd import switches
dev = torch.device("cuda:0")
V, K, N = 1000, 16, 1024
lm = M.LookupLanguageModel(V, V, [d.copy() for d in bench.synthetic_bigram_dicts(V)]).to(dev)
bs = M.BeamSearch(lm, K, eos=0).to(dev)
with torch.no_grad():
    bs(None, 8, 4)
    for search, flat in ((1, 1), (0, 1), (0, 0), (1, 1)):
        switches.set("PDT_BEAM_SEARCH", search)
        switches.set("PDT_STEP_FLAT", flat)
        ms = [bench.event_ms(lambda: bs(None, N, 100), reps=3, warm=1) for _ in range(3)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bs(None, N, 100)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("PDT_BEAM_SEARCH=%d PDT_STEP_FLAT=%d: ms %s; one search: host returned after %.2f ms, queue drained after %.2f ms" % (
            search, flat, ["%.3f" % m for m in ms], (t1 - t0) * 1e3, (t2 - t0) * 1e3))

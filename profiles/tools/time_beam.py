"""BeamSearch end to end: bigram LookupLanguageModel, width 16, eos=0, batch 1024, 100 iterations, V=1000."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
import bench
from pydrobert_amd import modules as M
dev = torch.device("cuda:0")
lm = bench.synthetic_bigram_lm(M, 1000, dev)
search = M.BeamSearch(lm, 16, eos=0).to(dev)
fn = lambda: search(dict(), batch_size=1024, max_iters=100)
with torch.no_grad():
    fn()
    print("BeamSearch_end_to_end ms %.2f" % bench.event_ms(fn, reps=3, warm=1))

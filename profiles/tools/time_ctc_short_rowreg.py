import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from pydrobert_amd import functional as F, switches
from bench import peaky_logits, event_ms
dev = torch.device("cuda:0")
for (T, N, V) in ((512, 4096, 200), (512, 4096, 256), (512, 4096, 320), (512, 4096, 384), (512, 4096, 511), (512, 4096, 1000)):
    lg = peaky_logits(T, N, V, dev, 1)
    outs=[]
    for m in (1, 2, 3):
        switches.set("PDT_CTC_ROWREG", m)
        outs.append(F.ctc_prefix_search(lg, 16))
        ms = event_ms(lambda: F.ctc_prefix_search(lg, 16), reps=5, warm=2)
        print(T, N, V, "mode", m, "%.3f ms" % ms, flush=True)
    print(" same", all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), all(torch.equal(a, b) for a, b in zip(outs[0], outs[2])))
    del lg

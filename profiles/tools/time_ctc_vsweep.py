import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from pydrobert_amd import functional as F, switches
from bench import peaky_logits, event_ms
dev = torch.device("cuda:0")
for (T, N, V) in ((128, 4096, 600), (128, 4096, 1000), (128, 4096, 1500), (128,4096,3000), (128, 4096, 4000), (128, 4096, 5100), (128, 4096, 5200), (128, 4096, 6000), (128, 4096, 8000), (128, 4096, 12000), (128, 4096, 16000), (1000, 256, 1000), (1000, 512, 1000), (1000,2048,1000)):
    lg = peaky_logits(T, N, V, dev, 1)
    res = []
    for m in (0, 1):
        switches.set("PDT_CTC_ROWREG", m)
        ms = event_ms(lambda: F.ctc_prefix_search(lg, 16), reps=3, warm=1)
        res.append("mode %d %.3f ms %.2f TB/s" % (m, ms, lg.numel() * 4 / ms / 1e9))
    print(T, N, V, " | ".join(res), flush=True)
    del lg

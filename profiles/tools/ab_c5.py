import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from pydrobert_amd import functional as F
from bench import peaky_logits, event_ms
dev = torch.device("cuda:0")
lg = peaky_logits(512, 4096, 5000, dev, 0x5EED0006)
ms = [event_ms(lambda: F.ctc_prefix_search(lg, 16), reps=5, warm=2) for _ in range(3)]
print(os.environ.get("PDT_AMD_LIB", "default"), ["%.3f" % m for m in ms])

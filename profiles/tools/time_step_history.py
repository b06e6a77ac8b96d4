"""ctc_prefix_search_advance at N=1024, K=16, V=1000 against the length S of the history it has to copy
(the (S, N, K) int64 tensor in, (S + 1, N, K) out -- what the step function's signature forces on a host loop
around an arbitrary language model): the slope is the copy, the intercept everything else."""
import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from bench import event_ms
from pydrobert_amd import functional as F
dev = torch.device("cuda:0")
N, K, V = 1024, 16, 1000
g = torch.Generator(device=dev).manual_seed(4)
p = torch.randn((N, V + 1), device=dev, generator=g).softmax(1)
ext = torch.rand((N, K, V), device=dev, generator=g) * p[:, None, :V]
nb, b = torch.rand((N, K), device=dev, generator=g), torch.rand((N, K), device=dev, generator=g)
out = {}
for S in (1, 100, 250, 500, 1000):
    y = torch.randint(0, V, (S, N, K), device=dev, generator=g)
    lens = torch.randint(1, S + 1, (N, K), device=dev, generator=g)
    last = y.gather(0, (lens - 1).unsqueeze(0)).squeeze(0)
    isp = torch.eye(K, dtype=torch.bool, device=dev).expand(N, K, K).contiguous()
    ms = event_ms(lambda: F.ctc_prefix_search_advance((ext, p[:, :V], p[:, V]), K, (nb, b), y, last, lens, isp), reps=7, warm=2)
    out[S] = ms
    print("S = %4d: %.4f ms  (history bytes in + out %.1f MB)" % (S, ms, (2 * S + 1) * N * K * 8 / 1e6), flush=True)
s = sorted(out)
slope = (out[s[-1]] - out[s[1]]) / (s[-1] - s[1])
print(json.dumps({"ms_by_history_rows": out, "us_per_history_row": slope * 1e3,
                  "history_GBs_at_the_slope": 2 * N * K * 8 / (slope * 1e-3) / 1e9,
                  "mean_copy_us_per_frame_of_a_1000_frame_search": slope * 1e3 * 500}))

"""Per-phase wave time of the LM-fused frame kernel (a -DPDT_LM_STAMPS build: bash profiles/tools/build_var.sh
lmstamps ctc_lm_step.hip -DPDT_LM_STAMPS; PDT_AMD_LIB=<that lib> python profiles/tools/stamps_lm.py)."""
import ctypes, os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "pydrobert-pytorch_amd")
import bench
from pydrobert_amd import modules as M, _cabi
dev = torch.device("cuda:0")
T, N, V, K = 1000, 1024, 1000, 16
lg = bench.peaky_logits(T, N, V, dev, 0x5EED0003)
lm = bench.synthetic_bigram_lm(M, V, dev)
search = M.CTCPrefixSearch(K, 0.2, lm)
L = _cabi.lib()
L.pdt_debug_lm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
with torch.no_grad():
    search(lg[:8])
    L.pdt_debug_lm_stamps(None, 1)
    search(lg)
    L.pdt_debug_lm_stamps(buf, 0)
names = ["setup", "LM rows", "statistics + mix", "list selection", "barrier waits", "frame (wave 0)", "slots + histories"]
launches = buf[7]
tot = sum(buf[i] for i in range(7))
waves = 4
for i, nm in enumerate(names):
    # 100 MHz counter: 10 ns per tick; per wave and launch
    print("%-20s %6.1f%%   %7.2f us per wave per frame" % (nm, 100.0 * buf[i] / tot, buf[i] * 0.01 / (launches * waves)))
print("launches (workgroups)", launches, " total per wave per frame %.2f us" % (tot * 0.01 / (launches * waves)))

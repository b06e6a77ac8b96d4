"""The host read in front of beam_search_advance with lengths: torch's `y_prev_lens.max().item()` against
pdt_lens_reach + stream synchronise (N=1024, K=16), wall clock per read on an idle device."""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "pydrobert-pytorch_amd"))
import torch
from pydrobert_amd import _cabi
dev = torch.device("cuda:0")
ypl = torch.full((1024, 16), 100, device=dev)
def old(): return int(ypl.max().item()) >= 100
def new():
    flag = _cabi.host_flag()
    _cabi.lib().pdt_lens_reach(_cabi.ptr(ypl), ypl.stride(0), ypl.stride(1), 1024, 16, 100, flag.ptr, _cabi.stream_ptr(dev))
    _cabi.stream_synchronize(dev)
    return bool(flag.value & 1)
def new_devsync():
    flag = _cabi.host_flag()
    _cabi.lib().pdt_lens_reach(_cabi.ptr(ypl), ypl.stride(0), ypl.stride(1), 1024, 16, 100, flag.ptr, _cabi.stream_ptr(dev))
    torch.cuda.synchronize(dev)
    return bool(flag.value & 1)
def new_poll():
    flag = _cabi.host_flag()
    _cabi.lib().pdt_lens_reach(_cabi.ptr(ypl), ypl.stride(0), ypl.stride(1), 1024, 16, 100, flag.ptr, _cabi.stream_ptr(dev))
    a = flag._np
    t_end = time.perf_counter() + 2e-3
    while a[0] == 0 and time.perf_counter() < t_end:
        pass
    if a[0] == 0:
        _cabi.stream_synchronize(dev)
    return bool(a[0] & 1)
for name, fn in (("max().item()", old), ("lens_reach + poll", new_poll), ("lens_reach + stream sync", new), ("lens_reach + device sync", new_devsync), ("max().item()", old)):
    assert fn() is True
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): fn()
    print("%-28s %.1f us per read" % (name, (time.perf_counter() - t0) / 500 * 1e6))

"""ctypes binding of libpdt_amd.so -- the C ABI declared in include/pdt_amd.h.

This is the only place the host package touches native code.  The library is built
in-tree by ``csrc/Makefile`` (``__graft_entry__.build()``); if it is missing or fails
to load, every operator raises -- there is no fallback path.
"""

import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDT_AMD_LIB", os.path.join(_HERE, "_lib", "libpdt_amd.so"))

PDT_OK = 0
PDT_E_ARG = -1
PDT_E_TOO_LONG = -2
PDT_E_UNSUPPORTED = -3
MODE_FINAL, MODE_PREFIX = 0, 1
WARN_REF_NO_EOS, WARN_HYP_NO_EOS, WARN_EMPTY_REF = 1, 2, 4

_c = ctypes
_P = _c.c_void_p
_I64 = _c.c_int64
_INT = _c.c_int
_F = _c.c_float

# name -> (restype, argtypes); mirrors include/pdt_amd.h declaration by declaration
SIGNATURES = {
    "pdt_amd_abi_version": (_INT, []),
    "pdt_amd_set_switch": (_INT, [_c.c_char_p, _INT]),
    "pdt_amd_get_switch": (_INT, [_c.c_char_p, _P]),
    "pdt_lev": (
        _INT,
        [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _I64, _INT, _I64, _INT, _F, _F, _F]
        + [_INT, _INT, _INT, _F, _INT, _P, _I64, _I64, _P, _P, _P, _P, _I64, _P],
    ),
    "pdt_lev_classified": (
        _INT,
        [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _I64, _INT, _I64, _INT, _F, _F, _F]
        + [_INT, _INT, _INT, _F, _INT, _P, _I64, _I64, _P, _P, _P, _P, _I64, _P],
    ),
    "pdt_lev_workspace_bytes": (_I64, [_I64, _I64, _I64]),
    "pdt_fill_after_eos": (_INT, [_P, _I64, _I64, _I64, _I64, _P, _I64, _I64, _P, _P]),
    "pdt_oc_mask_words": (_I64, [_I64]),
    "pdt_oc_mask": (
        _INT,
        [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _I64, _INT, _I64, _INT, _F, _F, _F]
        + [_INT, _P, _P, _P, _P, _P, _I64, _P],
    ),
    "pdt_oc_mask_workspace_bytes": (_I64, [_I64, _I64, _I64]),
    "pdt_oc_expand": (_INT, [_P, _P, _I64, _I64, _I64, _I64, _I64, _P, _I64, _I64, _P]),
    "pdt_ocd_loss_forward": (
        _INT, [_P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P, _I64, _P, _I64, _P, _P, _P, _P],
    ),
    "pdt_ocd_loss_backward": (
        _INT, [_P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P, _I64, _P, _I64, _P, _P, _P],
    ),
    "pdt_ctc_prefix_search_advance": (
        _INT,
        [_P, _I64, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64, _I64]
        + [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64, _P, _I64, _I64]
        + [_P, _I64, _I64, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    ),
    "pdt_ctc_prefix_search_advance_lm": (
        _INT,
        [_P, _F, _INT, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64, _I64]
        + [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64, _P, _I64, _I64]
        + [_P, _I64, _I64, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    ),
    "pdt_beam_search_table": (
        _INT,
        [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _I64, _I64, _INT, _I64, _INT, _P, _P, _P, _P, _P, _P],
    ),
    "pdt_beam_search_table_paths": (_INT, [_P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P]),
    "pdt_lens_reach": (_INT, [_P, _I64, _I64, _I64, _I64, _I64, _P, _P]),
    "pdt_beam_search_advance": (
        _INT,
        [_P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64]
        + [_P, _I64, _I64, _I64, _P, _P, _P, _P, _P],
    ),
    "pdt_ctc_lookup_lm_advance": (
        _INT,
        [_P, _I64, _I64, _P, _I64, _I64, _I64, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64]
        + [_P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64]
        + [_F, _INT, _P, _P, _P, _P, _P, _P, _P, _P, _INT, _P, _I64, _I64, _I64, _I64, _P],
    ),
    "pdt_ctc_lookup_lm_search_workspace_bytes": (_I64, [_I64, _I64, _I64, _I64, _I64, _I64]),
    "pdt_ctc_lookup_lm_search": (
        _INT,
        [_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _F, _INT]
        + [_P, _P, _P, _P, _P, _I64, _P],
    ),
    "pdt_lm_factor_table": (_INT, [_P, _I64, _I64, _F, _INT, _P, _I64, _P]),
    "pdt_ctc_lm_table_search_workspace_bytes": (_I64, [_I64, _I64, _I64, _I64]),
    "pdt_ctc_lm_table_search": (
        _INT,
        [_P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _I64, _I64, _P, _P, _I64, _I64, _I64, _I64, _I64, _F, _INT, _P, _P, _P, _P, _P],
    ),
    "pdt_beam_search_step": (
        _INT,
        [_P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64]
        + [_P, _I64, _I64, _INT, _I64, _INT, _I64, _P, _P, _P, _P, _P, _P, _P],
    ),
    "pdt_beam_search_step_table": (
        _INT,
        [_P, _I64, _I64, _I64, _P, _P, _I64, _I64, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64]
        + [_P, _I64, _I64, _INT, _I64, _INT, _I64, _P, _P, _P, _P, _P, _P, _P],
    ),
    "pdt_row_log_softmax_stats": (_INT, [_P, _I64, _I64, _I64, _I64, _P, _P]),
    "pdt_ctc_greedy_search": (
        _INT, [_P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _I64, _INT, _P, _P, _I64, _I64, _P, _P],
    ),
    "pdt_sequence_log_probs_forward": (_INT, [_P, _P, _I64, _I64, _I64, _I64, _INT, _I64, _P, _P]),
    "pdt_sequence_log_probs_backward": (
        _INT, [_P, _P, _I64, _I64, _I64, _I64, _INT, _I64, _P, _P, _P],
    ),
    "pdt_spline_workspace_bytes": (_I64, [_I64, _I64, _I64, _I64]),
    "pdt_spline_solve": (_INT, [_P, _P, _P, _I64, _I64, _I64, _I64, _INT, _F, _P, _P, _P]),
    "pdt_polyharmonic_spline": (
        _INT, [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _INT, _F, _P, _P, _P],
    ),
    "pdt_warp_1d_grid": (_INT, [_P, _P, _P, _I64, _I64, _INT, _P, _P]),
    "pdt_spec_augment_draw": (
        _INT,
        [_P, _I64, _I64, _P, _I64, _I64, _F, _F, _I64, _I64, _F, _I64, _F, _I64, _INT, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    ),
    "pdt_spec_augment_apply": (
        _INT,
        [_P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P, _P, _P, _I64, _P, _P, _I64, _P, _P],
    ),
    "pdt_spec_augment_apply_warp": (
        _INT,
        [_P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P, _P, _INT, _P, _P, _I64, _P, _P, _I64, _P, _P, _P],
    ),
    "pdt_dense_image_warp": (_INT, [_P, _P, _I64, _I64, _I64, _I64, _INT, _INT, _INT, _P, _P]),
    "pdt_sparse_image_warp": (
        _INT,
        [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _INT, _F, _INT, _INT, _INT, _P, _P, _INT, _P, _P],
    ),
    "pdt_spec_augment_apply_backward": (
        _INT,
        [_P, _I64, _I64, _I64, _P, _P, _P, _P, _I64, _P, _P, _I64, _P, _P],
    ),
    "pdt_dense_image_warp_backward": (
        _INT,
        [_P, _P, _I64, _I64, _I64, _I64, _INT, _INT, _INT, _P, _P],
    ),
    "pdt_sparse_image_warp_backward": (
        _INT,
        [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _INT, _F, _INT, _INT, _INT, _P, _P, _P],
    ),
    # float64 images: the same argument lists, double pixels
    "pdt_dense_image_warp_f64": (_INT, [_P, _P, _I64, _I64, _I64, _I64, _INT, _INT, _INT, _P, _P]),
    "pdt_dense_image_warp_backward_f64": (_INT, [_P, _P, _I64, _I64, _I64, _I64, _INT, _INT, _INT, _P, _P]),
    "pdt_sparse_image_warp_f64": (
        _INT,
        [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _INT, _F, _INT, _INT, _INT, _P, _P, _INT, _P, _P],
    ),
    "pdt_sparse_image_warp_backward_f64": (
        _INT,
        [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _INT, _F, _INT, _INT, _INT, _P, _P, _P],
    ),
    "pdt_lookup_lm_log_probs": (
        _INT,
        [_P, _I64, _I64, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P, _P],
    ),
    "pdt_fusion_ext": (_INT, [_P, _I64, _I64, _I64, _P, _I64, _I64, _P, _I64, _F, _INT, _P, _P]),
    "pdt_pad_variable": (_INT, [_P, _I64, _I64, _I64, _I64, _P, _P, _INT, _P, _I64, _P, _P]),
    "pdt_pad_variable_backward": (_INT, [_P, _I64, _I64, _I64, _P, _P, _INT, _I64, _P, _P]),
    "pdt_ctc_prefix_search_workspace_bytes": (_I64, [_I64, _I64, _I64, _I64]),
    "pdt_ctc_prefix_search_plan": (_INT, [_I64, _I64, _P]),
    "pdt_ctc_prefix_search": (
        _INT,
        [_P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _P, _P],
    ),
}

_lib = None


def lib():
    """Load (once) and return the native library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "pydrobert_amd: native library {} not found. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C pydrobert-pytorch_amd/csrc`; there is no fallback "
                "implementation.".format(LIB_PATH)
            )
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    """Map a C-ABI status to the reference's exception type (RuntimeError)."""
    if rc == PDT_OK:
        return
    if rc == PDT_E_ARG:
        raise RuntimeError("{}: invalid argument".format(what))
    if rc == PDT_E_TOO_LONG:
        raise RuntimeError("{}: sequence too long for the MI355X kernel".format(what))
    raise RuntimeError("{}: HIP error {}".format(what, rc))


def require_hip(*tensors):
    """All tensors must sit on one ROCm device; returns that device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise RuntimeError(
                "pydrobert_amd operators run only on ROCm (HIP) device tensors; got a "
                "tensor on '{}'. There is no CPU implementation in this package.".format(t.device)
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("tensors are on different devices: {} and {}".format(dev, t.device))
    return dev


_HOST_FLAGS = threading.local()


class HostFlag:
    """One int32 in pinned host memory: ``ptr`` for a kernel to store to, ``value`` for the host."""

    def __init__(self):
        self._t = torch.zeros(1, dtype=torch.int32, pin_memory=True)
        self._np = self._t.numpy()  # (a view: reads and writes without torch's indexing machinery)
        self.ptr = self._t.data_ptr()

    @property
    def value(self) -> int:
        return int(self._np[0])


def host_flag() -> HostFlag:
    """One word in pinned host memory per host thread, zeroed: a word a kernel raises (a data-dependent
    verdict the reference reaches with a reduction, a copy and a synchronisation in FRONT of the work)
    for the host to read once the stream has drained.  A caller synchronises before it returns, so a
    thread never has two in flight."""
    flag = getattr(_HOST_FLAGS, "flag", None)
    if flag is None:
        flag = _HOST_FLAGS.flag = HostFlag()
    flag._np[0] = 0
    return flag


def stream_ptr(device):
    try:  # (the raw handle, without a Stream object around it)
        return torch._C._cuda_getCurrentRawStream(device.index)
    except (AttributeError, TypeError):
        return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def stream_synchronize(device):
    """hipStreamSynchronize of the current stream of ``device``."""
    torch.cuda.current_stream(device).synchronize()


def wait_flag(flag, device, spin_seconds=5e-4):
    """The value of a word a kernel RELEASES at system scope (non-zero once written): polled for half a
    millisecond -- pinned host memory is coherent by default, and the interrupt behind hipStreamSynchronize
    costs ~15 us -- then the stream is synchronised (a non-coherent host allocation, a long queue)."""
    import time

    word = flag._np
    t_end = time.perf_counter() + spin_seconds
    while word[0] == 0:
        if time.perf_counter() > t_end:
            stream_synchronize(device)
            break
    return int(word[0])


def on_device(device):
    """``with on_device(d):`` -- torch.cuda.device(d), or nothing when ``d`` is the current device already
    (the guard's own bookkeeping is several microseconds of a step function's call)."""
    if device.index is None or device.index == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def plain_call(*tensors):
    """True when a call has nothing for the dispatcher to do: no tracing (torch.compile / make_fx / fake
    tensors), no function transform, no dispatch or function mode, and no input autograd would follow.
    The Python wrappers then call the implementation behind their ``custom_op`` directly -- the dispatch of
    a Python custom op costs ~16 us, as much as a step function's kernel.  Same code either way."""
    if torch.compiler.is_compiling():
        return False
    try:
        if (torch._C._len_torch_dispatch_stack() or torch._C._len_torch_function_stack()
                or torch._C._functorch.peek_interpreter_stack() is not None):
            return False
    except AttributeError:  # (another torch: leave it to the dispatcher)
        return False
    grad = torch.is_grad_enabled()
    for t in tensors:
        if t is not None and (type(t) is not torch.Tensor or (grad and t.requires_grad)):
            return False
    return True


def ptr(t):
    return 0 if t is None else t.data_ptr()

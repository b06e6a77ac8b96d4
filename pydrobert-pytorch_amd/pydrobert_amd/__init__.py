"""MI355X-native sequence-level hot path of pydrobert-pytorch.

``pydrobert_amd.functional`` / ``pydrobert_amd.modules`` mirror the names, argument
order, defaults, return shapes/dtypes and error behaviour of ``pydrobert.torch.functional``
/ ``pydrobert.torch.modules`` for the operators on that path.  Every operator runs as
hand-written HIP kernels (``csrc/``) reached through the C ABI in ``include/pdt_amd.h``;
there is no CPU path -- tensors must live on a ROCm device and a missing
``libpdt_amd.so`` is an error, never a silent fallback.
"""

__version__ = "0.1.0"

from . import config  # noqa: F401

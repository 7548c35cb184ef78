"""vkvolume_amd — MI355X-native volume ray-caster hot path (Chebyshev-distance-map empty-space skipping).

The product is the HIP library ``csrc/libvkvolume_amd.so`` behind the C ABI of ``include/vkvolume_amd.h``;
this package is its Python binding for tests and the bench driver (ctypes; torch only supplies device
memory, streams and torch.distributed).  Nothing here falls back to a CPU path: if the library is missing
every call raises.
"""
from . import abi  # noqa: F401

__all__ = ["abi"]
__version__ = "0.1.0"

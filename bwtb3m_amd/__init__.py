"""bwtb3m_amd -- MI355X-native BWT construction by balanced block merging.

Host-side mirror (Python) of the reference's interface for the hot path
(/root/reference/src/bwtb3m.cpp:62-63) over the C ABI of libb3m.so (include/b3m.h).
The compute lives in hand-written HIP kernels (bwtb3m_amd/csrc); there is no CPU
fallback: importing works anywhere, computing needs the built library and a GPU.
"""
from ._lib import lib, LibraryMissing, B3MError, INPUT_TYPES  # noqa: F401
from .options import BwtMergeSortOptions, ArgInfo  # noqa: F401
from .bwtmergesort import BwtMergeSort  # noqa: F401

__all__ = ["lib", "LibraryMissing", "B3MError", "INPUT_TYPES", "BwtMergeSortOptions", "ArgInfo", "BwtMergeSort"]

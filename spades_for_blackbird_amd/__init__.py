"""MI355X-native k-mer counting / de Bruijn graph construction engine (drop-in for the
`spades-kmercount` and `spades-gbuilder` hot path of SPAdes 3.15.4).

The product is the C-ABI shared library `libbbk.so` (include/bbk.h; HIP kernels for gfx950 in
csrc/).  This package is the thin Python binding used by tests and bench.py; there is no CPU
fallback: importing works anywhere, but creating a Context without the library or without a
GPU raises.
"""
from .engine import (Context, Reads, KMerSet, ExtIndex, Unitigs, BBKError, lib_path, load_library,  # noqa: F401
                     BOTH_STRANDS, CANONICAL, WITH_COUNTS, UNSORTED, REFERENCE_ORDER, WITH_MASKS, ORDER_SORTED, ORDER_REFERENCE_BUCKETS16)

__all__ = ["Context", "Reads", "KMerSet", "ExtIndex", "Unitigs", "BBKError", "lib_path", "load_library",
           "BOTH_STRANDS", "CANONICAL", "WITH_COUNTS", "UNSORTED", "REFERENCE_ORDER", "WITH_MASKS", "ORDER_SORTED", "ORDER_REFERENCE_BUCKETS16"]

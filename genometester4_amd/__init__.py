"""genometester4_amd -- MI355X-native sorted k-mer list set operations (glistcompare hot path).

The product is the C-ABI shared library built from genometester4_amd/csrc (HIP kernels for
gfx950 + a C host layer) and the `glistcompare` drop-in CLI linked against it.  The Python in
this package is plumbing for tests and bench.py: ctypes bindings (capi) and numpy .list I/O.
"""
__all__ = ["capi", "listio"]

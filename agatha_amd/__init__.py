"""agatha_amd -- MI355X-native guided (banded, affine-gap, z-drop) sequence alignment.

The product is the HIP library agatha_amd/libagatha_amd.so (sources in agatha_amd/csrc, C-ABI in
include/agatha_amd.h) plus the GASAL2/AGAThA-compatible C++ host layer (agatha_amd/host).  This Python
package is a thin ctypes binding of that C-ABI for tests, bench.py and scripting; it has NO CPU fallback:
loading fails loudly when the library is missing.
"""
from .engine import (Engine, Scores, DeviceBatch, load_library, library_path, build_library,  # noqa: F401
                     AgathaError, set_debug_option, get_debug_option, debug_options, pack_host, pack2_host)

__all__ = ["Engine", "Scores", "DeviceBatch", "load_library", "library_path", "build_library", "AgathaError",
           "set_debug_option", "get_debug_option", "debug_options", "pack_host", "pack2_host"]

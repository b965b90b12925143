"""gbnns_dim_red_amd -- MI355X-native two-stage graph ANN search (Shekhale/gbnns_dim_red drop-in).

The product is the C-ABI library `lib/libgbnns_hip.so` (include/gbnns.h) plus the C++ drop-in
headers in `search/`.  This Python package is tooling around it: a ctypes binding used by the
tests and by bench.py, and the build driver.  It never imports `oracle` and has no CPU
implementation of the search path: if the HIP library is missing, importing the binding raises.
"""
from .binding import (  # noqa: F401
    FLAG_AUX_GRAPH, FLAG_BITMAP_PASS, FLAG_LLF, FLAG_NO_FUSED_RERANK, FLAG_WIDE_INDEX, FLAG_SERIAL, FLAG_DEFER_JOIN, FLAG_MFMA_PROJECTION, GbnnsError, Index, MultiIndex, METRIC_L2, METRIC_NEG_DOT, MODE_LOWQ, MODE_NET, MODE_PLAIN, build_graph_gd, build_graph_gd_device,
    device_count, exact_knn, lib_path, load_library, version,
)
from .build import build_library  # noqa: F401

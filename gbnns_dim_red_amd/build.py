"""Build driver for lib/libgbnns_hip.so (hipcc, gfx950) and the C++ drop-in driver."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SEARCH = os.path.join(_HERE, "search")
LIB = os.path.join(_HERE, "lib", "libgbnns_hip.so")


def build_library(verbose=False):
    """Compile every HIP/C++ source of the package in-tree (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-j8", "-C", CSRC, "all"], stdout=out)  # a dozen independent objects
    if os.path.exists(os.path.join(SEARCH, "Makefile")):
        subprocess.check_call(["make", "-C", SEARCH, "all"], stdout=out)
    if not os.path.exists(LIB):
        raise RuntimeError("build did not produce " + LIB)
    return LIB

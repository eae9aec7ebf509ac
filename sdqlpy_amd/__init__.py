"""sdqlpy_amd — MI355X-native execution backend for sdqlpy's scan / hash-join / group-by hot path.

    from sdqlpy_amd.sdql_lib import *     # the reference's Python surface, HIP underneath
    from sdqlpy_amd import tpch           # TPCH schemas + deterministic generator

The compute path is the C-ABI library sdqlpy_amd/csrc/libsdqlhip.so (include/sdqh.h), built in-tree
by sdqlpy_amd.build (hipcc --offload-arch=gfx950).
"""
__version__ = "0.1.0"

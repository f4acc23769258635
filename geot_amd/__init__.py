"""geot_amd -- MI355X-native segment-reduction engine behind the GeoT operator surface.

    import geot_amd as geot          # or simply `import geot` (alias package at the repo root)
    out = geot.index_scatter(0, src, index, 'sum', sorted=True)

Layout: ``csrc/`` holds the hand-written HIP kernels and the C ABI (``include/geot_hip.h``);
``_lib`` loads ``libgeot_hip.so`` (import fails loudly when it is missing - there is no fallback);
``hip`` is the pointer-level doorway; ``ops`` mirrors the reference's operators, schemas, shape
rule and error texts; ``sharding`` partitions an edge list over ranks.
"""
from . import _lib

_lib.load()  # raises ImportError if the HIP library is not built

from . import hip  # noqa: E402
from .ops import (  # noqa: E402
    coo_to_csr,
    csr_gws,
    csr_gws_impl,
    gather_scatter,
    gather_scatter_impl,
    gather_weight_scatter,
    gather_weight_scatter_impl,
    get_reduction_enum,
    index_scatter,
    mh_spmm,
    mh_spmm_transposed,
    sddmm_coo_impl,
)

from .graph import Graph, PlanOrdered  # noqa: E402

__version__ = "0.1.0"

__all__ = [
    "index_scatter", "gather_scatter", "gather_weight_scatter", "mh_spmm", "mh_spmm_transposed",
    "csr_gws", "coo_to_csr", "csr_gws_impl",
    "gather_scatter_impl", "gather_weight_scatter_impl", "sddmm_coo_impl", "get_reduction_enum", "hip",
    "Graph", "PlanOrdered",
]

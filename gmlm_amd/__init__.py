"""gmlm_amd — MI355X (gfx950) kernel path behind the GraphTextLM forward/backward of chungimungi/GMLM.

Python host (this package) -> C ABI (include/gmlm_hip.h, libgmlm_hip.so) -> hand-written HIP kernels.
Importing the package does not need a GPU; calling any operator does, and fails loudly otherwise.
"""
from ._lib import GmlmHipError, LIB_PATH, lib  # noqa: F401
from .graph import GraphCache, RelCSR, build_rel_csr  # noqa: F401
from .model import GraphTextLM, TokenizedTexts  # noqa: F401
from .nn import CrossAttention, GraphNorm, MultiScaleFusion, RGCNConv, degree  # noqa: F401
from .ops import edge_types_from_degree, soft_masking_gnn_input  # noqa: F401

__all__ = ["GraphTextLM", "TokenizedTexts", "RGCNConv", "GraphNorm", "CrossAttention", "MultiScaleFusion", "degree",
           "soft_masking_gnn_input", "edge_types_from_degree", "build_rel_csr", "GraphCache", "RelCSR", "lib",
           "GmlmHipError", "LIB_PATH"]

"""gcgcn_amd -- MI355X-native CAGGC + MAGGC graph-convolution blocks of GCGCN.

Drop-in ``nn.Module``s (same names / constructor args / forward signatures / state_dict keys as the
reference's ``models/GCGCN_glove.py:18-168``) whose arithmetic runs in hand-written HIP kernels for
gfx950 behind a C ABI (``include/gcgcn.h`` -> ``gcgcn_amd/lib/libgcgcn_hip.so``).  No CPU fallback.
"""
from .modules import (ClassifierHead, EdgeFeatureProducer, GraphModelTail, GATAttention, GraphConv, GraphConvolution, GraphHops,  # noqa: F401
                      MultiGraphConvolution, MultiHeadAttention)
from .functional import manual_seed, pair_bce_loss  # noqa: F401
from . import functional, models, optim, params  # noqa: F401
from .optim import FusedAdam  # noqa: F401

__all__ = ["GraphConv", "GATAttention", "MultiHeadAttention", "GraphConvolution", "MultiGraphConvolution", "GraphHops",
           "EdgeFeatureProducer", "ClassifierHead", "GraphModelTail",
           "manual_seed", "pair_bce_loss", "FusedAdam", "functional", "models", "optim", "params"]

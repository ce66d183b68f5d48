"""plnlp_amd -- MI355X-native (gfx950) implementation of the PLNLP training hot
path: SAGE/GCN neighbour aggregation, pairwise pos/neg edge scoring and the
AUC/hinge ranking losses, behind the reference's nn.Module / factory surface
(plnlp/layer.py, plnlp/model.py, plnlp/loss.py of zhitao-wang/PLNLP).

Compute runs in hand-written HIP kernels (plnlp_amd/csrc) reached through a
C-ABI shared library (include/plnlp_hip.h); PyTorch-ROCm only owns device
memory, streams and torch.distributed.  There is no CPU fallback.
"""
from .graph import Graph, gcn_normalization, adj_normalization  # noqa: F401
from . import ops  # noqa: F401
from .ops import manual_seed  # noqa: F401
from .layer import (BaseGNN, SAGE, GCN, WSAGE, Transformer, SAGEConv, GCNConv,  # noqa: F401
                    MLPPredictor, MLPCatPredictor, MLPDotPredictor, MLPBilPredictor,
                    DotPredictor, BilinearPredictor)
from .model import (BaseModel, create_input_layer, create_gnn_layer,  # noqa: F401
                    create_predictor_layer, adjust_lr)
from . import loss, negative_sample, utils  # noqa: F401
from .logger import Logger  # noqa: F401

__version__ = "0.1.0"

"""plnlp_amd -- MI355X-native (gfx950) implementation of the PLNLP training hot
path: SAGE/GCN neighbour aggregation, pairwise pos/neg edge scoring and the
AUC/hinge ranking losses, behind the reference's nn.Module / factory surface
(plnlp/layer.py, plnlp/model.py, plnlp/loss.py of zhitao-wang/PLNLP).

Compute runs in hand-written HIP kernels (plnlp_amd/csrc) reached through a
C-ABI shared library (include/plnlp_hip.h); PyTorch-ROCm only owns device
memory, streams and torch.distributed.  There is no CPU fallback.
"""
import os as _os

import torch as _torch

# hipGraph replays (plnlp_amd/capture.py) need ROCm's "graph packet capture" OFF: with it on (the ROCm 7 default),
# a graph that has been launched once and is launched again after ANY device-to-host read in between (a loss
# `.item()`, the touched-row count) faults with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION -- reproduced on MI355X
# with nothing but torch.cuda.CUDAGraph + our kernels, gone with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
# (profiles/r03_capture_debug.md).  The HIP runtime reads the flag when it initialises, so it is set here, at import --
# but ONLY when the captured loop was asked for (PLNLP_CAPTURE=1, or =available: the flag alone) and the process has not touched the GPU yet: the flag is
# a process-wide runtime switch that an embedding application and its children inherit, and the captured loop is opt-in.
# Otherwise the environment is left alone and the step pipeline stays eager (StepPipeline.why_eager says why).
_PACKET_FLAG = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_preset = _os.environ.get(_PACKET_FLAG)
_early = not _torch.cuda.is_initialized()
if _preset is None and _early and _os.environ.get("PLNLP_CAPTURE") in ("1", "available"):     # "available": flag only, loop stays eager
    _os.environ[_PACKET_FLAG] = "0"
GRAPH_REPLAY_SAFE = _os.environ.get(_PACKET_FLAG) == "0" and (_preset == "0" or _early)

from .graph import Graph, gcn_normalization, adj_normalization  # noqa: F401,E402
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

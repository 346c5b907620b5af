"""Parameter holders shared by the SAM parts. ref: lib/sam_model/common.py (names only; compute is in csrc/)."""
import torch
from torch import nn


class MLPBlock(nn.Module):
    def __init__(self, embedding_dim: int, mlp_dim: int):
        super().__init__()
        self.lin1 = nn.Linear(embedding_dim, mlp_dim)
        self.lin2 = nn.Linear(mlp_dim, embedding_dim)


class LayerNorm2d(nn.Module):
    def __init__(self, num_channels: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps


def no_standalone_forward(self, *a, **k):
    raise RuntimeError(f"{type(self).__name__} is a parameter holder of the cor_amd HIP engine; call the top-level "
                       "CirSegModelWithQuerySupportFeat.forward (or cor_amd.engine functions) instead")

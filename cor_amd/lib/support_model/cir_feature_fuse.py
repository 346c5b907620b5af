"""Gated image-text fusion (AVTI) parameter tree (names of lib/support_model/cir_feature_fuse.py:20-42).
nn.Sequential indices 0 and 3 carry the two Linear layers, as in the reference (ReLU/Dropout/Sigmoid between)."""
from torch import nn


def _gate(d_in, d_mid, d_out):
    return nn.Sequential(nn.Linear(d_in, d_mid), nn.ReLU(), nn.Dropout(0.5), nn.Linear(d_mid, d_out), nn.Sigmoid())


class CirFuseModule(nn.Module):
    def __init__(self, image_embed_dim, text_embed_dim):
        super().__init__()
        cat = image_embed_dim + text_embed_dim
        self.atten_Image = _gate(cat, image_embed_dim, image_embed_dim)
        self.atten_Text = _gate(cat, text_embed_dim, text_embed_dim)
        self.dynamic_scalar = _gate(cat, image_embed_dim, 1)

"""Cut-down prompt encoder parameter tree (names of lib/sam_model/my_prompt_encoder.py:42,57,186-189).
Both of its outputs are input independent and are folded at pack time (cor_amd.engine.pack)."""
import torch
from torch import nn


class PositionEmbeddingRandom(nn.Module):
    def __init__(self, num_pos_feats: int = 64, scale: float = None):
        super().__init__()
        if scale is None or scale <= 0.0:
            scale = 1.0
        self.register_buffer("positional_encoding_gaussian_matrix", scale * torch.randn((2, num_pos_feats)))


class PromptEncoder(nn.Module):
    def __init__(self, embed_dim: int, image_embedding_size):
        super().__init__()
        self.embed_dim, self.image_embedding_size = embed_dim, tuple(image_embedding_size)
        self.pe_layer = PositionEmbeddingRandom(embed_dim // 2)
        self.no_mask_embed = nn.Embedding(1, embed_dim)

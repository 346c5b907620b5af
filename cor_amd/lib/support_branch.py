"""SupportBranch: parameter tree + validation of the reference (lib/support_branch.py:14-54)."""
from torch import nn

from .support_model.siglip_openclip import SigLIP
from .support_model.cir_feature_fuse import CirFuseModule
from .support_model.mask_adapter import MaskAdapterPooling, MaskedPooling, LayerNorm

_DIMS = {"ViT-SO400M-14-SigLIP-384": 1152, "ViT-B-16-SigLIP-384": 768, "ViT-B-16-SigLIP2-384": 768,
         "ViT-L-16-SigLIP-384": 1024, "ViT-L-16-SigLIP2-384": 1024}


class SupportBranch(nn.Module):
    def __init__(self, clip_model: str, siglip_path: str, mask_pooling: str = "MaskedPooling", siglip_cfg: dict = None):
        super().__init__()
        if clip_model not in _DIMS:
            raise ValueError(f"Invalid SigLIP model: {clip_model}")
        self.siglip = SigLIP(clip_model, siglip_path, cfg=siglip_cfg)
        self.siglip_dim = _DIMS[clip_model]
        self.mask_pooling_name = mask_pooling
        if mask_pooling == "MaskAdapterPooling":
            self.mask_pooling = MaskAdapterPooling(x_in_channel=self.siglip_dim, mask_adatpet_network_in_channel=512,
                                                   mask_downscaling_mid_channel=16, mask_adatpet_network_mid_channel=256,
                                                   num_output_maps=8)
        elif mask_pooling == "MaskedPooling":
            self.mask_pooling = MaskedPooling()
        else:
            raise ValueError(f"Invalid mask pooling method: {mask_pooling}")
        self.cir_fuse = CirFuseModule(image_embed_dim=self.siglip_dim, text_embed_dim=self.siglip_dim)
        self.ln_channel_first = LayerNorm(self.siglip_dim, eps=1e-6, data_format="channels_first")
        self.ln_channel_last = LayerNorm(self.siglip_dim, eps=1e-6, data_format="channels_last")
        self.dim_proj = nn.Sequential(nn.Linear(self.siglip_dim, 512), nn.GELU(), nn.Dropout(0.8),
                                      nn.Linear(512, 256), nn.GELU(), nn.Dropout(0.8))
